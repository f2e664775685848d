#!/usr/bin/env python3
"""Summarise one steady-state denoising step from a rocprofv3 kernel trace CSV."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# steps delimited by stem_conv ... cfg_ddpm
starts = [i for i, n in enumerate(names) if 'stem_conv' in n]
ends = [i for i, n in enumerate(names) if 'cfg_ddpm' in n]
# choose a step in the timed region: the 15th complete one
pairs = []
for s in starts:
    e = next((x for x in ends if x > s), None)
    if e is not None: pairs.append((s, e))
which = int(sys.argv[2]) if len(sys.argv) > 2 else min(15, len(pairs) - 1)
s, e = pairs[which]
step = rows[s:e + 1]
t0 = int(step[0]['Start_Timestamp']); t1 = int(step[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
print(f"step #{which}: {len(step)} kernels, wall {(t1-t0)/1e3:.1f} us, kernel-busy {busy/1e3:.1f} us, gaps {(t1-t0-busy)/1e3:.1f} us")
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r['Kernel_Name']
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = n[:60]
    agg[n][0] += 1; agg[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:62s} x{c:4d}  {d/1e3:9.1f} us  avg {d/c/1e3:7.2f} us")
if len(sys.argv) > 3:
    for r in step:
        print(r['Kernel_Name'][:70], (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3, r.get('Grid_Size_X'), r.get('Workgroup_Size_X'))
