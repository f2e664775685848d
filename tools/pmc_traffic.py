#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into HBM bytes per denoising step and kernel
family.  usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
FETCH_SIZE is in KB and reports half of wide reads on gfx950 (MI355X_MICROARCH.md): corrected x2.  WRITE_SIZE in KB."""
import collections, csv, json, sys


def family(name):
    if "igemm_kernel" in name or "conv3_halo" in name or "splitk_finalize" in name:
        return "igemm"
    if "attn_kernel" in name:
        return "attn"
    if "gn_" in name or "layernorm" in name:
        return "norm"
    return "other"


def per_step(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "stem_conv" in r["Kernel_Name"]]
    ends = [i for i, r in enumerate(rows) if "cfg_ddpm" in r["Kernel_Name"]]
    steps = []
    for s in starts:
        e = next((x for x in ends if x > s), None)
        if e is not None:
            steps.append((s, e))
    steps = steps[-8:]                       # steady state: the last 8 complete steps
    acc = collections.defaultdict(float)
    launches = collections.defaultdict(int)
    for s, e in steps:
        for r in rows[s:e + 1]:
            acc[family(r["Kernel_Name"])] += float(r["Counter_Value"])
            launches[family(r["Kernel_Name"])] += 1
    n = len(steps)
    return {k: v / n for k, v in acc.items()}, {k: v // n for k, v in launches.items()}, n


fetch, launches, n1 = per_step(sys.argv[1], "FETCH_SIZE")
write, _, n2 = per_step(sys.argv[2], "WRITE_SIZE")
out = {"note": "per denoising step (mean of the last 8 steps of each PMC pass); FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(gfx950 reports 1/2 of wide reads); separate rocprofv3 --pmc passes FETCH_SIZE / WRITE_SIZE around "
               "bench.py --steps 10 --warmup 10; produced by tools/pmc_traffic.py",
       "steps_averaged": [n1, n2], "per_step": {}}
for fam in sorted(set(fetch) | set(write)):
    fk, wk = fetch.get(fam, 0.0), write.get(fam, 0.0)
    out["per_step"][fam] = {"launches": launches.get(fam, 0), "fetch_KB_raw": fk, "fetch_bytes_corrected": fk * 1024 * 2,
                            "write_bytes": wk * 1024, "hbm_bytes": fk * 1024 * 2 + wk * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["per_step"], indent=1))
