#!/bin/bash
# SQ counters of the attention kernels (one rocprofv3 --pmc pass around tools/attn_bench.py)
set -e
out=$PWD/gpurun_out/${1:-attn_pmc}
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/p1 -o a -- python3 tools/attn_bench.py > $out/bench1.log 2> $out/err1.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/p2 -o a -- python3 tools/attn_bench.py > $out/bench2.log 2> $out/err2.log
python3 - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    f = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter csv in", p); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    print(p, "columns:", list(csv.DictReader(open(f[0])).fieldnames))
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        if "attn_kernel" not in name: continue
        key = (name.split("attn_kernel")[1][:12], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in agg.items():
        print(p, key, {k: round(sum(v) / len(v)) for k, v in cs.items()})
PY
rm -rf $out/p1 $out/p2
