#!/bin/bash
# Instruction mix of the attention launches from COUNTERS ("pin the bottleneck with counters instead of the
# ablation build"): two rocprofv3 --pmc passes (8 SQ slots each) around tools/attn_bench.py, per attention shape:
#   pass 1  SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
#   pass 2  SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
# Counters the installed rocprofv3 does not list are dropped from a pass (and named in the output).  usage: tools/attn_instmix.sh <tag>
set -e
out=$PWD/gpurun_out/${1:-attn_instmix}
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 -L > $out/counters_available.txt 2>&1 || true
pick() { for c in "$@"; do if grep -qw "$c" $out/counters_available.txt; then printf "%s " "$c"; else echo "$c" >> $out/counters_missing.txt; fi; done; }
: > $out/counters_missing.txt
P1=$(pick SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES)
P2=$(pick SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE)
echo "pass 1: $P1" > $out/passes.txt; echo "pass 2: $P2" >> $out/passes.txt
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $out/p1 -o a -- python3 tools/attn_bench.py > $out/bench1.log 2> $out/err1.log
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $out/p2 -o a -- python3 tools/attn_bench.py > $out/bench2.log 2> $out/err2.log
python3 tools/attn_bench.py > $out/bench_plain.log 2>&1
python3 - "$out" <<'PY'
import collections, csv, glob, json, re, sys
out = sys.argv[1]
shapes = collections.OrderedDict()
for p in ("p1", "p2"):
    f = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "attn_kernel" not in r["Kernel_Name"]:
            continue
        key = (re.sub(r"^.*attn_kernel", "attn_kernel", r["Kernel_Name"])[:40], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in agg.items():
        e = shapes.setdefault(" ".join(key), {})
        for k, v in cs.items():
            e[k] = sum(v) / len(v)
for k, e in shapes.items():
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; MFMA busy counts cycles per SIMD
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM"):
            if c in e:
                e[c + "_per_wave_cycle"] = round(e[c] / wc, 4)
    if e.get("SQ_INSTS_VALU_MFMA_MOPS_F16") and e.get("SQ_INSTS_VALU"):
        # MOPS counts 512 FLOP units: one v_mfma_f32_32x32x16_f16 = 32768 FLOP = 64 units; plain VALU instructions = INSTS_VALU - MFMAs
        mfma = e["SQ_INSTS_VALU_MFMA_MOPS_F16"] / 64.0
        e["mfma_instructions_32x32x16_equiv"] = round(mfma)
        e["valu_instructions_per_mfma"] = round((e["SQ_INSTS_VALU"] - mfma) / mfma, 2) if mfma else None
        if e.get("SQ_INSTS_LDS"):
            e["lds_instructions_per_mfma"] = round(e["SQ_INSTS_LDS"] / mfma, 2)
    if e.get("SQ_VALU_MFMA_BUSY_CYCLES") and e.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_frac_of_chip"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
res = {"note": "attention launches of tools/attn_bench.py (the UNet's six attention shapes, B = 2, H = 8) under two rocprofv3 --pmc passes; "
               "counter values are means over the launches of a shape; *_per_wave_cycle = share of the waves' resident time",
       "missing_counters": [l.strip() for l in open(f"{out}/counters_missing.txt") if l.strip()],
       "timing_unprofiled": [l.strip() for l in open(f"{out}/bench_plain.log") if l.startswith("d=")],
       "shapes": shapes}
json.dump(res, open(f"{out}/attn_instmix.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
rm -rf $out/p1 $out/p2
