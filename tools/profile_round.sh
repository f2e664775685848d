#!/bin/bash
# One-call profiling recipe of a round on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh r04
#   kernel trace + launch log -> per-shape times; separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy | MFMA ops)
# Counter passes carry only --kernel-trace besides --pmc (gpurun refuses other trace domains with counters).
set -e
tag=${1:-r05}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
export SDMI_LAUNCH_LOG=$PWD/$out/launch_log.txt
B="python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-image-latency --no-throughput --no-accurate"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/bench_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- $B > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- $B > /dev/null 2> $out/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/mfma -o m -- $B > /dev/null 2> $out/mfma.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/mops -o o -- $B > /dev/null 2> $out/mops.err
find $out -name "*.csv" | head -20
python3 tools/join_trace.py time $(find $out/trace -name "*kernel_trace.csv") $out/launch_log.txt --json $out/step_families.json > $out/step_by_shape.txt
python3 tools/join_trace.py pmc $(find $out/fetch -name "*counter_collection.csv") $(find $out/write -name "*counter_collection.csv") $out/launch_log.txt $out/hbm_traffic_by_shape.json > $out/pmc_summary.txt
python3 tools/join_trace.py mfma $(find $out/mfma -name "*counter_collection.csv") $out/launch_log.txt $out/mfma_busy.json $out/step_families.json > $out/mfma_summary.txt
python3 tools/join_trace.py mfma $(find $out/mops -name "*counter_collection.csv") $out/launch_log.txt $out/mfma_ops.json > $out/mops_summary.txt
cp $(find $out/trace -name "*kernel_stats.csv") $out/kernel_stats.csv
# regression gate: every logical shape of this step against the last COMMITTED round's profile; a shape more than 5 % slower per
# launch fails the script unless profiles/regression_allow.txt names it with a reason (SDMI_NO_REGRESSION_GATE=1: report only)
prev=$(ls profiles/r[0-9][0-9]_step_by_shape.txt 2>/dev/null | grep -v "/${tag}_" | sort | tail -n 1)
gate_rc=0
if [ -n "$prev" ]; then
  python3 tools/join_trace.py compare $prev $out/step_by_shape.txt --allow profiles/regression_allow.txt > $out/regression_vs_$(basename $prev) || gate_rc=$?
  tail -n 12 $out/regression_vs_$(basename $prev)
fi
python3 - "$out" <<'PY'
import json, sys
out = sys.argv[1]
b = json.loads(open(f"{out}/bench_trace.json").read().strip().splitlines()[-1])
n = b["config"]["launches_per_step"]
lib_hash = b["roofline"]["lib_hash"]          # FNV-1a 64 of the libsdmi.so these passes ran (bench.py quotes a profile only for that binary)
for f in ("hbm_traffic_by_shape.json", "mfma_busy.json", "mfma_ops.json", "step_families.json"):
    j = json.load(open(f"{out}/{f}"))
    j["lib_hash"] = lib_hash
    j["bench_launches_per_step"] = n
    j["bench_command"] = "python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-image-latency --no-throughput --no-accurate (under rocprofv3 --pmc ...)"
    j["family_definition"] = "mfma = igemm_kernel + conv3_halo_kernel + b2b_kernel; finalize = splitk_finalize; attn; norm = GroupNorm / LayerNorm kernels; other"
    json.dump(j, open(f"{out}/{f}", "w"), indent=1)
print("launches/step", n)
PY
# the raw CSVs are large: keep only the summaries (the merge back is capped at 64 MiB)
rm -rf $out/trace $out/fetch $out/write $out/mfma $out/mops
# what the round commits: gpurun_out/<tag>/p/<tag>_* -> copy to profiles/ after the call
mkdir -p $out/p
cp $out/step_by_shape.txt $out/p/${tag}_step_by_shape.txt
cp $out/step_families.json $out/p/${tag}_step_families.json
cp $out/kernel_stats.csv $out/p/${tag}_rocprofv3_kernel_stats.csv
cp $out/hbm_traffic_by_shape.json $out/p/${tag}_hbm_traffic_by_shape.json
cp $out/mfma_busy.json $out/p/${tag}_mfma_busy.json
cp $out/mfma_ops.json $out/p/${tag}_mfma_ops.json
cp $out/launch_log.txt $out/p/${tag}_launch_log.txt
cp $out/bench_trace.json $out/p/${tag}_bench_under_trace.json
cp $out/regression_vs_* $out/p/ 2>/dev/null || true
ls -la $out
if [ "$gate_rc" != "0" ] && [ -z "$SDMI_NO_REGRESSION_GATE" ]; then echo "profile_round: per-shape regression gate FAILED (see $out/regression_vs_*)"; exit 3; fi
