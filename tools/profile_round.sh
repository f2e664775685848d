#!/bin/bash
# Final-form profiles of one round: rocprofv3 kernel stats + per-step summary + HBM traffic (separate PMC passes).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01z
set -e
TAG=${1:-r01z}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/bench.py --steps 10 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/fetch.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $ROOT/bench.py --steps 10 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/write.log
cd $ROOT
python3 tools/trace_step.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_step_kernel_summary.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv
python3 tools/pmc_traffic.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/${TAG}_hbm_traffic.json > /dev/null
tail -1 $OUT/bench_under_rocprof.json > $OUT/${TAG}_bench_under_rocprof.json
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/bench_under_rocprof.json
head -12 $OUT/${TAG}_step_kernel_summary.txt
