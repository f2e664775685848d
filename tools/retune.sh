#!/bin/bash
# Re-tunes every GEMM shape the bench / tests touch (cold-L2 timing, csrc/engine.h tune_gemm) and collects the plan lines.
# usage (on the GPU box): tools/retune.sh <out dir> [bench|tests]   -> <out dir>/plans_new.txt ; merge with tools/merge_plans.py
set -e
out=$1; what=${2:-bench}
mkdir -p $out/pc
export SDMI_RETUNE=1 SDMI_PLAN_CACHE_DIR=$PWD/$out/pc
if [ "$what" = bench ]; then
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-throughput --no-accurate > $out/bench64.json
  echo "bench 64 done"; wc -l $out/pc/*.txt
  python bench.py --latent 96 --steps 10 --warmup 3 --no-cpu-baseline --no-image-latency > $out/bench96.json
  echo "bench 96 done"; wc -l $out/pc/*.txt
else
  python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || true
  tail -3 $out/tests.log; wc -l $out/pc/*.txt
fi
cat $out/pc/plans-*.txt > $out/plans_new_$what.txt
