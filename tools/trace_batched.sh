#!/bin/bash
# kernel trace of the BATCHED denoising step (bench.py --batch-prompts P) joined with the engine's launch log -> per-shape times
# usage: tools/trace_batched.sh <tag> <P> [ENV=VAL ...]      (run on the GPU box from the repo root)
set -e
tag=$1; P=$2; shift; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export SDMI_LAUNCH_LOG=$out/launch_log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-image-latency --no-accurate --batch-prompts $P > $out/bench_trace.json 2> $out/trace.err
python3 tools/join_trace.py time $(find $out/trace -name "*kernel_trace.csv") $out/launch_log.txt -1 > $out/step_by_shape.txt
rm -rf $out/trace
head -8 $out/step_by_shape.txt
