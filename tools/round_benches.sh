#!/bin/bash
# The bench lines a round commits (run on the GPU box from the repo root AFTER tools/profile_round.sh <tag> and after its
# gpurun_out/<tag>/p/* were copied to profiles/: the default line then quotes that profile's traffic / MFMA / trace figures).
# usage: tools/round_benches.sh r04      -> gpurun_out/<tag>_benches/<tag>_*.json
tag=${1:-r05}
out=gpurun_out/${tag}_benches
mkdir -p $out
export TMPDIR=/tmp
last() { tail -n 1 "$1" > "$2"; }
set -x
python3 bench.py --gpus 1 --steps 50 --warmup 10 > $out/full.log 2> $out/full.err && last $out/full.log $out/${tag}_bench.json
python3 bench.py --latent 96 --steps 20 --warmup 5 --no-cpu-baseline --no-image-latency --no-accurate > $out/b768.log 2> $out/b768.err && last $out/b768.log $out/${tag}_bench_768.json
python3 bench.py --chains 3 --steps 20 --warmup 5 --no-cpu-baseline --no-image-latency --no-accurate > $out/c3.log 2> $out/c3.err && last $out/c3.log $out/${tag}_bench_chains3.json
for P in 2 4 6 8; do
  python3 bench.py --batch-prompts $P --steps 20 --warmup 5 --no-cpu-baseline --no-image-latency --no-accurate > $out/p$P.log 2> $out/p$P.err && last $out/p$P.log $out/${tag}_bench_batched_p$P.json
done
python3 bench.py --chains 2 --batch-prompts 6 --steps 20 --warmup 5 --no-cpu-baseline --no-image-latency --no-accurate > $out/c2p6.log 2> $out/c2p6.err && last $out/c2p6.log $out/${tag}_bench_lanes2_batched6.json
# two ranks on the one GPU over gloo (rehearsal of the N > 1 path of bench.py: barriers, max over ranks, the self-check fields)
SDMI_BENCH_ONE_DEVICE=1 SDMI_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-image-latency --no-accurate > $out/r2.log 2> $out/r2.err && last $out/r2.log $out/${tag}_bench_2rank_rehearsal.json
set +x
tools/trace_batched.sh ${tag}_benches/trace_p4 4 > $out/trace_p4.log 2>&1 && cp $out/trace_p4/step_by_shape.txt $out/${tag}_step_by_shape_batched_p4.txt
ls -la $out
