set -e
mkdir -p gpurun_out/r5v; rm -f gpurun_out/r5v/*.log
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "groupnorm or gn_" > gpurun_out/r5v/tests.log 2>&1
SDMI_GNA_IT=4 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "groupnorm or gn_" > gpurun_out/r5v/tests_it4.log 2>&1
B="python bench.py --steps 100 --warmup 20 --no-throughput --no-accurate --no-cpu-baseline --no-image-latency"
for i in 1 2 3; do
SDMI_LIB=pytorch_stable_diffusion_amd/lib/variants/libsdmi_oldnorm.so timeout -k 10 200 $B 2>/dev/null | tail -1 >> gpurun_out/r5v/old.log
timeout -k 10 200 $B 2>/dev/null | tail -1 >> gpurun_out/r5v/new.log
done
tools/trace_step.sh r5v/trace > gpurun_out/r5v/trace.log 2>&1
echo done
