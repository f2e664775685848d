mkdir -p gpurun_out/r5y
SDMI_LIB=pytorch_stable_diffusion_amd/lib/variants/libsdmi_probe.so timeout -k 10 200 python tools/gna_clk_probe.py > gpurun_out/r5y/probe.log 2>&1
echo done
