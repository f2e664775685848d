#!/bin/bash
# workgroup life-line probes (diagnostic build: tools/build_variant.sh probe gemm -DSDMI_CLK_PROBE) of the halo conv kernels
export SDMI_LIB=$PWD/pytorch_stable_diffusion_amd/lib/variants/libsdmi_probe.so
echo "--- 8192x320x5760 (64x64, Cin 640), split 2"
PROBE_KS=3 PROBE_H=64 PROBE_SPLIT=2 python tools/phase_probe.py 8192 320 5760 h128x160s3 k128x160s3 k128x160s3p5 k128x160s3w8p5
echo "--- 8192x320x2880 (64x64, Cin 320), split 1"
PROBE_KS=3 PROBE_H=64 PROBE_SPLIT=1 python tools/phase_probe.py 8192 320 2880 h128x128s3 k128x128s3 k128x128s3p8
echo "--- 128x1280x11520 (8x8), split 10"
PROBE_KS=3 PROBE_H=8 PROBE_SPLIT=10 python tools/phase_probe.py 128 1280 11520 k128x64s4 k128x64s4p8 h64x128s3
