#!/bin/bash
# workgroup life-line probes (diagnostic build: tools/build_variant.sh probe gemm -DSDMI_CLK_PROBE) of the halo conv kernels
export SDMI_LIB=$PWD/pytorch_stable_diffusion_amd/lib/variants/libsdmi_probe.so
echo "--- 8192x320x5760 (64x64, Cin 640), split 2"
PROBE_KS=3 PROBE_H=64 PROBE_SPLIT=2 python tools/phase_probe.py 8192 320 5760 h128x160s3 h256x128s3
echo "--- 8192x320x2880 (64x64, Cin 320), split 1"
PROBE_KS=3 PROBE_H=64 PROBE_SPLIT=1 python tools/phase_probe.py 8192 320 2880 h128x128s3 h128x160s3
echo "--- 128x1280x11520 (8x8), split 6"
PROBE_KS=3 PROBE_H=8 PROBE_SPLIT=6 python tools/phase_probe.py 128 1280 11520 h64x128s3 h128x128s4
