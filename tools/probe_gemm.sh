#!/bin/bash
# workgroup life-lines of the K = C GEMMs (diagnostic build: tools/build_variant.sh probe gemm -DSDMI_CLK_PROBE)
export SDMI_LIB=$PWD/pytorch_stable_diffusion_amd/lib/variants/libsdmi_probe.so
python tools/phase_probe.py 512 1280 1280 t64x64s4p t64x64s4q2
python tools/phase_probe.py 2048 640 640 t64x64s4p
python tools/phase_probe.py 128 1280 1280 t64x64s4q2
