#!/bin/bash
export SDMI_LIB=$PWD/pytorch_stable_diffusion_amd/lib/variants/libsdmi_probe.so
PROBE_M=512 PROBE_N=1280 PROBE_K=1280 PROBE_KS=1 PROBE_H=1 PROBE_BLOCKS=160 python tools/clk_probe.py t64x64s4p t64x64s4q2 t64x64s4 t64x64s2p4
PROBE_M=128 PROBE_N=1280 PROBE_K=1280 PROBE_KS=1 PROBE_H=1 PROBE_BLOCKS=40 python tools/clk_probe.py t64x64s4p
