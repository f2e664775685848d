#!/bin/bash
mkdir -p gpurun_out/r3o
V=pytorch_stable_diffusion_amd/lib/variants
for m in 2048 100000; do
  echo "### SPLIT8_MINS=$m base"; SDMI_ATTN_SPLIT8_MINS=$m python tools/attn_bench.py 2>&1 | grep "Sq=4096 Skv=4096"
  for v in 1 2 4 8 24 7 31; do
    echo "### SPLIT8_MINS=$m ablate=$v"; SDMI_ATTN_SPLIT8_MINS=$m SDMI_LIB=$PWD/$V/libsdmi_ab$v.so python tools/attn_bench.py 2>&1 | grep "Sq=4096 Skv=4096"
  done
done
