#!/bin/bash
# Diagnostic builds of libsdmi.so with extra -D flags on gemm.hip only (A/B timing through SDMI_LIB=<path>).
# usage: tools/build_variant.sh <name> <flags...>   -> pytorch_stable_diffusion_amd/lib/variants/libsdmi_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
pkg=pytorch_stable_diffusion_amd
mkdir -p $pkg/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form "$@" -c $pkg/csrc/gemm.hip -o $pkg/lib/variants/gemm_$name.o
objs=$(ls $pkg/lib/obj/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/lib/variants/libsdmi_$name.so $pkg/lib/variants/gemm_$name.o $objs
echo $pkg/lib/variants/libsdmi_$name.so
