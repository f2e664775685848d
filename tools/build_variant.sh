#!/bin/bash
# Diagnostic builds of libsdmi.so with extra -D flags on gemm.hip only (A/B timing through SDMI_LIB=<path>).
# usage: tools/build_variant.sh <name> <source stem: gemm|attention|norm|...> <flags...>
#        -> pytorch_stable_diffusion_amd/lib/variants/libsdmi_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift; shift
pkg=pytorch_stable_diffusion_amd
mkdir -p $pkg/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form "$@" -c $pkg/csrc/$src.hip -o $pkg/lib/variants/${src}_$name.o
objs=$(ls $pkg/lib/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/lib/variants/libsdmi_$name.so $pkg/lib/variants/${src}_$name.o $objs
echo $pkg/lib/variants/libsdmi_$name.so
