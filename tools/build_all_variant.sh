#!/bin/bash
# Diagnostic build of the WHOLE library with extra -D flags (A/B timing through SDMI_LIB=<path>).
# usage: tools/build_all_variant.sh <name> <flags...>  -> pytorch_stable_diffusion_amd/lib/variants/libsdmi_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
pkg=pytorch_stable_diffusion_amd
d=$pkg/lib/variants/$name
mkdir -p $d
for src in $pkg/csrc/*.hip; do
  b=$(basename $src .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -Iinclude "$@" -c $src -o $d/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/lib/variants/libsdmi_$name.so $d/*.o
echo $pkg/lib/variants/libsdmi_$name.so
