#!/bin/bash
# Probe library that differs from the product library in gemm_pp.hip only:  tools/build_pp_variant.sh <tag> [flags]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p mvd_amd/csrc/build_$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c mvd_amd/csrc/gemm_pp.hip -o mvd_amd/csrc/build_$tag/gemm_pp.o
objs=$(ls mvd_amd/csrc/build/*.o | grep -v "/gemm_pp.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mvd_amd/libmvd_hip_$tag.so $objs mvd_amd/csrc/build_$tag/gemm_pp.o
echo mvd_amd/libmvd_hip_$tag.so
