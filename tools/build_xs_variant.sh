#!/bin/bash
# A/B or ablation build that differs from the product library in gemm_xs.hip only:
#   tools/build_xs_variant.sh <tag> [-D...]  ->  mvd_amd/libmvd_hip_<tag>.so   (other objects come from mvd_amd/csrc/build)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p mvd_amd/csrc/build_$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c mvd_amd/csrc/gemm_xs.hip -o mvd_amd/csrc/build_$tag/gemm_xs.o
objs=$(ls mvd_amd/csrc/build/*.o | grep -v gemm_xs.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mvd_amd/libmvd_hip_$tag.so $objs mvd_amd/csrc/build_$tag/gemm_xs.o
echo mvd_amd/libmvd_hip_$tag.so
