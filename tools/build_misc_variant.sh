#!/bin/bash
# Probe library that differs from the product library in misc.hip only:  tools/build_misc_variant.sh <tag> [flags]
# NOTE: misc.hip is compiled here WITH packed fp32 selection (the product build turns it off, mvd_amd/_build.py NO_PACKED_FP32):
# that is what reproduces the four-pixel conv_out's run-to-run differences; add the product's flags to get the stable build.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p mvd_amd/csrc/build_$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMVD_PROBE "$@" -c mvd_amd/csrc/misc.hip -o mvd_amd/csrc/build_$tag/misc.o
objs=$(ls mvd_amd/csrc/build/*.o | grep -v "/misc.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mvd_amd/libmvd_hip_$tag.so $objs mvd_amd/csrc/build_$tag/misc.o
echo mvd_amd/libmvd_hip_$tag.so
