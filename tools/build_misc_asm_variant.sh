#!/bin/bash
# Probe library whose misc.hip DEVICE code goes through an assembly edit before it is assembled:
#   tools/build_misc_asm_variant.sh <tag> <sed -E script applied inside conv_out4_kernel | py:tools/asm_edits/x.py> [hipcc flags]
# Used for the conv_out four-pixel diagnosis: e.g. insert wait states behind every v_pk_mov_b32 of the reverted
# kernel, or rewrite its op_sel forms, and see whether the run-to-run differences of the two-rank rehearsal go away.
set -e
cd "$(dirname "$0")/.."
tag=$1; edit=$2; shift 2
B=mvd_amd/csrc/build_$tag; mkdir -p $B
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMVD_PROBE $*"
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S mvd_amd/csrc/misc.hip -o $B/misc_dev.s
python3 tools/asm_edit_kernel.py $B/misc_dev.s conv_out4_kernel "$edit"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $B/misc_dev.s -o $B/misc_dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B/misc_dev.hsaco $B/misc_dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$B/misc_dev.hsaco -output=$B/misc_dev.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $B/misc_dev.hipfb -c mvd_amd/csrc/misc.hip -o $B/misc.o
objs=$(ls mvd_amd/csrc/build/*.o | grep -v "/misc.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mvd_amd/libmvd_hip_$tag.so $objs $B/misc.o
echo mvd_amd/libmvd_hip_$tag.so
