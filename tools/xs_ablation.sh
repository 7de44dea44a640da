# same-box A/B of gemm_xs variants: tools/xs_ablation.sh <shapes> <csplits> lib1 lib2 ...
SH=$1; CS=$2; shift 2
for r in 1 2; do
for L in "$@"; do
  echo "== $L (round $r)"
  MVD_HIP_LIB=$PWD/mvd_amd/$L XS_ONLY=1 XS_SHAPES=$SH XS_CSPLIT=$CS timeout -k 10 120 python tools/probe_xs.py 2>&1 | grep -v amdgpu.ids
done; done
