#!/bin/bash
# same-box A/B of two builds of libmvd_hip.so: tools/ab_bench.sh <libA> <libB> [rounds] [bench args...]
A=$1; B=$2; R=${3:-2}; shift 3
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    MVD_HIP_LIB=$PWD/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'])"
  done
done
