#!/bin/bash
# same-box A/B of two library builds under one probe script: tools/ab_probe.sh <probe.py> <libA> <libB> [rounds]
P=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    echo "== $L (round $r)"
    MVD_HIP_LIB=$PWD/$L timeout -k 10 300 python $P 2>&1 | grep -v amdgpu.ids
  done
done
