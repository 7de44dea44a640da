#!/bin/bash
# round 5, GPU call 7: timing-only ablation of the ping-pong convolution's A-operand DMAs (one tap in nine fetched): the upper bound of
# what an LDS-resident activation patch could save
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
bash tools/ab_probe.sh tools/probe_r05_conv_ablation.py mvd_amd/libmvd_hip.so mvd_amd/libmvd_hip_ablate_a.so 2 > $OUT/probe_conv_ablation.log 2>&1; cat $OUT/probe_conv_ablation.log
