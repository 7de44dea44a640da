#!/bin/bash
# round 5, last GPU call: after the last kernel-side edit (wave-uniform loop of skinny_mfma_kernel) -- the tests that cover it, the whole
# GPU suite, and stage A of the evidence again (the PMC traffic summary is stamped with the kernel source hash)
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/gputests_final.log 2>&1; echo "pytest rc $?"; tail -14 $OUT/gputests_final.log
bash tools/collect_profiles.sh r05 A > gpurun_out/r05_collect_A.log 2>&1; echo "collect rc $?"; grep -v "^-rw\|^total\|^drwx" gpurun_out/r05_collect_A.log | cut -c1-300 | tail -12
