#!/bin/bash
# round 5, final GPU call(s): the whole GPU suite at HEAD, then the evidence of tools/collect_profiles.sh (stages given as $1)
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [[ "$1" == *T* ]]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/gputests_final.log 2>&1; echo "pytest rc $?"; tail -14 $OUT/gputests_final.log
fi
S=${1//T/}
if [[ -n "$S" ]]; then bash tools/collect_profiles.sh r05 $S > gpurun_out/r05_collect_$S.log 2>&1; echo "collect rc $?"; grep -v "^-rw\|^total\|^drwx" gpurun_out/r05_collect_$S.log | cut -c1-300 | tail -40; fi
if [[ "$1" == *B* ]]; then
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --encoder-weights frozen-copy > $OUT/bench_cfg4_frozen_copy_encoder.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_cfg4_frozen_copy_encoder.json').read().strip().splitlines()[-1]); print('frozen-copy encoder', d['ms_per_step'], d['value'], d['weight_bytes_bf16_packed'], d['config']['encoder_weights_shared'])"
fi
