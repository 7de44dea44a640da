#!/bin/bash
# round 5, GPU call 5: GPU suite at HEAD; same-box A/B by debug flags (the product library reads no environment variable):
# 0 = default, 262144 = 8x8-level conv rule off, 131072 = row-major tile walk, 524288 / 1048576 = single-stream launch policy from up block 1 / 2
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/gputests5.log 2>&1; echo "pytest rc $?"; tail -14 $OUT/gputests5.log
val() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], round(d['value'],2))" $1; }
for r in 1 2 3; do
  for f in 0 262144 131072 524288 1048576; do
    timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-profile --no-check --debug-flags $f > $OUT/ab_flags_${f}_$r.json 2>/dev/null; val $OUT/ab_flags_${f}_$r.json
  done
done
