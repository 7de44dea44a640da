#!/bin/bash
# round 5, GPU call 22: small-M kernels up to M = 9216 (one image's 96x96 map): A/B at 768x768 batch 1 (flag 16777216 = the old limit 4608)
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
val() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], round(d['value'],2), d.get('output_check'))" $1; }
for r in 1 2; do
  for w in cfg3 cfg2; do
    for f in 0 16777216; do
      timeout -k 10 300 python bench.py --workload $w --latent 96 --steps 40 --warmup 5 --no-cpu-baseline --no-profile --debug-flags $f > $OUT/ab_sm9216_${w}_${f}_$r.json 2>/dev/null; val $OUT/ab_sm9216_${w}_${f}_$r.json
    done
  done
done
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "768 or two_stream" 2>&1 | tail -2
