#!/bin/bash
# round 5, GPU call 11: the camera / time MLP layers on the fp32 matrix pipe (debug flag 8388608 = the vector form): op tests, fixture and
# engine tests, same-box A/B at cfg4 / cfg3 / cfg2 (also with the front matter in front of the fork: 4194304), in-situ durations
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fixtures_gpu.py tests/test_engine_gpu.py tests/test_pipeline_gpu.py tests/test_global_stats_gpu.py -m gpu -x -q -k "skinny or fixture or camera or tiny or sd21_full_size_parity or deterministic or pipeline or global" > $OUT/gputests11.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/gputests11.log
val() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], round(d['value'],2), d.get('output_check'))" $1; }
for r in 1 2 3; do
  for f in 0 8388608 4194304; do
    timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-profile --debug-flags $f > $OUT/ab_skinny_cfg4_${f}_$r.json 2>/dev/null; val $OUT/ab_skinny_cfg4_${f}_$r.json
  done
done
for r in 1 2; do
  for w in cfg3 cfg2; do
    for f in 0 8388608; do
      timeout -k 10 300 python bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --debug-flags $f > $OUT/ab_skinny_${w}_${f}_$r.json 2>/dev/null; val $OUT/ab_skinny_${w}_${f}_$r.json
    done
  done
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof4 -o stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-profile > $OUT/skinny_stats_bench.json 2> $OUT/skinny_stats.err
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["OUT"] + "/prof4/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if any(k in n for k in ("skinny", "ln_f32", "film_params", "timestep", "silu_to", "camera_features")):
        print(f"{n[:70]:72s} calls/fw {int(r['Calls']) / 12:5.1f}  avg {float(r['AverageNs']) / 1e3:7.1f} us")
PY
rm -rf $OUT/prof4
