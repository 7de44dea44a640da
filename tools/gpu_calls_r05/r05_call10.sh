#!/bin/bash
# round 5, GPU call 10: the latency-bound front matter (camera / time MLPs of both passes) BEFORE the side stream forks
# (debug flag 4194304 = the old order: fork first) -- engine tests, same-box A/B at cfg4 / cfg3, the small kernels' in-situ durations
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_pipeline_gpu.py tests/test_global_stats_gpu.py tests/test_fixtures_gpu.py -m gpu -x -q > $OUT/gputests10.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/gputests10.log
val() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], round(d['value'],2), d.get('output_check'))" $1; }
for r in 1 2 3; do
  for f in 0 4194304; do
    timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-profile --debug-flags $f > $OUT/ab_fork_cfg4_${f}_$r.json 2>/dev/null; val $OUT/ab_fork_cfg4_${f}_$r.json
  done
done
for r in 1 2; do
  for f in 0 4194304; do
    timeout -k 10 300 python bench.py --workload cfg3 --steps 60 --warmup 5 --no-cpu-baseline --no-profile --debug-flags $f > $OUT/ab_fork_cfg3_${f}_$r.json 2>/dev/null; val $OUT/ab_fork_cfg3_${f}_$r.json
  done
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof4 -o stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-profile > $OUT/fork_stats_bench.json 2> $OUT/fork_stats.err
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["OUT"] + "/prof4/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if any(k in n for k in ("skinny", "ln_f32", "film", "timestep", "silu_to", "camera_features")):
        print(f"{n[:70]:72s} calls/fw {int(r['Calls']) / 12:5.1f}  avg {float(r['AverageNs']) / 1e3:7.1f} us")
PY
rm -rf $OUT/prof4
