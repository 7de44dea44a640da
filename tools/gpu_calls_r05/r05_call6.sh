#!/bin/bash
# round 5, GPU call 6: GroupNorm one-pass form for one image's 64x64 map (flag 2097152 = the round-4 rule), batch-1 A/B + kernel counts
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
val() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], round(d['value'],2), d.get('output_check'))" $1; }
for r in 1 2 3; do
  for w in cfg2 cfg3; do
    for f in 0 2097152; do
      timeout -k 10 300 python bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --debug-flags $f > $OUT/ab_gn_${w}_${f}_$r.json 2>/dev/null; val $OUT/ab_gn_${w}_${f}_$r.json
    done
  done
done
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "groupnorm or sd21_full_size_parity or deterministic" > $OUT/gputests6.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/gputests6.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof2 -o stats -- python3 bench.py --workload cfg2 --steps 25 --warmup 1 --no-cpu-baseline --no-check --no-profile > $OUT/bench_cfg2_under_rocprof.json 2> $OUT/rocprof2.err
cp $(find $OUT/prof2 -name '*kernel_stats.csv' | head -1) $OUT/bench_cfg2_kernel_stats.csv; rm -rf $OUT/prof2
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["OUT"] + "/bench_cfg2_kernel_stats.csv")))
eng = [r for r in rows if not r["Name"].startswith(("at::", "__amd", "void at::")) and "elementwise" not in r["Name"] and "at::native" not in r["Name"]]
n = sum(int(r["Calls"]) for r in eng)
print("engine kernels over 26 forwards:", n, "=", round(n / 26, 1), "per forward;", round(sum(float(r["TotalDurationNs"]) for r in eng) / 26e6, 3), "ms of kernel time per forward")
PY
