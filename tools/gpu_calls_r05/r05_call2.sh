#!/bin/bash
# round 5, GPU call 2: Winograd core with perfect L2 locality; tile-walk / deep-conv A/B; graph capture with the side stream; tests
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_wino_core tools/probe_wino_core.hip 2> $OUT/wino_build.err || { echo "wino build failed"; exit 1; }
timeout -k 10 300 /tmp/probe_wino_core > $OUT/probe_wino_core_v2.log 2>&1 || { echo "wino probe failed"; exit 1; }
tail -5 $OUT/probe_wino_core_v2.log
timeout -k 10 600 python tools/probe_r05_walk.py > $OUT/probe_walk.log 2>&1; echo "walk probe rc $?"; cat $OUT/probe_walk.log
for CS in FETCH_SIZE WRITE_SIZE; do
  PMC_MODE=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $CS -d $OUT/walk_pmc_$CS -o s -- python3 tools/probe_r05_walk.py > /dev/null 2> $OUT/walk_pmc_$CS.err || echo "pmc pass $CS failed"
done
python tools/pmc_ops_summary.py $OUT/walk_pmc_FETCH_SIZE $OUT/walk_pmc_WRITE_SIZE > $OUT/probe_walk_pmc.txt 2>&1; cat $OUT/probe_walk_pmc.txt
rm -rf $OUT/walk_pmc_FETCH_SIZE $OUT/walk_pmc_WRITE_SIZE
timeout -k 10 900 python -m pytest tests/test_cfg4_shapes_gpu.py tests/test_ops_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > $OUT/gputests2.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/gputests2.log
for w in cfg2 cfg3; do
  timeout -k 10 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_${w}.json 2> $OUT/bench_${w}.err; echo "bench $w rc $?"
  timeout -k 10 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --graph > $OUT/bench_${w}_graph.json 2> $OUT/bench_${w}_graph.err; echo "bench $w graph rc $?"
done
timeout -k 10 300 python bench.py --workload cfg3 --steps 50 --warmup 5 --no-cpu-baseline --graph --debug-flags 65536 > $OUT/bench_cfg3_graph_one_stream.json 2> $OUT/bench_cfg3_graph_one_stream.err
timeout -k 10 300 python bench.py --workload cfg3 --latent 96 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_latent96.json 2>/dev/null
timeout -k 10 300 python bench.py --workload cfg3 --latent 96 --steps 30 --warmup 5 --no-cpu-baseline --graph > $OUT/bench_cfg3_latent96_graph.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["OUT"] + "/bench_cfg[23]*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["value"], d.get("output_check"))
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
timeout -k 10 600 python bench.py --no-cpu-baseline > $OUT/bench_cfg4_b.json 2> $OUT/bench_cfg4_b.err; echo "bench rc $?"; python -c "
import json,os; d=json.loads(open(os.environ['OUT']+'/bench_cfg4_b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print({k:(v['ms_per_step'],v['tflops']) for k,v in d['kernel_classes'].items()})"
