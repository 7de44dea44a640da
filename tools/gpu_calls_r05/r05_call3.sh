#!/bin/bash
# round 5, GPU call 3: conv_ws on 12/24-wide maps (tests + 768x768 A/B), the 8x8-level conv rule (same-box A/B), attention overlap trace
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests/test_conv_ws_gpu.py tests/test_cfg4_shapes_gpu.py tests/test_engine_gpu.py -m gpu -x -q > $OUT/gputests3.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/gputests3.log
val() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], round(d['value'],2))" $1; }
for r in 1 2; do
  MVD_GEMM_DEEP_CONV_SPLIT=0 timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-profile > $OUT/ab_deepconv_off_$r.json 2>/dev/null; val $OUT/ab_deepconv_off_$r.json
  timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-profile > $OUT/ab_deepconv_on_$r.json 2>/dev/null; val $OUT/ab_deepconv_on_$r.json
  MVD_GEMM_PP_WALK=0 timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-profile > $OUT/ab_walk_off_$r.json 2>/dev/null; val $OUT/ab_walk_off_$r.json
done
for w in cfg2 cfg3; do
  timeout -k 10 300 python bench.py --workload $w --latent 96 --steps 30 --warmup 5 --no-cpu-baseline --no-profile --debug-flags 256 > $OUT/bench_${w}_latent96_ws_off.json 2>/dev/null; val $OUT/bench_${w}_latent96_ws_off.json
  timeout -k 10 300 python bench.py --workload $w --latent 96 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_${w}_latent96.json 2>/dev/null; val $OUT/bench_${w}_latent96.json
done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_cfg4 -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-check > $OUT/trace_bench.json 2> $OUT/trace_bench.err; echo "trace rc $?"
python tools/overlap_from_trace.py $OUT/trace_cfg4 > $OUT/attention_overlap.txt 2>&1; cat $OUT/attention_overlap.txt
rm -rf $OUT/trace_cfg4
