#!/bin/bash
# round 5, GPU call 4: two-stream kernel trace -> attention overlap by shape + what the main queue runs alone; full GPU suite at HEAD
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_cfg4 -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-check > $OUT/trace_bench.json 2> $OUT/trace_bench.err; echo "trace rc $?"
head -2 $(find $OUT/trace_cfg4 -name "*kernel_trace.csv" | head -1) | cut -c1-600
python tools/overlap_from_trace.py $OUT/trace_cfg4 > $OUT/attention_overlap.txt 2>&1; cat $OUT/attention_overlap.txt
rm -rf $OUT/trace_cfg4
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/gputests4.log 2>&1; echo "pytest rc $?"; tail -14 $OUT/gputests4.log
