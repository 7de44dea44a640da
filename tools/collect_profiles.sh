#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: the judged evidence of one round.
#   tools/collect_profiles.sh <round-tag, e.g. r03>
# Writes under gpurun_out/<tag>/: bench lines of every configuration (+ per-shape tables), the rocprofv3 kernel-trace stats of
# the cfg4 and cfg2 commands, two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) summarised by tools/pmc_summary.py, one
# GRBM_GUI_ACTIVE pass (shader clock under load, tools/clock_summary.py) and the SQ-counter passes on isolated launches.
set -e -o pipefail
TAG=${1:-r04}
STAGE=${2:-ABCD}         # A: PMC traffic + the headline line + rocprof stats + sustained + peaked; B: batch 1, 768^2, e2e; C: SQ counters;
                         # D: shared-GPU determinism (operator forms beside a second process; the two-rank rehearsal of bench.py)
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [[ $STAGE == *A* ]]; then
# the PMC traffic passes FIRST: bench.py quotes roofline.traffic from profiles/<tag>_pmc_hbm_traffic.json (stamped with the kernel
# source hash), so the file has to be in place -- in this box's copy of the tree -- before the bench line is produced
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2> $OUT/pmc_write.err
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_hbm_traffic.json > $OUT/pmc_hbm_traffic.txt
cp $OUT/pmc_hbm_traffic.json profiles/${TAG}_pmc_hbm_traffic.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_clk -o c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2> $OUT/pmc_clk.err
python tools/clock_summary.py $OUT/pmc_clk $OUT/clock_under_load.json > $OUT/clock_under_load.txt 2> $OUT/clock_summary.err || (echo "clock summary failed"; head -3 $(find $OUT/pmc_clk -name '*counter_collection.csv' | head -1)) > $OUT/clock_under_load.txt
rm -rf $OUT/pmc_clk
echo "--- cfg4 (the headline configuration, with the CPU baseline)"
python bench.py --steps 20 --warmup 5 --shapes-out $OUT/shapes_cfg4.txt > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
cat $OUT/bench_cfg4.json
# rocprofv3 kernel stats of the SAME command in its two launch schedules, each pure (--no-profile --no-check: every forward under the
# profiler is a timed-region forward): two streams (the default; compare with roofline.avg_launch_us_overlapped) and one stream
# (--debug-flags 16; compare with roofline.avg_launch_us) -- tools/frac_from_stats.py recomputes frac from either
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof4 -o stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-profile > $OUT/bench_cfg4_under_rocprof_two_streams.json 2> $OUT/rocprof4.err
cp $(find $OUT/prof4 -name '*kernel_stats.csv' | head -1) $OUT/bench_cfg4_kernel_stats_two_streams.csv
rm -rf $OUT/prof4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof4 -o stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-profile --debug-flags 16 > $OUT/bench_cfg4_under_rocprof_one_stream.json 2> $OUT/rocprof4b.err
cp $(find $OUT/prof4 -name '*kernel_stats.csv' | head -1) $OUT/bench_cfg4_kernel_stats_one_stream.csv
rm -rf $OUT/prof4
python tools/frac_from_stats.py $OUT/bench_cfg4_kernel_stats_two_streams.csv $OUT/bench_cfg4.json two > $OUT/frac_from_stats.txt
python tools/frac_from_stats.py $OUT/bench_cfg4_kernel_stats_one_stream.csv $OUT/bench_cfg4.json one >> $OUT/frac_from_stats.txt
cat $OUT/frac_from_stats.txt
echo "--- sustained: 400 steps"
python bench.py --steps 400 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg4_400steps.json 2> /dev/null
python -c "import json;a=json.load(open('$OUT/bench_cfg4.json'));b=json.load(open('$OUT/bench_cfg4_400steps.json'));print('20 steps', a['value'], '400 steps', b['value'], 'ratio', round(b['value']/a['value'],4))" | tee $OUT/sustained.txt
echo "--- peaked softmaxes (--attn-stats peaked: query projections x 8)"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --attn-stats peaked > $OUT/bench_cfg4_attn_peaked.json 2> /dev/null
python -c "import json;a=json.load(open('$OUT/bench_cfg4.json'));b=json.load(open('$OUT/bench_cfg4_attn_peaked.json'));print('flat', a['value'], a['roofline']['kernel'], a['roofline']['achieved'], a['config']['attn_stats']);print('peaked', b['value'], b['roofline']['kernel'], b['roofline']['achieved'], b['config']['attn_stats'])" | tee $OUT/attn_stats.txt
fi
if [[ $STAGE == *B* ]]; then
echo "--- batch 1"
python bench.py --workload cfg2 --steps 50 --warmup 5 --no-cpu-baseline --shapes-out $OUT/shapes_cfg2.txt > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof2 -o stats -- python3 bench.py --workload cfg2 --steps 25 --warmup 1 --no-cpu-baseline --no-check --no-profile > $OUT/bench_cfg2_under_rocprof.json 2> $OUT/rocprof2.err
cp $(find $OUT/prof2 -name '*kernel_stats.csv' | head -1) $OUT/bench_cfg2_kernel_stats.csv
rm -rf $OUT/prof2
python bench.py --workload cfg3 --steps 50 --warmup 5 --no-cpu-baseline --shapes-out $OUT/shapes_cfg3.txt > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --workload cfg3 --steps 50 --warmup 5 --no-cpu-baseline --debug-flags 16 --no-profile > $OUT/bench_cfg3_one_stream.json 2> /dev/null
python bench.py --workload cfg3 --cached --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_cached.json 2> /dev/null
python bench.py --cached --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg4_cached.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --debug-flags 16 --no-profile > $OUT/bench_cfg4_one_stream.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --debug-flags 32 --no-profile > $OUT/bench_cfg4_two_streams_single_stream_launch_policy.json 2> /dev/null
echo "--- 768 x 768 (96 x 96 latents)"
for w in cfg2 cfg3 cfg4; do python bench.py --workload $w --latent 96 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${w}_latent96.json 2> /dev/null; done
echo "--- seconds per image"
python bench.py --workload e2e --steps 5 --warmup 1 > $OUT/bench_e2e_512.json 2> /dev/null
python bench.py --workload e2e --steps 5 --warmup 1 --cached --no-cpu-baseline > $OUT/bench_e2e_512_cached.json 2> /dev/null
python bench.py --workload e2e --latent 96 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_e2e_768.json 2> /dev/null
python bench.py --workload e2e --latent 96 --steps 5 --warmup 1 --cached --no-cpu-baseline > $OUT/bench_e2e_768_cached.json 2> /dev/null
for f in cfg2 cfg3 cfg3_one_stream cfg3_cached cfg4 cfg4_cached cfg4_one_stream cfg4_two_streams_single_stream_launch_policy cfg2_latent96 cfg3_latent96 cfg4_latent96; do python -c "import json;d=json.load(open('$OUT/bench_$f.json'));print('$f', d['ms_per_step'], d['value'])" || true; done
for f in e2e_512 e2e_512_cached e2e_768 e2e_768_cached; do python -c "import json;d=json.load(open('$OUT/bench_$f.json'));print('$f', d['seconds_per_image'], d['phases_ms'])"; done
fi
if [[ $STAGE == *C* ]]; then
# SQ counters on isolated launches of the hot kernels (tools/pmc_ops.py), a few counters per pass
i=0
for CS in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $CS -d $OUT/pmc_sq$i -o s -- python3 tools/pmc_ops.py > /dev/null 2> $OUT/pmc_sq$i.err || echo "SQ pass $i failed (see pmc_sq$i.err)"
done
python tools/pmc_ops_summary.py $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/pmc_sq4 $OUT/pmc_sq5 > $OUT/pmc_sq_counters.txt || true
rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/pmc_sq4 $OUT/pmc_sq5
fi
if [[ $STAGE == *D* ]]; then
# every B = 32 operator form, 30 launches each, while a second process runs the 32-pair forward on the same GPU
python bench.py --steps 4000 --no-check --no-cpu-baseline --no-profile > /dev/null 2> /dev/null &
BG=$!
sleep 60
REPS=30 timeout -k 10 500 python tools/probe_determinism_shared.py 2>&1 | grep -v amdgpu.ids > $OUT/determinism_shared_gpu.log || echo "determinism probe failed" >> $OUT/determinism_shared_gpu.log
kill $BG 2> /dev/null || true
wait $BG 2> /dev/null || true
grep -c "^ok" $OUT/determinism_shared_gpu.log; grep "^FAIL" $OUT/determinism_shared_gpu.log || echo "no operator form differed"
# the whole bench.py control flow with two ranks on this ONE GPU (its determinism screen runs on both ranks)
MVD_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $OUT/rehearsal_2rank_one_gpu.json 2> $OUT/rehearsal_2rank_one_gpu.err
python -c "import json;d=json.load(open('$OUT/rehearsal_2rank_one_gpu.json'));print('rehearsal', d['value'], d['output_check'], 'distinct_gpus', d['distinct_gpus'])"
fi
# the raw counter csvs are large: keep the summaries only
ls -la $OUT
