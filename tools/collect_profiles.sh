#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: the judged evidence of one round.
#   tools/collect_profiles.sh <round-tag, e.g. r02>
# Writes under gpurun_out/<tag>/: the bench line + per-shape table, the rocprofv3 kernel-trace stats of the same command,
# two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) summarised by tools/pmc_summary.py, and one SQ-counter pass.
set -e -o pipefail
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the PMC traffic passes FIRST: bench.py quotes roofline.traffic from profiles/<tag>_pmc_hbm_traffic.json (stamped with the kernel
# source hash), so the file has to be in place -- in this box's copy of the tree -- before the bench line is produced
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2> $OUT/pmc_write.err
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_hbm_traffic.json > $OUT/pmc_hbm_traffic.txt
cp $OUT/pmc_hbm_traffic.json profiles/${TAG}_pmc_hbm_traffic.json
python bench.py --steps 20 --warmup 5 --shapes-out $OUT/shapes.txt > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
cat $OUT/bench_cfg4.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
# SQ counters on isolated launches of the hot kernels (tools/pmc_ops.py), a few counters per pass
i=0
for CS in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $CS -d $OUT/pmc_sq$i -o s -- python3 tools/pmc_ops.py > /dev/null 2> $OUT/pmc_sq$i.err || echo "SQ pass $i failed (see pmc_sq$i.err)"
done
python tools/pmc_ops_summary.py $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/pmc_sq4 $OUT/pmc_sq5 > $OUT/pmc_sq_counters.txt || true
rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/pmc_sq4 $OUT/pmc_sq5
# the raw counter csvs are large: keep the summaries only
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/prof
ls -la $OUT
