set -e -o pipefail
OUT=$PWD/gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for CS in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $CS -d $OUT/pmc_sq$i -o s -- python3 tools/pmc_ops.py > /dev/null 2> $OUT/pmc_sq$i.err || echo "SQ pass $i failed"
done
python tools/pmc_ops_summary.py $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/pmc_sq4 $OUT/pmc_sq5 > $OUT/pmc_sq_counters.txt
rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/pmc_sq4 $OUT/pmc_sq5
head -16 $OUT/pmc_sq_counters.txt
