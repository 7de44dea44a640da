#!/bin/bash
# round 5, GPU call 1: Winograd core pricing (+ counters), the GPU suite with the un-skippable headline tests, a cfg4 bench line
set -o pipefail
export OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_wino_core tools/probe_wino_core.hip 2> $OUT/wino_build.err || { echo "wino build failed"; cat $OUT/wino_build.err; exit 1; }
timeout -k 10 300 /tmp/probe_wino_core > $OUT/probe_wino_core.log 2>&1 || { echo "wino probe failed"; tail -5 $OUT/probe_wino_core.log; exit 1; }
cat $OUT/probe_wino_core.log
for CS in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $CS | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $CS -d $OUT/wino_pmc_$tag -o s -- /tmp/probe_wino_core > /dev/null 2> $OUT/wino_pmc_$tag.err || echo "pmc pass $tag failed"
done
python - <<'PY' > $OUT/probe_wino_core_counters.txt 2>&1
import csv, glob, os, collections
out = os.environ.get("OUT", "gpurun_out/r05")
rows = collections.OrderedDict()
for d in sorted(glob.glob(f"{out}/wino_pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino_core" not in r.get("Kernel_Name", ""):
                continue
            key = (r["Counter_Name"], r.get("Grid_Size"), )
            rows.setdefault(key, []).append(float(r["Counter_Value"]))
print("# per-launch averages of the wino_core_kernel launches, by counter and grid size (one grid size = one shape of the probe)")
for (c, g), v in rows.items():
    print(f"{c:32s} grid {g:>10s}  launches {len(v):3d}  mean {sum(v)/len(v):.4g}")
PY
cat $OUT/probe_wino_core_counters.txt | head -60
rm -rf $OUT/wino_pmc_*/
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/gputests.log 2>&1; echo "pytest rc $?"; tail -25 $OUT/gputests.log
timeout -k 10 600 python bench.py > $OUT/bench_cfg4_lean.json 2> $OUT/bench_cfg4_lean.err; echo "bench rc $?"; cat $OUT/bench_cfg4_lean.json
