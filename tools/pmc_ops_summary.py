#!/usr/bin/env python3
"""Per-kernel counters from rocprofv3 --pmc runs of tools/pmc_ops.py: pmc_ops_summary.py <dir> [<dir>...]
pmc_ops.py launches every shape 3 times in a row, so consecutive triples of one kernel name are one shape: the
summary prints one block per (kernel, triple ordinal), each counter averaged over the triple."""
import collections, csv, glob, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))        # kernel -> counter -> [(dispatch, value)]
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_kernel" not in k and "gemm_pp_kernel" not in k and "attn_kernel" not in k and "gemm_xs_kernel" not in k and "conv_ws_kernel" not in k: continue
            k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
            rows[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, cs in rows.items():
    ngroups = max(len(v) for v in cs.values()) // 3
    for g in range(max(ngroups, 1)):
        print(f"{k}  [shape #{g}]")
        for c, v in sorted(cs.items()):
            v = [x[1] for x in sorted(v)][3 * g:3 * g + 3]
            if v:
                print(f"   {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
