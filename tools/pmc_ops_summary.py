#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 --pmc output dirs: pmc_ops_summary.py <dir> [<dir>...]"""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_kernel" not in k and "attn_kernel" not in k: continue
            k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
            a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, (n, s) in sorted(cs.items()):
        print(f"   {c:32s} {s / n:16.0f}  (n={n})")
