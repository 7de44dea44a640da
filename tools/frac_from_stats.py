#!/usr/bin/env python3
"""Recompute roofline.frac of a bench line from a rocprofv3 --kernel-trace --stats summary of the same command:
    tools/frac_from_stats.py <kernel_stats.csv> <bench line .json> [one|two]
frac = flops_per_launch / (average duration of the dominant kernel's launches in the CSV) / peak.  `one`: the CSV is of
`bench.py --debug-flags 16 --no-profile` (one stream: compare with roofline.avg_launch_us / frac); `two` (default): of the default
two-stream schedule (`--no-profile`: compare with avg_launch_us_overlapped / frac_overlapped)."""
import csv, json, re, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import CLASSES


def main(stats, line, mode="two"):
    d = json.loads(open(line).read().strip().splitlines()[-1])
    r = d["roofline"]
    pat = CLASSES[r["kernel"]]
    rx = pat if pat.startswith("gemm_pp") else re.escape(pat)
    calls = ns = 0
    for row in csv.DictReader(open(stats)):
        if re.search(rx, row["Name"]):
            calls += int(row["Calls"]); ns += float(row["TotalDurationNs"])
    avg_us = ns / calls / 1e3
    frac = r["flops_per_launch"] / (avg_us * 1e-6) / (r["peak"] * 1e12)
    ref_us = r["avg_launch_us"] if mode == "one" else r.get("avg_launch_us_overlapped")
    ref_frac = r["frac"] if mode == "one" else r.get("frac_overlapped")
    print(f"{r['kernel']}: {calls} launches in the CSV, average {avg_us:.2f} us -> {r['flops_per_launch'] / (avg_us * 1e-6) / 1e12:.1f} TFLOP/s = frac {frac:.4f}"
          f"  |  bench line ({'one stream' if mode == 'one' else 'two streams'}): {ref_us} us, frac {ref_frac}"
          f"  |  ratio {avg_us / ref_us:.3f}" if ref_us else "")


if __name__ == "__main__":
    main(*sys.argv[1:])
