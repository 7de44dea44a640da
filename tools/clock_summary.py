#!/usr/bin/env python3
"""Shader clock under load per kernel class from one `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` pass of bench.py:
effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (MI355X_MICROARCH.md, DVFS section; valid for dispatches of
>= 0.3 ms, so only classes whose average launch is that long are reported)."""
import collections, csv, glob, json, re, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import CLASSES


def main(d, out):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name", "GRBM_GUI_ACTIVE") != "GRBM_GUI_ACTIVE":
            continue
        t0, t1 = r.get("Start_Timestamp"), r.get("End_Timestamp")
        if not t0 or not t1:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += float(t1) - float(t0)
    res = {}
    for cls, pat in CLASSES.items():
        rx = pat if pat.startswith("gemm_pp") else re.escape(pat)
        n = sum(v[0] for k, v in acc.items() if re.search(rx, k))
        if not n:
            continue
        cyc = sum(v[1] for k, v in acc.items() if re.search(rx, k))
        ns = sum(v[2] for k, v in acc.items() if re.search(rx, k))
        if ns / n < 0.25e6:
            continue
        res[cls] = {"launches": n, "avg_launch_us": ns / n / 1e3, "clock_ghz": cyc / 8.0 / ns}
        print(f"{cls:26s} launches={n:4d} avg {ns / n / 1e3:8.1f} us  clock {cyc / 8.0 / ns:.3f} GHz (profiled pass: counters lower the clock by 2-3 %)")
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
