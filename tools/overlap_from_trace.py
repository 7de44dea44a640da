#!/usr/bin/env python3
"""Round-4 verdict, item 8: does attention run slower in the two-stream forward because it MEETS ATTENTION of the other pass?

Input: the kernel trace of a two-stream bench run (`rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --no-profile
--no-check --no-cpu-baseline --steps 4 --warmup 2`): one row per dispatch with start / end timestamps and the queue it ran on.
For every launch of the dominant attention kernel this script finds what ran on the OTHER queue(s) during its lifetime and splits its
duration by the partner's class (attention / matrix-bound GEMM or convolution / HBM-bound short-K GEMM, GroupNorm, LayerNorm /
nothing).  Then, per shape (grid size), it compares the launch duration by dominant partner class.  If attention-beside-attention is
not slower than attention-beside-anything-else, re-phasing the two passes cannot buy anything.

    python tools/overlap_from_trace.py <dir with *kernel_trace.csv> [--kernel attn_kernel]
"""
import collections
import csv
import glob
import sys


def klass(name: str) -> str:
    n = name
    if "attn_kernel" in n:
        return "attention"
    if "gemm_xs_kernel" in n or "gn_" in n or "ln_kernel" in n or "refnorm" in n or "film" in n or "splitk_reduce" in n or "im2col" in n:
        return "hbm-bound"
    if "gemm" in n or "conv" in n:
        return "matrix"
    return "other"


def main():
    d = sys.argv[1]
    target = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "attn_kernel<4"
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Grid_Size", "0")))
    rows.sort()
    if not rows:
        print("no kernel trace rows found under", d)
        return
    queues = collections.Counter(q for _, _, _, q, _ in rows)
    print(f"# {len(rows)} dispatches on queues {dict(queues)}")
    by_q = collections.defaultdict(list)
    for r in rows:
        by_q[r[3]].append(r)
    stats = collections.defaultdict(lambda: collections.defaultdict(list))   # grid -> dominant partner -> [durations]
    share = collections.Counter()
    for (s, e, name, q, grid) in rows:
        if target not in name:
            continue
        dur = e - s
        ov = collections.Counter()
        for q2, lst in by_q.items():
            if q2 == q:
                continue
            for (s2, e2, n2, _, _) in lst:          # (lists are short enough: a few thousand dispatches)
                if e2 <= s:
                    continue
                if s2 >= e:
                    break
                ov[klass(n2)] += min(e, e2) - max(s, s2)
        covered = sum(ov.values())
        ov["alone"] = max(dur - covered, 0)
        dom = max(ov, key=ov.get)
        stats[grid][dom].append(dur / 1e3)
        for k, v in ov.items():
            share[k] += v
    tot = sum(share.values())
    print("# share of the attention kernel's lifetime by what the other queue was running: " +
          ", ".join(f"{k} {100 * v / tot:.1f} %" for k, v in share.most_common()))
    print("# per shape (grid size): mean duration in us by DOMINANT partner class (launch count)")
    for grid, dd in sorted(stats.items(), key=lambda kv: -sum(len(v) for v in kv[1].values())):
        n = sum(len(v) for v in dd.values())
        if n < 8:
            continue
        line = "  ".join(f"{k}: {sum(v) / len(v):7.1f} ({len(v)})" for k, v in sorted(dd.items()))
        print(f"grid {grid:>10s}  launches {n:4d}   {line}")


if __name__ == "__main__":
    main()
