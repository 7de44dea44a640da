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
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Grid_Size_X", r.get("Grid_Size", "0"))))
    rows.sort()
    # the forwards only: from the first launch of the target kernel on (model set-up and weight filling run torch kernels before it)
    first = next((i for i, r in enumerate(rows) if target in r[2]), 0)
    first = max(0, first - 40)                      # (the few dozen engine launches in front of a forward's first attention)
    rows = rows[first:]
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
    # ---- timeline: how long does each queue run alone, and what does the main queue run while the side queue is idle?
    main_q = max(queues, key=queues.get)
    side = sorted((s, e) for (s, e, _, q, _) in rows if q != main_q)
    merged = []
    for s_, e_ in side:
        if merged and s_ <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e_)
        else:
            merged.append([s_, e_])
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    side_busy = sum(e_ - s_ for s_, e_ in merged)
    print(f"# wall {1e-6 * (t1 - t0):.2f} ms; side queue(s) busy {100 * side_busy / (t1 - t0):.1f} % of it")
    import bisect
    starts = [m[0] for m in merged]
    alone = collections.defaultdict(lambda: [0, 0, 0])      # (kernel, workgroups) -> [launches, ns alone, ns total]
    for (s_, e_, name, q, grid) in rows:
        if q != main_q:
            continue
        i = bisect.bisect_right(starts, s_) - 1
        ov = 0
        for j in range(max(i, 0), len(merged)):
            if merged[j][0] >= e_:
                break
            ov += max(0, min(e_, merged[j][1]) - max(s_, merged[j][0]))
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
        rec = alone[(short, grid)]
        rec[0] += 1; rec[1] += (e_ - s_) - ov; rec[2] += e_ - s_
    tot_alone = sum(v[1] for v in alone.values())
    print(f"# main-queue kernel time with the side queue idle: {1e-6 * tot_alone:.2f} ms of {1e-6 * sum(v[2] for v in alone.values()):.2f} ms; the largest items "
          "(kernel, grid size in work-items = workgroups x threads): launches, ms alone, share of that kernel's own time")
    for (k, g), v in sorted(alone.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"   {k:44s} grid {g:>9s}  launches {v[0]:4d}  alone {1e-6 * v[1]:7.3f} ms  ({100 * v[1] / max(v[2], 1):5.1f} % of its time)")
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
