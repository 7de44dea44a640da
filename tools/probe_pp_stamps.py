#!/usr/bin/env python3
"""Phase timing of the ping-pong GEMM from in-kernel shader-clock stamps (probe build: tools/build_variant.py probe -DMVD_PROBE,
run with MVD_HIP_LIB=mvd_amd/libmvd_hip_probe.so MVD_GEMM_DEBUG=32).  For each shape: per-phase work time (release of the
previous barrier -> arrival at the next) and barrier wait (arrival -> release), per wave group, median over workgroups."""
import ctypes as C, math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import _lib as L

def rnd(*s, scale=1.0): return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)
def p(t): return C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

for (m, n, k, res) in [(131072, 1280, 320, 0), (131072, 320, 320, 1), (131072, 2560, 320, 2), (32768, 5120, 640, 2)]:
    geglu = res == 2                                            # (res = 2: the GEGLU form, tile config 6)
    a, w, b = rnd(m, k), rnd(n, k, scale=1 / math.sqrt(k)), torch.randn(n, device="cuda")
    r = rnd(m, n) if res == 1 else None
    out = torch.empty(m, n // 2 if geglu else n, device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(64 * 2 * 512, device="cuda", dtype=torch.int64)
    for _ in range(3):
        L.call("mvd_op_linear", p(a), None, k, 0, p(w), p(b), None, 0, 0, p(r), 1.0, int(geglu), p(out), 0, m, n, 6 if geglu else 7, 1, C.c_void_p(ws.data_ptr()), st)
    torch.cuda.synchronize()
    s = ws.view(64, 2, 512).cpu()
    nslab = k // 64
    print(f"== dense M={m} N={n} K={k} res={res}: {nslab} slabs x 4 phases per tile")
    for g in (0, 1):
        work, wait = {}, {}
        per = 4 * nslab + 2                      # phases per tile after the first: (idle, E | E, idle) + 4 per slab
        for wg in range(64):
            cnt = int(s[wg, g, 0])
            v = s[wg, g, 1:1 + cnt].tolist()
            # stamps alternate (arrive, release); skip the prologue barriers (1 for group 0, 2 for group 1)
            pairs = [(v[i], v[i + 1]) for i in range(0, cnt - 1, 2)]
            pairs = pairs[(1 if g == 0 else 2):]
            for i in range(4 * nslab + 1, len(pairs)):    # from the second tile on (the first has no E phase)
                ph = (i - 4 * nslab) % per
                work.setdefault(ph, []).append(pairs[i][0] - pairs[i - 1][1])
                wait.setdefault(ph, []).append(pairs[i][1] - pairs[i][0])
        head = ["idle", "E"] if g == 0 else ["E", "idle"]
        names = head + [f"{n}{k}" for k in range(nslab) for n in ("R0.", "M0.", "R1.", "M1.")]
        line = []
        for ph in sorted(work):
            line.append(f"{names[ph]}:{statistics.median(work[ph]):.0f}+{statistics.median(wait[ph]):.0f}")
        tot = sum(statistics.median(work[ph]) + statistics.median(wait[ph]) for ph in work)
        print(f"  group {g} (work+wait cycles, {tot:.0f} per tile): " + " ".join(line))
