#!/usr/bin/env python3
"""How many host threads should the CPU oracle use on the GPU box?  The box's cgroup grants a CPU QUOTA (cpu.max, 16 cores on the
pool's one-GPU boxes) while os.cpu_count() / the affinity mask show all 256 hardware threads and torch defaults to 128 intra-op
threads.  Times the fp32 base-UNet forward of oracle/sd21_unet.py (B = 1, 32 x 32 latent) at several thread counts.  CPU only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle import sd21_unet as OU  # noqa: E402


def cgroup_cpus():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except OSError:
        return None


print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "cgroup quota (cores)", cgroup_cpus(),
      "torch default threads", torch.get_num_threads(), flush=True)
cfg = OU.UNetConfig.sd21()
t0 = time.perf_counter()
p = OU.init_params(cfg, 0)
print(f"init_params {time.perf_counter() - t0:.1f} s", flush=True)
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 4, 32, 32, generator=g)
text = torch.randn(1, 77, 1024, generator=g)
order = [int(a) for a in sys.argv[1:]] or [16, 32, 128]
with torch.no_grad():
    for n in order:
        torch.set_num_threads(n)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            OU.unet_forward(p, cfg, x, torch.tensor([500]), text)
            ts.append(time.perf_counter() - t0)
        print(f"threads {n:4d}: {' '.join(f'{t:.2f}' for t in ts)} s per forward (first includes warm-up)", flush=True)
