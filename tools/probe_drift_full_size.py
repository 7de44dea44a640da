#!/usr/bin/env python3
"""One-off measurement (round 5; too long for the suite: ~5 min of CPU oracle): bf16 drift of the HIP path against the fp32 oracle over
the PIPELINE DEFAULT schedule at FULL SD-2.1 size -- 50 steps, classifier-free guidance 7.5 (pipeline.py:12-38 defaults; BASELINE
configs[1]'s step count), B = 1 object = 2 latents per forward (Q4 re-chunking), camera + image conditioning, Q1's projection pinned,
the oracle's own ancestral noise draws.  Prints rel-L2 of the latents after every step.

    python tools/probe_drift_full_size.py [steps=50] [guidance=7.5]      (needs a GPU; imports oracle/ as the checker)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import oracle
from mvd_amd.pipeline import MVDDenoiser
from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler
from oracle import scheduler as OS
from tests.parity_util import make_inputs, rel_l2, shared_pair

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
gs = float(sys.argv[2]) if len(sys.argv) > 2 else 7.5
torch.set_num_threads(oracle.host_threads())
cfg, params, model = shared_pair("sd21")
inp = make_inputs(cfg, 1, 64, 77, seed=47, cam_dim=1024)
sched = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
g = torch.Generator().manual_seed(9)
noises = [torch.randn(1, 4, 64, 64, generator=g) for _ in range(steps)]
neg = torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
lat0 = torch.randn(1, 4, 64, 64, generator=g)
model.fourier_projection = inp["proj"]
got_tr = []
den = MVDDenoiser(model, sched)
t0 = time.time()
den(inp["text"].cuda(), steps, gs, negative_prompt_embeds=neg.cuda() if gs > 1 else None, latents=lat0.cuda(), source_camera=inp["src"].cuda(),
    target_camera=inp["tgt"].cuda(), source_image_latents=inp["lat"].cuda(), noise_per_step=[n.cuda() for n in noises],
    callback=lambda i, t, l: got_tr.append(l.float().cpu().clone()))
torch.cuda.synchronize()
t_gpu = time.time() - t0
want_tr = []
t0 = time.time()
OS.denoise_loop(params, cfg, sched.betas, inp["text"], neg if gs > 1 else None, lat0, inp["src"], inp["tgt"], inp["lat"], steps, gs, noises,
                [inp["proj"]] * steps, trace=want_tr, img_ref_scale=0.3, cam_modulation_strength=0.2)
t_cpu = time.time() - t0
errs = [rel_l2(a, b) for a, b in zip(got_tr, want_tr)]
print(f"# full SD-2.1 size, 64x64 latents, B = 1, {steps} steps, guidance {gs}: HIP loop {t_gpu:.2f} s, CPU oracle {t_cpu:.0f} s on {torch.get_num_threads()} threads")
print("rel-L2 of the latents after each step: " + " ".join(f"{e:.2e}" for e in errs))
growth = max((errs[i] / max(errs[i - 1], 1e-9) for i in range(1, steps)), default=0.0)
print(f"final {errs[-1]:.3e}; largest step-to-step growth factor {growth:.2f}; finite {all(torch.isfinite(t).all().item() for t in got_tr)}")
