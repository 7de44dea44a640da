"""End-to-end parity on the GPU: MultiViewUNet mirror -> C ABI -> HIP engine vs the CPU oracle
(oracle/mvd.py, pinned by the reference's golden vectors) on identical weights and inputs.

Tolerance (north_star "within a stated fp tolerance"): bf16 storage / fp32 accumulate against an
fp32 oracle through ~300 sequential ops: relative L2 <= 2e-2 and max-abs <= 5e-2 * max|ref|
(SURVEY.md 8d); intermediate feature maps likewise."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_L2, TOL_MAX = 2e-2, 5e-2


@pytest.fixture(scope="module")
def tiny():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.parity_util import build_pair
    return build_pair("tiny", 0, 96, 48)


def _run(model, inp, t, cam=True, img=True):
    kw = {}
    if cam:
        kw.update(source_camera=inp["src"].cuda(), target_camera=inp["tgt"].cuda())
        model.fourier_projection = inp["proj"]
    if img:
        kw.update(source_image_latents=inp["lat"].cuda())
    with torch.no_grad():
        return model(inp["sample"].cuda(), t, inp["text"].cuda(), **kw).sample


def _oracle(params, cfg, inp, t, cam=True, img=True, feats=None):
    from oracle import mvd as OM
    return OM.multiview_unet_forward(params, cfg, inp["sample"], t, inp["text"], inp["src"] if cam else None,
                                     inp["tgt"] if cam else None, inp["lat"] if img else None,
                                     fourier_proj=inp["proj"], img_ref_scale=0.3, cam_modulation_strength=0.2,
                                     features_out=feats)


@pytest.mark.parametrize("cam,img", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("batch", [1, 3])
def test_tiny_forward_parity(tiny, cam, img, batch):
    from tests.parity_util import make_inputs, max_rel, rel_l2
    cfg, params, model = tiny
    inp = make_inputs(cfg, batch, 16, 7, seed=batch, cam_dim=96)
    t = torch.tensor(321)
    want = _oracle(params, cfg, inp, t, cam, img)
    got = _run(model, inp, t, cam, img)
    assert got.shape == want.shape and torch.isfinite(got).all()
    assert rel_l2(got, want) <= TOL_L2, (rel_l2(got, want), max_rel(got, want))
    assert max_rel(got, want) <= TOL_MAX


@pytest.mark.parametrize("hw,batch,text_len", [((24, 40), 2, 11), ((8, 56), 1, 77), ((40, 8), 3, 5)])
def test_tiny_nonsquare_latents(tiny, hw, batch, text_len):
    """Non-square, non-power-of-two latents (768x1280-style images): ragged GEMM/conv tiles, ragged attention query and
    key tiles at every level, camera + image conditioning on."""
    from tests.parity_util import make_inputs, max_rel, rel_l2
    cfg, params, model = tiny
    inp = make_inputs(cfg, batch, hw, text_len, seed=7, cam_dim=96)
    t = torch.tensor(77)
    want = _oracle(params, cfg, inp, t, True, True)
    got = _run(model, inp, t, True, True)
    assert got.shape == want.shape and torch.isfinite(got).all()
    assert rel_l2(got, want) <= TOL_L2, (rel_l2(got, want), max_rel(got, want))
    assert max_rel(got, want) <= TOL_MAX


def test_tiny_features_and_camera_embedding(tiny):
    """The 16 encoder feature maps (image_encoder.py hooks) and the camera embedding match the oracle."""
    from tests.parity_util import make_inputs, max_rel, rel_l2
    cfg, params, model = tiny
    inp = make_inputs(cfg, 2, 16, 7, seed=5, cam_dim=96)
    feats = {}
    _oracle(params, cfg, inp, torch.tensor(10), feats=feats)
    got = model.image_encoder(inp["lat"].cuda(), inp["text"].cuda(), torch.tensor([0]))
    assert list(got) == list(feats)
    for k in feats:
        assert rel_l2(got[k], feats[k]) <= TOL_L2, (k, rel_l2(got[k], feats[k]))
    from oracle import mvd as OM
    emb_want = OM.camera_embedding(OM._sub(params, "camera_encoder."), inp["src"], inp["tgt"], inp["proj"])
    emb = model.camera_encoder.encode_cameras(inp["src"].cuda(), inp["tgt"].cuda(), inp["proj"].cuda())
    assert max_rel(emb, emb_want) <= 1e-3      # fp32 path


def test_tiny_cfg_batch_mismatch_q4(tiny):
    """Classifier-free guidance: sample batch 2B vs reference batch B (reference quirk Q4)."""
    from oracle import mvd as OM
    from tests.parity_util import make_inputs, rel_l2
    cfg, params, model = tiny
    inp = make_inputs(cfg, 1, 16, 7, seed=9, cam_dim=96)
    x2 = torch.cat([inp["sample"], inp["sample"] * 0.5])
    want = OM.multiview_unet_forward(params, cfg, x2, torch.tensor(77), inp["text"], None, None, inp["lat"])
    with torch.no_grad():
        got = model(x2.cuda(), torch.tensor(77), inp["text"].cuda(), source_image_latents=inp["lat"].cuda()).sample
    assert rel_l2(got, want) <= TOL_L2, rel_l2(got, want)


def test_tiny_reference_cache_is_exact(tiny):
    """Q5: reusing cached K_ref/V_ref across steps is bit-identical to recomputing them."""
    from tests.parity_util import make_inputs
    cfg, params, model = tiny
    inp = make_inputs(cfg, 2, 16, 7, seed=3, cam_dim=96)
    lat, text, x = inp["lat"].cuda(), inp["text"].cuda(), inp["sample"].cuda()
    with torch.no_grad():
        model.cache_reference = False
        a = model(x, torch.tensor(400), text, source_image_latents=lat).sample
        model.cache_reference = True
        model(x, torch.tensor(900), text, source_image_latents=lat)
        b = model(x, torch.tensor(400), text, source_image_latents=lat).sample   # served from the cache
        model.cache_reference = False
    assert torch.equal(a, b)


def test_tiny_streams_shapes_and_two_engines(tiny):
    """Host-side robustness of the boundary: (i) a forward issued on a non-default stream equals the default-stream one,
    (ii) batch / latent size may change from call to call (workspace and reference cache are re-sized), (iii) a second
    engine alive in the same process does not disturb the first."""
    from tests.parity_util import build_pair, make_inputs
    cfg, params, model = tiny

    def run(m, inp, t=250):
        m.fourier_projection = inp["proj"].cuda()
        with torch.no_grad():
            return m(inp["sample"].cuda(), torch.tensor(t), inp["text"].cuda(), source_camera=inp["src"].cuda(),
                     target_camera=inp["tgt"].cuda(), source_image_latents=inp["lat"].cuda()).sample

    a_in = make_inputs(cfg, 2, 16, 7, seed=11, cam_dim=96)
    b_in = make_inputs(cfg, 5, (8, 24), 9, seed=12, cam_dim=96)
    a0 = run(model, a_in).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a1 = run(model, a_in).clone()
    side.synchronize()
    assert torch.equal(a0, a1)                       # (i)
    b0 = run(model, b_in).clone()                    # (ii) bigger batch, other latent size, other text length
    a2 = run(model, a_in).clone()                    #      ... and back
    assert torch.isfinite(b0).all() and b0.shape == (5, cfg.in_channels, 8, 24)
    assert torch.equal(a0, a2)
    _, _, other = build_pair("tiny", 1, 96, 48)      # (iii) second engine, different weights
    o = run(other, a_in)
    assert not torch.equal(o, a0)
    assert torch.equal(run(model, a_in), a0)
    del other


def test_sd21_full_size_parity():
    """Full SD-2.1 shapes (865.9 M + 99.2 M + 19.1 M parameters), B=1, 64x64 latent, 77 text tokens."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.parity_util import run_tiny_parity
    stats = run_tiny_parity(batch=1, verbose=True, cfg_name="sd21", hw=64, text_len=77)
    assert stats["finite"]
    assert stats["rel_l2"] <= TOL_L2, stats
    assert stats["max_rel"] <= TOL_MAX, stats


def test_sd21_batch1_two_stream_forward_is_bit_deterministic():
    """The reference's own usage (infer.py:111-122): batch 1, camera + image conditioning, i.e. the encoder pass on the side stream
    beside the main pass.  Six forwards of the same inputs are bit-identical.  Timing-dependent hazards of the batch-1 kernels
    show HERE and not at op level: round 4's weight-streaming convolution passed every operator test and still produced, in this
    schedule only, one workgroup in ~10^4 with a refill that overtook a read.  Also with the plain UNet (one
    stream) and at batch 2."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.parity_util import make_inputs, shared_pair
    ocfg, _, model = shared_pair("sd21")
    # (round 5: also on 96 x 96 latents -- the 12- / 24-wide forms of the weight-streaming convolution inside the two-stream forward,
    #  and the hipGraph replay whose capture now holds both branches)
    for batch, cond, hw, graph in ((1, True, 64, False), (2, True, 64, False), (1, False, 64, False), (1, True, 96, False), (1, True, 64, True)):
        inp = make_inputs(ocfg, batch, hw, 77, 5, 1024)
        model.use_hip_graph = graph
        model.fourier_projection = inp["proj"].cuda()
        kw = dict(source_camera=inp["src"].cuda(), target_camera=inp["tgt"].cuda(), source_image_latents=inp["lat"].cuda()) if cond else {}
        outs = []
        with torch.no_grad():
            for _ in range(6):
                outs.append(model(inp["sample"].cuda(), torch.tensor(400), inp["text"].cuda(), **kw).sample.clone())
        torch.cuda.synchronize()
        assert torch.isfinite(outs[0]).all()
        model.use_hip_graph = False
        for i in range(1, 6):
            assert torch.equal(outs[0], outs[i]), (f"batch {batch} conditioning {cond} latent {hw} graph {graph}: forward {i} differs from "
                                                   f"forward 0 in {int((outs[0] != outs[i]).sum())} elements")


def test_sd21_full_size_parity_base_unet_only():
    """BASELINE configs[1]: the plain SD-2.1 UNet (no camera, no image conditioning) at full size, B=1 -- the batch-1 kernel
    choices (64x64 / 128x64 tiles, deep split-K, the two-kernel GroupNorm at 64x64) under the checker."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.parity_util import run_tiny_parity
    stats = run_tiny_parity(batch=1, verbose=True, cfg_name="sd21", hw=64, text_len=77, cam=False, img=False)
    assert stats["finite"]
    assert stats["rel_l2"] <= TOL_L2, stats
    assert stats["max_rel"] <= TOL_MAX, stats


def test_ddpm_step_and_cfg_kernels():
    """Row N2: fused DDPM step / CFG combine kernels vs the oracle algebra."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mvd_amd import ops
    from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler
    from oracle import scheduler as OS
    s = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
    s.set_timesteps(20)
    g = torch.Generator().manual_seed(0)
    x, mo, nz = (torch.randn(3, 4, 16, 16, generator=g) for _ in range(3))
    for t in (950, 50, 0):
        got = s.step(mo.cuda(), t, x.cuda(), noise=nz.cuda()).prev_sample
        want = OS.ddpm_step(mo, t, x, s.alphas_cumprod, 1000, 20, "v_prediction", nz)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)
    both = torch.randn(4, 4, 8, 8, generator=g)
    u, c = both.chunk(2)
    torch.testing.assert_close(ops.cfg_combine(both.cuda(), 7.5).cpu(), u + 7.5 * (c - u), rtol=1e-5, atol=1e-5)


def test_tiny_denoise_loop_parity(tiny):
    """Row N1: 4-step CFG denoising loop (pipeline.py:140-166 semantics) vs the oracle loop, same noise draws."""
    from mvd_amd.pipeline import MVDDenoiser
    from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler
    from oracle import scheduler as OS
    from tests.parity_util import make_inputs, rel_l2
    cfg, params, model = tiny
    inp = make_inputs(cfg, 2, 16, 7, seed=11, cam_dim=96)
    sched = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
    steps, gs = 4, 3.0
    g = torch.Generator().manual_seed(5)
    noises = [torch.randn(2, 4, 16, 16, generator=g) for _ in range(steps)]
    neg = torch.randn(2, 7, cfg.cross_attention_dim, generator=g)
    lat0 = torch.randn(2, 4, 16, 16, generator=g)
    want = OS.denoise_loop(params, cfg, sched.betas, inp["text"], neg, lat0, None, None, inp["lat"], steps, gs, noises, None,
                           img_ref_scale=0.3, cam_modulation_strength=0.2)
    model.cache_reference = True          # Q5: reference K/V computed once, reused over the steps (bit-identical)
    den = MVDDenoiser(model, sched)
    got = den(inp["text"].cuda(), steps, gs, negative_prompt_embeds=neg.cuda(), latents=lat0.cuda(),
              source_image_latents=inp["lat"].cuda(), noise_per_step=[n.cuda() for n in noises])
    model.cache_reference = False
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) <= 4e-2, rel_l2(got, want)      # 4 chained bf16 UNet evaluations


@pytest.mark.parametrize("steps,gs,final_tol,growth", [(20, 1.0, 2e-2, 1.5), (50, 7.5, 1.6e-1, 1.5)])
def test_bf16_drift_over_a_real_schedule(tiny, steps, gs, final_tol, growth):
    """How far bf16 storage drifts from the fp32 oracle over a REAL schedule: 20 steps at guidance 1.0 (infer.py's defaults,
    infer.py:181-184) and 50 steps under classifier-free guidance 7.5 (BASELINE configs[1]'s step count, the pipeline default
    guidance), tiny topology, camera + image conditioning, Q1's projection pinned per step, the oracle's own ancestral noise
    draws (so the two trajectories differ by arithmetic only).  Checked on the WHOLE trajectory:
      * final latents: rel-L2 <= 8e-2 (one forward is ~1e-2; the tests above bound 4 steps by 4-5e-2);
      * no blow-up: the error after any step is <= 2.5x the error of a single forward times sqrt(steps so far) + 1e-2,
        i.e. it grows like a random walk of per-step rounding errors that the (contractive) DDPM update keeps damping,
        not linearly or geometrically."""
    from mvd_amd.pipeline import MVDDenoiser
    from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler
    from oracle import scheduler as OS
    from tests.parity_util import make_inputs, rel_l2
    cfg, params, model = tiny
    B = 1
    inp = make_inputs(cfg, B, 16, 7, seed=21, cam_dim=96)
    sched = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
    g = torch.Generator().manual_seed(17)
    noises = [torch.randn(B, 4, 16, 16, generator=g) for _ in range(steps)]
    neg = torch.randn(B, 7, cfg.cross_attention_dim, generator=g)
    lat0 = torch.randn(B, 4, 16, 16, generator=g)
    projs = [inp["proj"]] * steps
    want_tr = []
    OS.denoise_loop(params, cfg, sched.betas, inp["text"], neg if gs > 1 else None, lat0, inp["src"], inp["tgt"], inp["lat"], steps, gs,
                    noises, projs, trace=want_tr, img_ref_scale=0.3, cam_modulation_strength=0.2)
    model.fourier_projection = inp["proj"]
    model.cache_reference = True
    got_tr = []
    den = MVDDenoiser(model, sched)
    den(inp["text"].cuda(), steps, gs, negative_prompt_embeds=neg.cuda() if gs > 1 else None, latents=lat0.cuda(),
        source_camera=inp["src"].cuda(), target_camera=inp["tgt"].cuda(), source_image_latents=inp["lat"].cuda(),
        noise_per_step=[n.cuda() for n in noises], callback=lambda i, t, l: got_tr.append(l.float().cpu().clone()))
    model.cache_reference = False
    model.fourier_projection = None
    assert len(got_tr) == steps == len(want_tr)
    errs = [rel_l2(a, b) for a, b in zip(got_tr, want_tr)]
    print(f"bf16 drift, {steps} steps, guidance {gs}: rel-L2 per step = " + " ".join(f"{e:.1e}" for e in errs))
    assert all(torch.isfinite(t).all() for t in got_tr)
    assert errs[-1] <= final_tol, errs[-5:]
    for i in range(1, steps):     # no jump: a step multiplies the accumulated error by at most `growth` (+ its own rounding error)
        assert errs[i] <= growth * errs[i - 1] + 2e-3, (i, errs[i - 1], errs[i])


def test_small_m_kill_switch_falls_back_instead_of_failing():
    """with the small-M kernels switched off (mvd_debug_set_flags bit 2) a batch-1 forward at SD-2.1 widths must fall
    back -- ln_kernel + the general GEMMs at C = 1280, where the ping-pong LayerNorm fold does not apply -- not stop with
    "LayerNorm fold not available".  32 x 32 latents keep the CPU oracle short; the base UNet only (no adapter, no camera)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mvd_amd import _lib as L
    from tests.parity_util import run_tiny_parity
    L.lib().mvd_debug_set_flags(4)
    try:
        stats = run_tiny_parity(batch=1, verbose=True, cfg_name="sd21", hw=32, text_len=77, cam=False, img=False)
    finally:
        L.lib().mvd_debug_set_flags(0)
    assert stats["finite"] and stats["rel_l2"] <= TOL_L2 and stats["max_rel"] <= TOL_MAX, stats


def test_lean_packing_without_small_batch_twins_falls_back(monkeypatch):
    """MVD_PACK_SMALL_BATCH_TWINS=0 leaves out the ``.ws`` and the C > 640 ``.wf/.cf`` copies; the engine must
    then take the tiled convolutions / ln_kernel + plain GEMM at batch 1 and stay within the same tolerance."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.parity_util import run_tiny_parity
    monkeypatch.setenv("MVD_PACK_SMALL_BATCH_TWINS", "0")
    stats = run_tiny_parity(batch=1, verbose=True)
    assert stats["finite"] and stats["rel_l2"] <= TOL_L2 and stats["max_rel"] <= TOL_MAX, stats


def test_sd21_full_size_parity_768():
    """The reference's own default: 768 x 768 images = 96 x 96 latents (infer.py:187, config/train_config.yaml sample_size 96), full
    SD-2.1 shapes, B = 1, camera FiLM + cross-view adapter, cold forward: 9216 tokens at the first level (the split-KV attention,
    the small-M GEMMs at M = 9216 / 2304 / 576 / 144) under the checker.  The oracle needs ~40 s of CPU."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.parity_util import run_tiny_parity
    stats = run_tiny_parity(batch=1, verbose=True, cfg_name="sd21", hw=96, text_len=77)
    assert stats["finite"]
    assert stats["rel_l2"] <= TOL_L2, stats
    assert stats["max_rel"] <= TOL_MAX, stats


def test_sd21_full_size_denoise_loop_cfg():
    """A 4-step classifier-free-guidance loop of ``MVDDenoiser`` (pipeline.py:140-166) at full SD-2.1 size: B = 1 object (2 latents
    per forward under CFG: the Q4 re-chunking of the reference tokens), camera + image conditioning, the per-step Fourier
    projection pinned, the same ancestral noise draws as ``oracle/scheduler.denoise_loop``.  Quantifies the error growth over
    chained bf16 forwards (one forward: rel-L2 ~1e-2): stated tolerance rel-L2 <= 5e-2 on the final latents."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mvd_amd.pipeline import MVDDenoiser
    from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler
    from oracle import scheduler as OS
    from tests.parity_util import make_inputs, rel_l2, shared_pair
    cfg, params, model = shared_pair("sd21")
    inp = make_inputs(cfg, 1, 64, 77, seed=41, cam_dim=1024)
    sched = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
    steps, gs = 4, 3.0
    g = torch.Generator().manual_seed(6)
    noises = [torch.randn(1, 4, 64, 64, generator=g) for _ in range(steps)]
    neg = torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
    lat0 = torch.randn(1, 4, 64, 64, generator=g)
    want = OS.denoise_loop(params, cfg, sched.betas, inp["text"], neg, lat0, inp["src"], inp["tgt"], inp["lat"], steps, gs, noises,
                           [inp["proj"]] * steps, img_ref_scale=0.3, cam_modulation_strength=0.2)
    model.fourier_projection = inp["proj"]
    den = MVDDenoiser(model, sched)
    got = den(inp["text"].cuda(), steps, gs, negative_prompt_embeds=neg.cuda(), latents=lat0.cuda(), source_camera=inp["src"].cuda(),
              target_camera=inp["tgt"].cuda(), source_image_latents=inp["lat"].cuda(), noise_per_step=[n.cuda() for n in noises])
    model.fourier_projection = None
    err = rel_l2(got, want)
    print("full-size 4-step CFG loop rel-L2", err, flush=True)
    assert torch.isfinite(got).all()
    assert err <= 5e-2, err


def test_sd21_full_size_infer_defaults_loop():
    """The reference's own driver settings under the checker (round-4 verdict, item 2): ``infer.py:181-187`` runs B = 1, **20 steps,
    guidance scale 1.0** (no CFG: one latent per forward), camera + image conditioning, the reference encoder re-run every step
    (the reference-faithful cold forward, Q5 off) -- here at full SD-2.1 size on 64 x 64 latents (BASELINE's 512 x 512), Q1's
    projection pinned per step, the oracle's ancestral noise draws (the trajectories differ by arithmetic only), against
    ``oracle/scheduler.denoise_loop`` (pipeline.py:140-166).  Checked on the WHOLE trajectory.  Stated tolerance: rel-L2 <= 2.5e-2 on
    the final latents (one forward is ~1e-2; the DDPM update damps it: the tiny-topology drift test ends at 6.5e-3 after the same
    20 steps), and no step multiplies the accumulated error by more than 1.5 (+2e-3).  ~55 s of CPU oracle on 16 threads."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mvd_amd.pipeline import MVDDenoiser
    from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler
    from oracle import scheduler as OS
    from tests.parity_util import make_inputs, rel_l2, shared_pair
    cfg, params, model = shared_pair("sd21")
    inp = make_inputs(cfg, 1, 64, 77, seed=43, cam_dim=1024)
    sched = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
    steps, gs = 20, 1.0
    g = torch.Generator().manual_seed(8)
    noises = [torch.randn(1, 4, 64, 64, generator=g) for _ in range(steps)]
    lat0 = torch.randn(1, 4, 64, 64, generator=g)
    want_tr = []
    OS.denoise_loop(params, cfg, sched.betas, inp["text"], None, lat0, inp["src"], inp["tgt"], inp["lat"], steps, gs, noises,
                    [inp["proj"]] * steps, trace=want_tr, img_ref_scale=0.3, cam_modulation_strength=0.2)
    model.fourier_projection = inp["proj"]
    got_tr = []
    try:
        den = MVDDenoiser(model, sched)
        den(inp["text"].cuda(), steps, gs, latents=lat0.cuda(), source_camera=inp["src"].cuda(), target_camera=inp["tgt"].cuda(),
            source_image_latents=inp["lat"].cuda(), noise_per_step=[n.cuda() for n in noises],
            callback=lambda i, t, l: got_tr.append(l.float().cpu().clone()))
    finally:
        model.fourier_projection = None
    assert len(got_tr) == steps == len(want_tr)
    errs = [rel_l2(a, b) for a, b in zip(got_tr, want_tr)]
    print("full-size 20-step guidance-1.0 loop, rel-L2 per step = " + " ".join(f"{e:.1e}" for e in errs), flush=True)
    assert all(torch.isfinite(t).all() for t in got_tr)
    assert errs[-1] <= 2.5e-2, errs[-5:]
    for i in range(1, steps):
        assert errs[i] <= 1.5 * errs[i - 1] + 2e-3, (i, errs[i - 1], errs[i])


def test_sd21_full_size_pipeline_defaults_loop_cfg():
    """The pipeline's own defaults at full SD-2.1 size (round 5; the round-4 verdict found drift bounded "loosely and only at toy
    size"): **50 steps, classifier-free guidance 7.5** (pipeline.py:12-38; BASELINE configs[1]'s step count), B = 1 object = 2 latents
    per forward (Q4 re-chunking), camera + image conditioning with the reference encoder re-run every step (cold, as the reference),
    Q1's projection pinned, the oracle's ancestral noise draws, against ``oracle/scheduler.denoise_loop`` on the whole trajectory.
    Measured (profiles/r05_probe_drift_full_size.log): 5.0e-4 after one step, 1.44e-2 at the end, ~3e-4 per step, largest
    step-to-step factor 1.46.  Stated tolerance: final rel-L2 <= 3e-2, no step multiplies the accumulated error by more than 1.6
    (+2e-3).  ~135 s of CPU oracle on 16 threads."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mvd_amd.pipeline import MVDDenoiser
    from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler
    from oracle import scheduler as OS
    from tests.parity_util import make_inputs, rel_l2, shared_pair
    cfg, params, model = shared_pair("sd21")
    inp = make_inputs(cfg, 1, 64, 77, seed=47, cam_dim=1024)
    sched = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
    steps, gs = 50, 7.5
    g = torch.Generator().manual_seed(9)
    noises = [torch.randn(1, 4, 64, 64, generator=g) for _ in range(steps)]
    neg = torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
    lat0 = torch.randn(1, 4, 64, 64, generator=g)
    want_tr = []
    OS.denoise_loop(params, cfg, sched.betas, inp["text"], neg, lat0, inp["src"], inp["tgt"], inp["lat"], steps, gs, noises,
                    [inp["proj"]] * steps, trace=want_tr, img_ref_scale=0.3, cam_modulation_strength=0.2)
    model.fourier_projection = inp["proj"]
    got_tr = []
    try:
        den = MVDDenoiser(model, sched)
        den(inp["text"].cuda(), steps, gs, negative_prompt_embeds=neg.cuda(), latents=lat0.cuda(), source_camera=inp["src"].cuda(),
            target_camera=inp["tgt"].cuda(), source_image_latents=inp["lat"].cuda(), noise_per_step=[n.cuda() for n in noises],
            callback=lambda i, t, l: got_tr.append(l.float().cpu().clone()))
    finally:
        model.fourier_projection = None
    assert len(got_tr) == steps == len(want_tr)
    errs = [rel_l2(a, b) for a, b in zip(got_tr, want_tr)]
    print("full-size 50-step CFG 7.5 loop, rel-L2 per step = " + " ".join(f"{e:.1e}" for e in errs), flush=True)
    assert all(torch.isfinite(t).all() for t in got_tr)
    assert errs[-1] <= 3e-2, errs[-5:]
    for i in range(1, steps):
        assert errs[i] <= 1.6 * errs[i - 1] + 2e-3, (i, errs[i - 1], errs[i])
