"""The callers of the hot path on the GPU:

* ``MVDPipeline.__call__`` / ``create_mvd_pipeline`` imported THROUGH the drop-in ``src.models`` shims (integration/src),
  driven the way infer.py:113-122 drives the reference: classifier-free guidance (2B latents), ONE camera pair for the
  whole batch, source latents, against ``oracle/scheduler.py::denoise_loop``;
* camera batch smaller than the sample batch (the (1,C,1,1) FiLM broadcast of the reference);
* the Q5 reference cache can never serve a different object's K/V;
* N4: identical base / image-encoder weights are packed once.

Tolerances: single forward rel-L2 <= 2e-2 / max-abs <= 5e-2*max|ref| (as tests/test_engine_gpu.py); a 4-step loop of
chained bf16 UNet evaluations rel-L2 <= 4e-2.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tiny():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.parity_util import build_pair
    return build_pair("tiny", 0, 96, 48)


@pytest.fixture()
def shim_path():
    """Put integration/ at the head of sys.path so that ``import src.models...`` resolves to the drop-in shims."""
    p = os.path.join(ROOT, "integration")
    sys.path.insert(0, p)
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    yield p
    sys.path.remove(p)
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]


def test_camera_batch_one_broadcasts_over_cfg_batch(tiny):
    """2B latents with B cameras (B = 1, infer.py) -- scale/shift broadcast like torch's (1,C,1,1)."""
    from oracle import mvd as OM
    from tests.parity_util import make_inputs, max_rel, rel_l2
    cfg, params, model = tiny
    inp = make_inputs(cfg, 1, 16, 7, seed=21, cam_dim=96)
    x2 = torch.cat([inp["sample"], inp["sample"] * 0.7 + 0.1])
    want = OM.multiview_unet_forward(params, cfg, x2, torch.tensor(123), inp["text"], inp["src"], inp["tgt"], inp["lat"],
                                     fourier_proj=inp["proj"], img_ref_scale=0.3, cam_modulation_strength=0.2)
    model.fourier_projection = inp["proj"]
    with torch.no_grad():
        got = model(x2.cuda(), torch.tensor(123), inp["text"].cuda(), source_camera=inp["src"].cuda(),
                    target_camera=inp["tgt"].cuda(), source_image_latents=inp["lat"].cuda()).sample
    assert model.current_camera_embedding.shape == (1, 96)
    assert rel_l2(got, want) <= 2e-2 and max_rel(got, want) <= 5e-2, (rel_l2(got, want), max_rel(got, want))
    # a camera batch that does not divide the sample batch is an error, not a silent mis-broadcast
    from mvd_amd._lib import MvdError
    x3 = torch.cat([x2, x2[:1]])
    with pytest.raises(MvdError, match="divide"):
        cams = torch.cat([inp["src"], inp["src"]]).cuda()
        model(x3.cuda(), torch.tensor(123), inp["text"].cuda(), source_camera=cams, target_camera=cams)


def test_pipeline_through_shims_cfg_with_cameras(tiny, shim_path):
    """a17 + N1: ``from src.models.mvd_unet import create_mvd_pipeline`` / ``src.models.pipeline.MVDPipeline`` (the
    imports of infer.py:1-2, val.py:24-26), reference call signature, CFG + ONE camera pair + source latents, 4 steps,
    vs the oracle loop on the same noise draws."""
    from src.models.mvd_unet import MultiViewUNet, create_mvd_pipeline          # noqa: F401  (the shims)
    from src.models.pipeline import MVDPipeline
    from src.models.camera_encoder import CameraEncoder                        # noqa: F401
    from src.models.attention import ImageCrossAttentionProcessor              # noqa: F401
    from src.models.image_encoder import ImageEncoder                          # noqa: F401
    from src.training.scheduler import ShiftSNRScheduler                       # noqa: F401
    from src.utils import create_camera_matrix
    from mvd_amd.config import UNetConfig
    from oracle import scheduler as OS
    from tests.parity_util import make_inputs, rel_l2
    cfg, params, _ = tiny
    pipe = create_mvd_pipeline(None, dtype=torch.float32, img_ref_scale=0.3, cam_modulation_strength=0.2, cam_output_dim=96,
                               cam_hidden_dim=48, unet_config=UNetConfig.tiny(), init="empty")
    assert isinstance(pipe, MVDPipeline) and isinstance(pipe.unet, MultiViewUNet)
    assert pipe.use_camera_conditioning and pipe.use_image_conditioning and pipe.img_ref_scale == 0.3
    missing, unexpected = pipe.unet.load_state_dict(params, strict=False)
    assert not missing and not unexpected
    pipe = pipe.to("cuda")
    pipe.unet.eval()

    B, steps, gs = 2, 4, 3.0
    inp = make_inputs(cfg, B, 16, 7, seed=31, cam_dim=96)
    src = create_camera_matrix([0, 0, 2.0], [0, 0, 0]).unsqueeze(0)           # infer.py:97-103: one 3x4 pair
    tgt = create_camera_matrix([1.5, 0, 1.5], [0, 0, 0]).unsqueeze(0)
    g = torch.Generator().manual_seed(5)
    noises = [torch.randn(B, 4, 16, 16, generator=g) for _ in range(steps)]
    neg = torch.randn(B, 7, cfg.cross_attention_dim, generator=g)
    lat0 = torch.randn(B, 4, 16, 16, generator=g)
    projs = [inp["proj"]] * steps                                              # Q1 pinned: the same matrix every step
    want = OS.denoise_loop(params, cfg, pipe.scheduler.betas, inp["text"], neg, lat0, src, tgt, inp["lat"], steps, gs,
                           noises, projs, img_ref_scale=0.3, cam_modulation_strength=0.2)
    pipe.unet.fourier_projection = inp["proj"]
    pipe.unet.cache_reference = True
    seen = []
    out = pipe(prompt_embeds=inp["text"].cuda(), negative_prompt_embeds=neg.cuda(), num_inference_steps=steps,
               guidance_scale=gs, latents=lat0.cuda(), source_camera=src, target_camera=tgt,
               source_image_latents=inp["lat"].cuda(), output_type="latent", noise_per_step=[n.cuda() for n in noises],
               callback=lambda i, t, l: seen.append((i, int(t))), callback_steps=2, ref_scale=0.1)
    got = out["images"]
    assert [i for i, _ in seen] == [0, 2]
    assert torch.isfinite(got).all() and got.shape == lat0.shape
    assert rel_l2(got, want) <= 4e-2, rel_l2(got, want)
    # return_dict=False returns the bare tensor; asking for pixels without a VAE fails loudly
    from mvd_amd._lib import MvdError
    with pytest.raises(MvdError, match="no VAE"):
        pipe(prompt_embeds=inp["text"].cuda(), num_inference_steps=1, guidance_scale=1.0, latents=lat0.cuda(), output_type="pt")
    with pytest.raises(MvdError, match="text encoder"):
        pipe(prompt="a chair", num_inference_steps=1, guidance_scale=1.0, latents=lat0.cuda(), output_type="latent")


def test_reference_cache_never_serves_another_object(tiny):
    """two objects of the same shape denoised back to back with cache_reference=True.  The second call's
    tensors may land on the addresses the first call freed; its result must equal an uncached run."""
    from mvd_amd.pipeline import MVDDenoiser
    from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler
    from tests.parity_util import make_inputs
    cfg, params, model = tiny
    sched = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
    den = MVDDenoiser(model, sched)
    g = torch.Generator().manual_seed(9)
    noise = [torch.randn(1, 4, 16, 16, generator=g).cuda() for _ in range(2)]
    lat0 = torch.randn(1, 4, 16, 16, generator=g).cuda()

    def run(seed, cached):
        model.cache_reference = cached
        inp = make_inputs(cfg, 1, 16, 7, seed=seed, cam_dim=96)
        # fresh temporaries every call, freed on return: the caching allocator is free to recycle their addresses
        out = den(inp["text"].cuda().clone(), 2, 1.0, latents=lat0.clone(), source_image_latents=inp["lat"].cuda().clone(),
                  noise_per_step=noise)
        model.cache_reference = False
        return out

    a_ref, b_ref = run(41, False), run(42, False)
    a = run(41, True)
    b = run(42, True)
    assert torch.equal(a, a_ref)
    assert torch.equal(b, b_ref), "second object was denoised with the first object's cached reference K/V"
    assert not torch.equal(a, b)
    # same-address hazard without the denoiser: same storage, new contents, bumped version -> the cache must miss
    model.cache_reference = True
    i1 = make_inputs(cfg, 1, 16, 7, seed=43, cam_dim=96)
    i2 = make_inputs(cfg, 1, 16, 7, seed=44, cam_dim=96)
    lat, text, x = i1["lat"].cuda(), i1["text"].cuda(), i1["sample"].cuda()
    with torch.no_grad():
        model(x, torch.tensor(500), text, source_image_latents=lat)
        lat.copy_(i2["lat"].cuda())
        y = model(x, torch.tensor(500), text, source_image_latents=lat).sample
        model.reset_reference_cache()
        model.cache_reference = False
        y_ref = model(x, torch.tensor(500), text, source_image_latents=lat).sample
    assert torch.equal(y, y_ref)


def test_identical_encoder_weights_are_packed_once():
    """N4 (training.py:60-65, train_config.yaml:43): frozen base UNet == image-encoder UNet -> one packed copy, the
    encoder pass reads weight set 0; results equal the two-copy engine bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mvd_amd.config import UNetConfig
    from mvd_amd.mvd_unet import MultiViewUNet
    from oracle import mvd as OM
    from oracle import sd21_unet as OU
    from tests.parity_util import make_inputs
    ocfg = OU.UNetConfig.tiny()
    params = OM.init_mvd_params(ocfg, 3, cam_dim=96, cam_hidden=48, share_encoder=True)
    outs, nbytes = [], []
    for dedup in ("auto", False):
        m = MultiViewUNet(None, unet_config=UNetConfig.tiny(), init="empty", cam_output_dim=96, cam_hidden_dim=48,
                          dedup_encoder_weights=dedup)
        m.load_state_dict(params, strict=False)
        m = m.to("cuda").eval()
        inp = make_inputs(ocfg, 2, 16, 7, seed=2, cam_dim=96)
        m.fourier_projection = inp["proj"]
        with torch.no_grad():
            outs.append(m(inp["sample"].cuda(), torch.tensor(300), inp["text"].cuda(), source_camera=inp["src"].cuda(),
                          target_camera=inp["tgt"].cuda(), source_image_latents=inp["lat"].cuda()).sample.clone())
        assert m.encoder_weights_shared == (dedup == "auto")
        nbytes.append(m._engine.weight_bytes())
    assert torch.equal(outs[0], outs[1])
    assert nbytes[0] < 0.62 * nbytes[1], nbytes          # one UNet copy instead of two (+ adapter + camera)
    # different encoder weights are detected: no sharing
    params2 = OM.init_mvd_params(ocfg, 3, cam_dim=96, cam_hidden=48, share_encoder=False)
    m = MultiViewUNet(None, unet_config=UNetConfig.tiny(), init="empty", cam_output_dim=96, cam_hidden_dim=48)
    m.load_state_dict(params2, strict=False)
    m.to("cuda")._sync_engine()
    assert not m.encoder_weights_shared


def test_camera_conditioning_needs_four_input_channels():
    """in_channels != 4 with camera conditioning is rejected on the host before any launch."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mvd_amd._lib import MvdError
    from mvd_amd.config import UNetConfig
    from mvd_amd.mvd_unet import MultiViewUNet
    cfg = UNetConfig(in_channels=6, out_channels=4, block_out_channels=(64, 128), layers_per_block=1, num_heads=(1, 2),
                     cross_attention_dim=64, sample_size=8)
    m = MultiViewUNet(None, unet_config=cfg, cam_output_dim=96, cam_hidden_dim=48, use_image_conditioning=False).to("cuda")
    x, text = torch.randn(1, 6, 8, 8).cuda(), torch.randn(1, 5, 64).cuda()
    cam = torch.eye(4)[None].cuda()
    with torch.no_grad():
        assert torch.isfinite(m(x, torch.tensor(10), text).sample).all()           # without cameras: fine
        with pytest.raises(MvdError, match="in_channels == 4"):
            m(x, torch.tensor(10), text, source_camera=cam, target_camera=cam)


def test_hip_graph_replay_is_bit_exact(tiny):
    """whole forwards replayed as one hipGraphLaunch (``mvd_engine_set_graph``).  The first call of a
    kind runs normally, the second is captured, later ones replay; inputs are refreshed in place between calls.  Covers the
    cold forward, the cached pair (MVD_REUSE_REF has its own graph) and a return to plain launches -- all bit-identical to
    the un-graphed engine on the same inputs."""
    from tests.parity_util import make_inputs
    cfg, params, model = tiny
    dev = "cuda"
    runs = []
    for seed in (21, 22, 23, 24):
        inp = make_inputs(cfg, 2, 16, 7, seed=seed, cam_dim=96)
        runs.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()})

    def fwd(inp, t):
        model.fourier_projection = inp["proj"]
        with torch.no_grad():
            return model(inp["sample"], torch.tensor(t), inp["text"], source_camera=inp["src"], target_camera=inp["tgt"],
                         source_image_latents=inp["lat"]).sample.clone()

    try:
        model.use_hip_graph = False
        model.cache_reference = False
        want = [fwd(inp, 100 + 50 * i) for i, inp in enumerate(runs)]
        model.use_hip_graph = True
        got = [fwd(inp, 100 + 50 * i) for i, inp in enumerate(runs)]        # run, capture, replay, replay
        for g, w in zip(got, want):
            assert torch.equal(g, w)
        # cached mode: the first call of an object recomputes the reference K/V, the following ones reuse it
        model.cache_reference = True
        model.reset_reference_cache()
        inp = runs[0]
        cached = [fwd(inp, t) for t in (900, 600, 300, 100, 50)]
        model.use_hip_graph = False
        model.reset_reference_cache()
        plain = [fwd(inp, t) for t in (900, 600, 300, 100, 50)]
        for g, w in zip(cached, plain):
            assert torch.equal(g, w)
    finally:
        model.use_hip_graph = False
        model.cache_reference = False
        model.reset_reference_cache()


def test_infer_py_call_sequence_from_cached_hub_name(tmp_path, shim_path):
    """a17 + N1 end to end: the call sequence of infer.py:33-122 through the integration/ shims with a hub NAME whose
    snapshot sits in a (fake, tiny) huggingface cache -- pipeline from the name + cache_dir, checkpoint key rewriting,
    .to(device), the eval() calls, create_camera_matrix, then pipeline(prompt=..., source_images=..., output_type="pt")
    with the snapshot's own CLIP text encoder and VAE.  The denoising loop itself is checked against the oracle above; here
    the result must be finite, of the image shape, reproducible, and sensitive to the checkpoint that was loaded."""
    pytest.importorskip("transformers")
    from src.models.mvd_unet import create_mvd_pipeline                          # infer.py:1
    from src.utils import create_camera_matrix                                  # infer.py:3
    from tests.hub_fixture import REPO, build_fake_hf_cache
    cache, snap, sds = build_fake_hf_cache(str(tmp_path))
    device = torch.device("cuda")

    def build(perturb):
        pipeline = create_mvd_pipeline(pretrained_model_name_or_path=REPO, use_memory_efficient_attention=True,
                                       enable_gradient_checkpointing=False, dtype=torch.float32, use_camera_conditioning=True,
                                       use_image_conditioning=True, simple_cam_encoder=False, cache_dir=str(tmp_path),
                                       cam_output_dim=96, cam_hidden_dim=48)
        # infer.py:46-69: a Lightning checkpoint ("unet." prefix; old files name the encoder "image_encoder.<key>")
        g = torch.Generator().manual_seed(3)
        ckpt = {}
        for k, v in pipeline.unet.state_dict().items():
            w = v.detach().cpu().clone()                     # (torch.load(..., map_location="cpu"), infer.py:46)
            if perturb and k.endswith("proj_in.weight"):
                w = w + 0.5 * torch.randn(w.shape, generator=g)
            ck = k.replace("image_encoder.unet.", "image_encoder.", 1) if k.startswith("image_encoder.unet.") else k
            ckpt["unet." + ck] = w
        unet_keys = {k: v for k, v in ckpt.items() if k.startswith("unet.")}
        sd = {k.replace("unet.", "", 1): v for k, v in unet_keys.items()}
        fixed = {}
        for key, value in sd.items():
            nk = key
            if key.startswith("image_encoder.") and not key.startswith("image_encoder.unet."):
                nk = "image_encoder.unet." + key.split(".", 1)[1]
            fixed[nk] = value
        missing, unexpected = pipeline.unet.load_state_dict(fixed, strict=False)
        assert not missing and not unexpected
        pipeline = pipeline.to(device)
        pipeline.unet.eval(); pipeline.vae.eval(); pipeline.text_encoder.eval()
        pipeline.unet.image_encoder.eval(); pipeline.unet.camera_encoder.eval()
        return pipeline

    src = create_camera_matrix([0, 0, 2.0], [0, 0, 0]).unsqueeze(0).to(device)
    tgt = create_camera_matrix([1.5, 0, 1.5], [0, 0, 0]).unsqueeze(0).to(device)
    image = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(device)      # load_image's [0, 1] tensor

    def run(pipeline):
        torch.manual_seed(11)                      # (Q1's projection and the ancestral noise come from torch's global RNG)
        with torch.no_grad():
            out = pipeline(prompt="a photo of a red chair", num_inference_steps=3, source_camera=src, target_camera=tgt,
                           source_images=image, guidance_scale=1.0, ref_scale=0.1, output_type="pt",
                           generator=torch.Generator(device="cuda").manual_seed(5))
        return out["images"]

    p0 = build(False)
    a, b = run(p0), run(p0)
    assert a.shape == (1, 3, 32, 32) and torch.isfinite(a).all() and float(a.min()) >= 0 and float(a.max()) <= 1
    assert torch.equal(a, b)
    c = run(build(True))
    assert not torch.allclose(a, c)                # the loaded checkpoint, not the snapshot's base weights, produced the image
