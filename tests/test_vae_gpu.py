"""Row N3 on the GPU: AutoencoderKLHIP (C ABI ``mvd_vae_encode`` / ``mvd_vae_decode``) vs the CPU oracle oracle/vae.py on
identical seeded weights and inputs, and the ``MVDPipeline`` entry / exit paths that use it
(/root/reference/src/models/pipeline.py:100-117, 168-181).

Tolerance: bf16 storage / fp32 accumulate through ~60 (tiny) to ~120 (SD-2.1) chained ops against an fp32 oracle:
rel-L2 <= 2e-2 and max-abs <= 5e-2 * max|ref| (the UNet's end-to-end tolerance)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _pair(kind="tiny", seed=0):
    from mvd_amd.vae import AutoencoderKLHIP, VAEConfig
    from oracle import vae as OV
    ocfg = OV.VAEConfig.tiny() if kind == "tiny" else OV.VAEConfig.sd21()
    hcfg = VAEConfig(block_out_channels=ocfg.block_out_channels, layers_per_block=ocfg.layers_per_block)
    p = OV.init_params(ocfg, seed)
    m = AutoencoderKLHIP(hcfg)
    missing, unexpected = m.load_state_dict(p, strict=True)
    assert not missing and not unexpected
    return ocfg, p, m.to("cuda").eval()


def _cmp(got, want, what):
    got, want = got.float().cpu(), want.float()
    assert got.shape == want.shape and torch.isfinite(got).all(), what
    rel = ((got - want).norm() / want.norm()).item()
    mx = ((got - want).abs().max() / want.abs().max()).item()
    assert rel <= 2e-2 and mx <= 5e-2, (what, rel, mx)
    return rel


def test_asymmetric_pad_stride2_conv_op():
    """Downsample2D(padding=0): F.pad(x, (0,1,0,1)) + stride-2 pad-0 conv, both GEMM kernels (B=32 -> the ping-pong tile)."""
    from mvd_amd import ops
    from mvd_amd.packing import _conv_w
    for B, H, cin, cout in ((2, 16, 64, 128), (32, 32, 128, 320)):
        g = torch.Generator().manual_seed(B)
        x = torch.randn(B, cin, H, H, generator=g).to(torch.bfloat16)
        w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(torch.bfloat16)
        b = torch.randn(cout, generator=g)
        want = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b, stride=2).permute(0, 2, 3, 1)
        got = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), _conv_w(w).to(torch.bfloat16).cuda(), b.cuda(), stride=2, asym_pad=True)
        assert (got.float().cpu() - want).abs().max().item() <= 2 ** -7 * want.abs().max().item(), (B, H)
    from mvd_amd._lib import MvdError
    with pytest.raises(MvdError, match="even input size"):
        ops.conv3x3(torch.zeros(1, 7, 7, 64, dtype=torch.bfloat16, device="cuda"), torch.zeros(64, 576, dtype=torch.bfloat16, device="cuda"),
                    stride=2, asym_pad=True)


@pytest.mark.parametrize("batch,hw", [(1, 32), (3, (32, 64))])
def test_tiny_vae_encode_decode_parity(batch, hw):
    from oracle import vae as OV
    ocfg, p, m = _pair("tiny", 0)
    H, W = (hw, hw) if isinstance(hw, int) else hw
    g = torch.Generator().manual_seed(7)
    img = torch.randn(batch, 3, H, W, generator=g).clamp(-1, 1)
    want_m = OV.encode_moments(p, ocfg, img)
    dist = m.encode(img.cuda()).latent_dist
    _cmp(dist.parameters, want_m, "moments")
    noise = torch.randn(batch, 4, H // 4, W // 4, generator=g)
    z = dist.sample(noise=noise.cuda())
    torch.testing.assert_close(z.cpu(), OV.sample_latents(dist.parameters.cpu(), noise), rtol=1e-5, atol=1e-5)   # the sampling kernel itself
    assert torch.equal(dist.mode(), dist.parameters[:, :4])
    lat = 0.5 * torch.randn(batch, 4, H // 4, W // 4, generator=g)
    _cmp(m.decode(lat.cuda()).sample, OV.decode(p, ocfg, lat), "decoded image")


def test_sd21_vae_full_size_parity():
    """The real SD-2.1 VAE shapes (83.65 M parameters), one 512x512 image: encoder to 64x64x8 moments, decoder back."""
    from oracle import vae as OV
    ocfg, p, m = _pair("sd21", 1)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(1, 3, 512, 512, generator=g).clamp(-1, 1)
    _cmp(m.encode(img.cuda()).latent_dist.parameters, OV.encode_moments(p, ocfg, img), "sd21 moments")
    lat = torch.randn(1, 4, 64, 64, generator=g)
    _cmp(m.decode(lat.cuda()).sample, OV.decode(p, ocfg, lat), "sd21 decoded image")


def test_pipeline_with_vae_source_images_and_pixel_output():
    """MVDPipeline with a VAE attached: ``source_images`` in [0,1] are rescaled, repeated to the batch, encoded and sampled
    (pipeline.py:100-117); ``output_type='pt'`` decodes latents / scaling_factor and maps to [0,1] (pipeline.py:168-181)."""
    from mvd_amd.pipeline import MVDPipeline
    from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler
    from oracle import vae as OV
    from tests.parity_util import build_pair, make_inputs
    cfg, params, unet = build_pair("tiny", 0, 96, 48)
    ocfg, vp, vae = _pair("tiny", 2)
    sched = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
    pipe = MVDPipeline(unet, sched, vae=vae, vae_scale_factor=4)
    inp = make_inputs(cfg, 2, 16, 7, seed=5, cam_dim=96)
    g = torch.Generator().manual_seed(11)
    src01 = torch.rand(1, 3, 64, 64, generator=g)                         # one source image in [0,1] for a batch of 2
    torch.manual_seed(123)                                                # latent_dist.sample() draws from the global generator
    out = pipe(prompt_embeds=inp["text"].cuda(), num_inference_steps=2, guidance_scale=1.0, height=64, width=64,
               source_images=src01, source_camera=inp["src"][:1], target_camera=inp["tgt"][:1], output_type="pt",
               generator=torch.Generator(device="cuda").manual_seed(1))
    img = out["images"]
    assert img.shape == (2, 3, 64, 64) and torch.isfinite(img).all() and float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    # the encode half against the oracle: same rescale / repeat / sample / scaling factor
    want_m = OV.encode_moments(vp, ocfg, (2 * src01 - 1).repeat(2, 1, 1, 1))
    got_m = vae.encode((2 * src01 - 1).repeat(2, 1, 1, 1).cuda()).latent_dist.parameters
    _cmp(got_m, want_m, "pipeline source-image moments")
