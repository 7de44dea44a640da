"""The VAE oracle (oracle/vae.py, diffusers AutoencoderKL restated -- parity unpinned) and the host side of
mvd_amd/vae.py without a GPU: parameter-count identity, state-dict key schema, slot packing, loud failure."""
import numpy as np
import pytest
import torch

from oracle import vae as OV


def test_sd21_vae_parameter_count_identity():
    """AutoencoderKL of SD-2.x has 83,653,863 parameters; the restated layer table reproduces it exactly."""
    shapes = OV.param_shapes(OV.VAEConfig.sd21())
    assert sum(int(np.prod(s)) for s in shapes.values()) == 83_653_863
    assert len(shapes) == 248


def test_hip_vae_state_dict_schema_matches_diffusers_keys():
    from mvd_amd.vae import AutoencoderKLHIP, VAEConfig
    for ocfg, hcfg in ((OV.VAEConfig.sd21(), VAEConfig()),
                       (OV.VAEConfig.tiny(), VAEConfig(block_out_channels=(64, 64, 128), layers_per_block=1))):
        want = OV.param_shapes(ocfg)
        with torch.device("meta"):
            m = AutoencoderKLHIP(hcfg)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == want


def test_pack_vae_slots():
    from mvd_amd.vae import VAEConfig, pack_vae
    ocfg = OV.VAEConfig.tiny()
    sd = OV.init_params(ocfg, 0)
    slots = pack_vae(sd, VAEConfig(block_out_channels=(64, 64, 128), layers_per_block=1), "cpu")
    assert slots["encoder.conv_in.w"].shape == (64, 64) and slots["encoder.conv_in.w"].dtype == torch.bfloat16
    # conv2 of a channel-changing resnet carries the 1x1 shortcut along K, biases summed
    k = "encoder.down_blocks.2.resnets.0"
    assert slots[f"{k}.conv2.w"].shape == (128, 9 * 128 + 64)
    torch.testing.assert_close(slots[f"{k}.conv2.b"], sd[f"{k}.conv2.bias"] + sd[f"{k}.conv_shortcut.bias"])
    assert slots["encoder.conv_out.w"].shape == (8, 9 * 128) and slots["decoder.conv_out.w"].shape == (3, 9 * 64)
    assert slots["quant_conv.w"].shape == (8, 8) and slots["quant_conv.w"].dtype == torch.float32
    assert slots["decoder.mid_block.attn.q.w"].shape == (128, 128)


def test_oracle_vae_round_trip_shapes_and_sampling():
    cfg = OV.VAEConfig.tiny()
    p = OV.init_params(cfg, 1)
    g = torch.Generator().manual_seed(0)
    img = torch.randn(2, 3, 32, 32, generator=g)
    mom = OV.encode_moments(p, cfg, img)
    assert mom.shape == (2, 8, 8, 8)
    nz = torch.randn(2, 4, 8, 8, generator=g)
    z = OV.sample_latents(mom, nz)
    mean, logvar = mom.chunk(2, 1)
    torch.testing.assert_close(z, mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * nz)
    assert OV.decode(p, cfg, z).shape == (2, 3, 32, 32)


def test_hip_vae_fails_loudly_without_gpu():
    from mvd_amd._lib import MvdError
    from mvd_amd.vae import AutoencoderKLHIP, VAEConfig
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    m = AutoencoderKLHIP(VAEConfig(block_out_channels=(64, 64, 128), layers_per_block=1))
    with pytest.raises(MvdError, match="no CPU fallback"):
        m.encode(torch.zeros(1, 3, 32, 32))
