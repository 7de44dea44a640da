"""a17 (mvd_unet.py:388-453, infer.py:33-90): a hub NAME resolves offline to its cached snapshot, the components come from it,
and a name nothing local answers to raises instead of random-initialising."""
import os

import pytest
import torch

from tests.hub_fixture import REPO, REV, build_fake_hf_cache


def test_resolve_snapshot_rules(tmp_path, monkeypatch):
    from mvd_amd._lib import MvdError
    from mvd_amd.hub import resolve_snapshot
    cache, snap, _ = build_fake_hf_cache(str(tmp_path), with_text_encoder=False, with_vae=False)
    assert resolve_snapshot(None) is None                                   # checkpoint-free construction
    assert resolve_snapshot(snap) == snap                                   # a directory is itself
    assert resolve_snapshot(REPO, cache_dir=cache) == snap                  # cache_dir = the hub cache (from_pretrained's meaning)
    assert resolve_snapshot(REPO, cache_dir=str(tmp_path)) == snap          # cache_dir = HF_HOME (what infer.py:24-29 passes)
    assert resolve_snapshot(REPO, cache_dir=cache, revision=REV) == snap
    monkeypatch.setenv("HF_HOME", str(tmp_path))
    assert resolve_snapshot(REPO) == snap                                   # the environment's cache
    monkeypatch.delenv("HF_HOME")
    monkeypatch.setenv("HOME", str(tmp_path / "nohome"))
    with pytest.raises(MvdError, match="no cached snapshot"):
        resolve_snapshot(REPO)
    with pytest.raises(MvdError, match="no cached snapshot"):
        resolve_snapshot("stabilityai/stable-diffusion-2-1", cache_dir=cache)
    with pytest.raises(MvdError):
        resolve_snapshot(REPO, cache_dir=cache, revision="deadbeef")
    assert resolve_snapshot("nobody/nothing", cache_dir=cache, required=False) is None


def test_named_model_never_random_initialises(tmp_path):
    from mvd_amd._lib import MvdError
    from mvd_amd.mvd_unet import MultiViewUNet, create_mvd_pipeline
    with pytest.raises(MvdError, match="no cached snapshot"):
        MultiViewUNet("stabilityai/stable-diffusion-2-1", cache_dir=str(tmp_path))
    with pytest.raises(MvdError, match="no cached snapshot"):
        create_mvd_pipeline("stabilityai/stable-diffusion-2-1", cache_dir=str(tmp_path))
    # a snapshot directory without UNet weights is an error too
    os.makedirs(tmp_path / "empty" / "unet")
    with pytest.raises(MvdError, match="no unet/diffusion_pytorch_model.safetensors"):
        MultiViewUNet(str(tmp_path / "empty"))


def test_create_mvd_pipeline_from_cached_hub_name(tmp_path):
    """The call of infer.py:33-44 with a hub name and the cache directory: every component comes from the snapshot."""
    transformers = pytest.importorskip("transformers")     # noqa: F841
    from mvd_amd.mvd_unet import MultiViewUNet, create_mvd_pipeline
    from mvd_amd.pipeline import MVDPipeline
    from mvd_amd.scheduler import ShiftSNRScheduler
    cache, snap, sds = build_fake_hf_cache(str(tmp_path))
    pipe = create_mvd_pipeline(pretrained_model_name_or_path=REPO, use_memory_efficient_attention=True,
                               enable_gradient_checkpointing=False, dtype=torch.float32, use_camera_conditioning=True,
                               use_image_conditioning=True, simple_cam_encoder=False, cache_dir=str(tmp_path),
                               cam_output_dim=96, cam_hidden_dim=48)
    assert isinstance(pipe, MVDPipeline) and isinstance(pipe.unet, MultiViewUNet)
    assert pipe.unet.pretrained_snapshot == snap
    assert tuple(pipe.unet.unet_config.block_out_channels) == (64, 128, 128, 128) and pipe.unet.config.sample_size == 16
    got = pipe.unet.base_unet.state_dict()
    for k, v in sds["unet"].items():
        assert torch.equal(got[k], v), k                     # the snapshot's weights, not a random initialisation
    enc = pipe.unet.image_encoder.unet.state_dict()
    assert all(torch.equal(enc[k], v) for k, v in sds["unet"].items())      # image_encoder.py:18-22 loads the same snapshot
    assert pipe.vae is not None and pipe.text_encoder is not None and pipe.tokenizer is not None
    vsd = pipe.vae.state_dict()
    assert all(torch.equal(vsd[k], v) for k, v in sds["vae"].items())
    assert isinstance(pipe.scheduler, ShiftSNRScheduler) or hasattr(pipe.scheduler, "betas")
    assert pipe.scheduler.config.prediction_type == "v_prediction"          # scheduler_config.json of the snapshot
    # infer.py:76-90
    pipe.unet.eval(); pipe.vae.eval(); pipe.text_encoder.eval()
    pipe.unet.image_encoder.eval(); pipe.unet.camera_encoder.eval()
    ids = pipe.tokenizer("a photo of a red chair", padding="max_length", max_length=pipe.tokenizer.model_max_length,
                         truncation=True, return_tensors="pt").input_ids
    assert ids.shape == (1, 77) and int(ids[0, 0]) == 0
    emb = pipe.text_encoder(ids)[0]
    assert emb.shape == (1, 77, 128)
