"""Host-side behaviour added in round 3 (CPU only): the self-launching bench entry, the caller-side contract of the
denoise loop (scheduler.step gets no generator, pipeline.py:161), the deprecated VAE attention key names."""
import json
import os
import subprocess
import sys
from types import SimpleNamespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher in front: the parent starts torch.distributed.run, two gloo ranks run
    the control flow (arena broadcast, barriers, max over ranks) and rank 0's single JSON line comes back with n_gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--launch-dry-run"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["weight_broadcast"]["bytes"] > 0 and line["broadcast_ok"] is True
    assert line["ms_per_step"] >= 20.0        # the max over ranks (rank 1 sleeps 20 ms), not rank 0's own 10 ms
    # round 4: who took part.  Two ranks = two processes with their LOCAL_RANKs; the collective library is named
    assert line["weight_broadcast"]["backend"] == "gloo" and line["weight_broadcast"]["rehearsal"] is True
    seen = line["ranks_seen"]
    assert [d["local_rank"] for d in seen] == [0, 1] and len({d["pid"] for d in seen}) == 2
    assert line["distinct_gpus"] == 2         # (no GPU here: distinct processes stand in for distinct devices)


def test_device_identity_helpers():
    from mvd_amd import distributed as D
    assert D.collective_library() == "none"                         # no process group in this process
    assert D.gather_identities(3)[0]["local_rank"] == 3
    same = [{"local_rank": 0, "uuid": "GPU-a", "device": 0}, {"local_rank": 1, "uuid": "GPU-a", "device": 0}]
    two = [{"local_rank": 0, "uuid": "GPU-a", "device": 0}, {"local_rank": 1, "uuid": "GPU-b", "device": 1}]
    assert D.distinct_devices(same) == 1 and D.distinct_devices(two) == 2
    assert D.distinct_devices([{"device": 0}, {"device": 1}, {"device": 1}]) == 2
    # a runtime that reports ONE uuid for every device must not make eight GPUs count as one
    clones = [{"local_rank": r, "uuid": "GPU-a", "pci_bus_id": str(10 + r), "device": r} for r in range(8)]
    assert D.distinct_devices(clones) == 8
    assert D.distinct_devices([{"local_rank": 0, "pid": 11}, {"local_rank": 1, "pid": 12}]) == 2    # CPU rehearsal
    D.check_enough_devices(8, rehearsal=True)                        # a rehearsal may oversubscribe one GPU
    D.check_enough_devices(8)                                        # no GPU visible: nothing to check (CPU control-flow tests)


def test_bench_flop_counter_reproduces_survey_figures():
    sys.path.insert(0, ROOT)
    import bench
    assert abs(bench.unet_flops(64, 64) - 804.26e9) < 1e7
    assert abs(bench.unet_flops(64, 64, adapter=True) - 1151.59e9) < 1e7
    assert bench.unet_flops(96, 96) > 2.25 * bench.unet_flops(64, 64)      # self-attention grows with tokens squared


def test_denoise_loop_does_not_hand_the_generator_to_scheduler_step():
    """/root/reference/src/models/pipeline.py:161 is ``self.scheduler.step(noise_pred, t, latents)``: the caller's generator
    seeds the initial latents only; the ancestral noise comes from the global RNG."""
    from mvd_amd.pipeline import MVDDenoiser
    seen = []

    class Sched:
        init_noise_sigma = 1.0
        timesteps = torch.tensor([9, 5, 1])

        def set_timesteps(self, n):
            pass

        def step(self, model_output, t, sample, **kw):
            seen.append(kw)
            return SimpleNamespace(prev_sample=sample - 0.1 * model_output)

    class UNet:
        def _exec_device(self):
            return torch.device("cpu")

        def __call__(self, sample, timestep, encoder_hidden_states, **kw):
            return SimpleNamespace(sample=sample * 0.5)

    g = torch.Generator().manual_seed(3)
    out = MVDDenoiser(UNet(), Sched())(torch.zeros(2, 7, 16), num_inference_steps=3, guidance_scale=1.0, generator=g,
                                       height=8, width=8)
    assert out.shape == (2, 4, 8, 8) and len(seen) == 3
    assert all("generator" not in kw or kw["generator"] is None for kw in seen)
    want = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(3))
    for _ in range(3):
        want = want - 0.1 * (want * 0.5)
    assert torch.allclose(out, want)


def test_vae_accepts_deprecated_attention_key_names():
    """The published SD-2.1 VAE file uses mid_block.attentions.0.{query,key,value,proj_attn} (conv-shaped in the oldest
    files); diffusers renames them on load and so must the mirror (strict load)."""
    from mvd_amd.vae import AutoencoderKLHIP, VAEConfig
    cfg = VAEConfig(block_out_channels=(32, 64), layers_per_block=1, norm_num_groups=8)
    a, b = AutoencoderKLHIP(cfg), AutoencoderKLHIP(cfg)
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    old = {}
    for k, v in a.state_dict().items():
        for new, dep in ren.items():
            if f".attentions.0.{new}." in k:
                k = k.replace(f".{new}.", f".{dep}.")
                if k.endswith(".weight"):
                    v = v.reshape(*v.shape, 1, 1)
        old[k] = v
    assert any(".query." in k for k in old) and not any(".to_q." in k for k in old)
    missing, unexpected = b.load_state_dict(old, strict=True)
    assert not missing and not unexpected
    for (k1, v1), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_vae_snapshot_that_fails_to_load_is_an_error_not_a_missing_vae(tmp_path):
    from mvd_amd import _lib as L
    from mvd_amd.pipeline import _optional_components
    from safetensors.torch import save_file
    d = tmp_path / "snap" / "vae"
    d.mkdir(parents=True)
    (d / "config.json").write_text(json.dumps({"block_out_channels": [32, 64], "layers_per_block": 1, "norm_num_groups": 8}))
    save_file({"bogus.weight": torch.zeros(3)}, str(d / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(L.MvdError, match="failed to load"):
        _optional_components(str(tmp_path / "snap"), torch.float32)


@pytest.mark.parametrize("name", ["gemm_sm", "gemm_pp", "gemm_xs", "conv_ws", "attention"])
def test_hand_pipelined_gemms_use_no_scratch_memory(name):
    """hipcc turns a runtime-indexed register array -- or one live value too many in a 256-register kernel -- into scratch_load /
    scratch_store, which also count on vmcnt beside the hand-counted LDS-DMA waits: the small-M GEMM, the ping-pong GEMM, the X-stationary GEMM
    (every instantiation, the split-K forms with their in-kernel combine included) must compile without a single scratch
    instruction and without spills."""
    import tempfile
    src = os.path.join(ROOT, "mvd_amd", "csrc", name + ".hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, name + ".s")
        from mvd_amd import _build as B        # the product's own code-generation flags (packed fp32 selection off)
        r = subprocess.run([B.HIPCC, *B.FLAGS, "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        asm = open(out).read()
    assert "scratch_load" not in asm and "scratch_store" not in asm
    import re
    assert not re.search(r"\.vgpr_spill_count:\s*[1-9]", asm) and not re.search(r"\.private_segment_fixed_size:\s*[1-9]", asm)


def test_built_library_holds_no_packed_fp32_arithmetic():
    """Round 4 traced the four-pixel conv_out's run-to-run differences on a shared GPU to hipcc's SLP-vectorised
    `v_pk_fma_f32 ... op_sel:[0,1,0]`: the product is built with packed fp32 selection off, and the library that
    ships (the one build() just produced) is disassembled here to prove no v_pk_{fma,mul,add}_f32 is left in any code object."""
    import importlib.util
    from mvd_amd import _build as B
    lib = B.build()
    spec = importlib.util.spec_from_file_location("lint_device_isa", os.path.join(ROOT, "tools", "lint_device_isa.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    res = lint.count_packed_fp32(lib)
    assert res["code_objects"] == len(B.SOURCES) and res["instructions"] > 100000
    assert res["packed_fp32"] == 0, res


def test_block_weight_layout_is_the_lds_image():
    """packing.block_weight: [N/32][K/64] blocks of 32 rows x 128 bytes, chunk c of row r in slot c ^ ((r >> 1) & 7)."""
    from mvd_amd.packing import block_weight
    n, k = 96, 192
    w = torch.arange(n * k, dtype=torch.float32).reshape(n, k)
    b = block_weight(w).reshape(-1)
    for nn_ in (0, 5, 31, 32, 77, 95):
        for kk in (0, 7, 8, 63, 64, 100, 191):
            r, c = nn_ % 32, (kk % 64) // 8
            off = ((nn_ // 32) * (k // 64) + kk // 64) * 2048 + r * 64 + (c ^ ((r >> 1) & 7)) * 8 + kk % 8
            assert b[off] == w[nn_, kk]


def test_pack_camera_concatenates_the_hooked_modulators():
    """cam.modcat.*: down_0..3, up_0..3, output (the engine's hook order; `mid` is never addressed, Q3), layer by layer."""
    from mvd_amd.camera_encoder import CameraEncoder
    from mvd_amd.packing import pack_camera
    dims = {"down_0": 320, "down_1": 640, "down_2": 1280, "down_3": 1280, "mid": 1280, "up_0": 1280, "up_1": 1280, "up_2": 640,
            "up_3": 320, "output": 4}
    enc = CameraEncoder(output_dim=64, hidden_dim=32, modulation_hidden_dims=dims)
    sd = enc.state_dict()
    out = pack_camera(sd, "cpu", 4)
    order = [f"down_{i}" for i in range(4)] + [f"up_{i}" for i in range(4)] + ["output"]
    assert out["cam.modcat.w0"].shape == (9 * 32, 64) and out["cam.modcat.g1"].shape == (9 * 32,)
    tot = sum(2 * dims[n] for n in order)
    assert out["cam.modcat.w3"].shape == (tot, 32) and out["cam.modcat.b3"].shape == (tot,)
    off = 0
    for i, n in enumerate(order):
        assert torch.equal(out["cam.modcat.w0"][i * 32:(i + 1) * 32], sd[f"modulators.{n}.0.weight"].float())
        assert torch.equal(out["cam.modcat.w3"][off:off + 2 * dims[n]], sd[f"modulators.{n}.3.weight"].float())
        off += 2 * dims[n]
