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
    # the exact byte count of the two arenas (300 x 64 bf16 weights + 300 fp32 biases, each slot padded to the arena's 256-byte grid)
    from mvd_amd import distributed as D
    arenas, _ = D.pack_into_arenas({"a.w": torch.zeros(300, 64, dtype=torch.bfloat16), "a.b": torch.zeros(300)})
    assert line["weight_broadcast"]["bytes"] == sum(a.numel() * a.element_size() for a in arenas.values()) >= 300 * 64 * 2 + 300 * 4
    assert line["broadcast_ok"] is True and line["weight_broadcast"]["buckets"] == len(arenas)
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
    # with real (differing) UUIDs the key is (host, UUID) alone -- one physical GPU reached through two visibility
    # masks / device indices counts ONCE, and equal tuples on two hosts count twice; the identity carries the host name
    masks = [{"host": "n0", "uuid": "GPU-a", "device": 0}, {"host": "n0", "uuid": "GPU-a", "device": 1},
             {"host": "n0", "uuid": "GPU-b", "device": 0}]
    assert D.distinct_devices(masks) == 2
    hosts = [{"host": "n0", "uuid": "GPU-a", "device": 0}, {"host": "n1", "uuid": "GPU-a", "device": 0},
             {"host": "n1", "uuid": "GPU-b", "device": 1}]
    assert D.distinct_devices(hosts) == 3
    nouuid = [{"host": "n0", "pci_bus_id": "5", "device": 0}, {"host": "n1", "pci_bus_id": "5", "device": 0}]
    assert D.distinct_devices(nouuid) == 2
    zeros = [{"host": "n0", "uuid": "00000000-0000-0000-0000-000000000000", "device": r} for r in range(4)] + \
            [{"host": "n0", "uuid": "GPU-b", "device": 4}]
    assert D.distinct_devices(zeros) == 5           # a degenerate UUID among them: every rank falls back to the PCI / index key
    import socket
    assert D.device_identity(0)["host"] == socket.gethostname()
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


def _lint():
    import importlib.util
    spec = importlib.util.spec_from_file_location("lint_device_isa", os.path.join(ROOT, "tools", "lint_device_isa.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    return lint


def test_built_library_keeps_the_hand_scheduling_invariants():
    """tools/lint_device_isa.py on the library that ships (the one build() just produced) and on the compiler's assembly of
    conv_ws.hip / attention.hip -- invariants I1..I5 of DESIGN.md section 0: no packed fp32 arithmetic (the precaution of DESIGN 4.3), no
    scratch / VGPR spills in any shipped kernel, every conv_ws ring-slot refill behind the retired read of that slot, every
    asm-issued MFMA behind its own s_nop, every SGPR-soffset 16-byte store followed by its wait states."""
    from mvd_amd import _build as B
    lint = _lint()
    lib = B.build()
    summary, bad = lint.lint_all(lib)
    assert summary["code_objects"] == len(B.SOURCES) and summary["instructions"] > 100000 and summary["kernels"] > 100, summary
    assert summary["packed_fp32"] == 0 and summary["conv_ws_refill_fences"] >= 50 and summary["asm_issued_mfma"] >= 4, summary
    assert summary["soffset_stores"] > 100, summary
    assert not bad, bad[:10]
    assert lint.count_packed_fp32(lib)["packed_fp32"] == 0


def test_isa_lint_catches_what_it_is_there_for():
    """The lint's own negative cases, on hand-written assembly: a ring refill hoisted above the retirement of its slot's read (round
    4's conv_ws race), a wait that is too weak, a scalar load in flight, a missing fence, an asm MFMA without its s_nop, a
    rewritten store-data register without wait states, a kernel with scratch."""
    lint = _lint()
    ok = """
kern:
\tds_read_b128 v[48:51], v47 offset:2048
\tds_read_b128 v[56:59], v70 offset:18432
\ts_waitcnt lgkmcnt(1)
\tv_mfma_f32_16x16x32_bf16 a[0:3], v[48:51], v[56:59], a[0:3]
\t;;#ASMSTART
\t; MVD_REFILL_FENCE v[48:51]
\t;;#ASMEND
\tbuffer_load_dwordx4 v2, s[4:7], s10 offen lds
"""
    assert lint.check_refill_fences(ok) == (1, [])
    hoisted = ok.replace("\ts_waitcnt lgkmcnt(1)\n", "")                       # the DMA's fence right behind the read's ISSUE
    n, bad = lint.check_refill_fences(hoisted)
    assert n == 1 and len(bad) == 1 and "has not retired" in bad[0]
    weak = ok.replace("lgkmcnt(1)", "lgkmcnt(2)")                               # two may still be in flight: the slot's read too
    assert "has not retired" in lint.check_refill_fences(weak)[1][0]
    smem = ok.replace("\ts_waitcnt lgkmcnt(1)", "\ts_load_dword s3, s[0:1], 0x0\n\ts_waitcnt lgkmcnt(1)")
    assert "scalar load" in lint.check_refill_fences(smem)[1][0]
    assert "only 0 MVD_REFILL_FENCE" in lint.check_refill_fences(ok.replace("; MVD_REFILL_FENCE v[48:51]", ""))[1][0]
    other = ok.replace("MVD_REFILL_FENCE v[48:51]", "MVD_REFILL_FENCE v[90:93]")
    assert "no ds_read filled it" in lint.check_refill_fences(other)[1][0]

    asm_ok = "\t;;#ASMSTART\n\ts_nop 1\n\tv_mfma_f32_32x32x16_bf16 v[68:83], v[56:59], v[84:87], v[36:51]\n\t;;#ASMEND\n"
    assert lint.check_asm_mfma(asm_ok) == (1, [])
    assert len(lint.check_asm_mfma(asm_ok.replace("\ts_nop 1\n", ""))[1]) == 1
    assert len(lint.check_asm_mfma(asm_ok.replace("s_nop 1", "s_nop 0"))[1]) == 1
    assert "only 0 asm-issued" in lint.check_asm_mfma("\tv_mfma_f32_32x32x16_bf16 v[0:15], v[56:59], v[84:87], v[0:15]\n")[1][0]

    st = "buffer_store_dwordx4 v[156:159], v164, s[8:11], s40 offen"
    assert lint.check_store_hazard([st, "s_nop 1", "v_mov_b32_e32 v157, v3"], "k") == []
    assert lint.check_store_hazard([st, "s_waitcnt lgkmcnt(2)", "v_add_f32_e32 v1, v2, v3", "v_mov_b32_e32 v157, v3"], "k") == []
    assert len(lint.check_store_hazard([st, "v_mov_b32_e32 v157, v3"], "k")) == 1
    assert len(lint.check_store_hazard([st, "s_nop 0", "ds_read_b128 v[156:159], v1"], "k")) == 1
    assert lint.check_store_hazard([st.replace("s40 offen", "0 offen"), "v_mov_b32_e32 v157, v3"], "k") == []      # no SGPR soffset
    assert lint.check_store_hazard([st, "v_mov_b32_e32 v160, v3", "v_mov_b32_e32 v161, v3", "v_mov_b32_e32 v157, v3"], "k") == []


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
