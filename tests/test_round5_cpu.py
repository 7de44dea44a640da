"""Host-side behaviour added in round 5 (CPU only): the resnet input split the packer derives the weight-streaming twins from, the
kernel-trace analysis behind DESIGN.md 4.2 / 4.5 (what the other stream runs during attention, what the main queue runs alone), the
Winograd numerics pricing, and the lean packing switch of the mirror."""
import csv
import importlib.util
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_resnet_input_split_matches_the_skip_bookkeeping():
    """UNetConfig.resnet_input_split: up-block resnets read cat([hidden, skip]) -- hidden first, the skip popped from the end of the
    12-tensor list -- every other resnet has one source; the halves add up to the resnet's input width."""
    from mvd_amd.config import UNetConfig
    cfg = UNetConfig.sd21()
    split = cfg.resnet_input_split()
    res = {k: (ci, co) for k, ci, co in cfg.resnets()}
    assert set(split) == set(res)
    for k, (c0, c1) in split.items():
        assert c0 + c1 == res[k][0] and c0 > 0 and (c1 > 0) == k.startswith("up_blocks."), k
    # the SD-2.1 up path, hidden + skip (layer table of SURVEY 8a)
    want = {"up_blocks.0.resnets.0": (1280, 1280), "up_blocks.0.resnets.2": (1280, 1280), "up_blocks.1.resnets.2": (1280, 640),
            "up_blocks.2.resnets.0": (1280, 640), "up_blocks.2.resnets.1": (640, 640), "up_blocks.2.resnets.2": (640, 320),
            "up_blocks.3.resnets.0": (640, 320), "up_blocks.3.resnets.2": (320, 320)}
    for k, v in want.items():
        assert split[k] == v, (k, split[k])


def test_overlap_from_trace_on_a_synthetic_two_queue_trace(tmp_path):
    """tools/overlap_from_trace.py: attention launches on queue 1; queue 2 runs an attention launch over the first one's second half
    and a GroupNorm over the second one; a GEMM on queue 1 runs alone."""
    d = tmp_path / "trace"
    d.mkdir()
    with open(d / "x_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Queue_Id", "Grid_Size_X", "Start_Timestamp", "End_Timestamp"])
        w.writerow(["attn_kernel<4, 2, true>(A)", "1", "1000", 0, 100])
        w.writerow(["attn_kernel<4, 2, true>(A)", "2", "1000", 50, 100])
        w.writerow(["attn_kernel<4, 2, true>(A)", "1", "1000", 100, 200])
        w.writerow(["gn_slice_kernel<4>(B)", "2", "64", 100, 200])
        w.writerow(["gemm_pp_kernel<256>(C)", "1", "65536", 200, 400])
        for i in range(8):
            w.writerow(["attn_kernel<4, 2, true>(A)", "1", "1000", 400 + 10 * i, 410 + 10 * i])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "overlap_from_trace.py"), str(d)], capture_output=True, text=True, check=True).stdout
    assert "13 dispatches" in out or "dispatches on queues" in out
    share = [ln for ln in out.splitlines() if ln.startswith("# share of the attention kernel's lifetime")][0]
    # queue-1 attention lifetime 280 ns: 50 beside attention, 100 beside GroupNorm (HBM-bound), 130 alone; the queue-2 attention
    # launch itself (50 ns) runs entirely beside queue 1's attention
    assert "attention 30.3 %" in share and "hbm-bound 30.3 %" in share and "alone 39.4 %" in share, share
    alone = [ln for ln in out.splitlines() if "gemm_pp_kernel<256>" in ln][0]
    assert "100.0 % of its time" in alone


def test_winograd_numerics_pricing_runs_and_says_what_design_says():
    """tools/probe_winograd_numerics.py: F(2x2, 3x3) is exact in fp32 (the algebra), costs about 2x the direct
    convolution's error with bf16 transformed operands and less than it with fp16 ones."""
    wn = _tool("probe_winograd_numerics")
    torch.manual_seed(1)
    x = torch.nn.functional.silu(torch.randn(1, 64, 8, 8)).to(torch.bfloat16).float()
    w = torch.randn(32, 64, 3, 3) / 24.0
    ref = torch.nn.functional.conv2d(x, w, padding=1)
    assert wn.rel(wn.winograd(x, w, torch.float32), ref) < 5e-6
    e_bf, e_h = wn.rel(wn.winograd(x, w, torch.bfloat16), ref), wn.rel(wn.winograd(x, w, torch.float16), ref)
    e_direct = wn.rel(torch.nn.functional.conv2d(x, w.to(torch.bfloat16).float(), padding=1), ref)
    assert e_h < e_direct < e_bf < 3.0 * e_direct, (e_h, e_direct, e_bf)


def test_mirror_passes_the_lean_packing_switch_to_the_engine(monkeypatch):
    """MultiViewUNet(small_batch_twins=...) reaches MVDEngine (bench.py packs many-image shards lean: 3.96 GB instead of 6.54 GB)."""
    from mvd_amd import engine as E
    from mvd_amd.config import UNetConfig
    from mvd_amd.mvd_unet import MultiViewUNet
    seen = {}

    class FakeEngine:
        def __init__(self, *a, **kw):
            seen.update(kw)
            raise RuntimeError("stop here")

    monkeypatch.setattr(E, "MVDEngine", FakeEngine)
    m = MultiViewUNet(None, unet_config=UNetConfig.tiny(), init="empty", cam_output_dim=96, cam_hidden_dim=48, small_batch_twins=False)
    assert m.small_batch_twins is False
    monkeypatch.setattr(m, "_exec_device", lambda: torch.device("cpu"))
    try:
        m._sync_engine()
    except RuntimeError as ex:
        assert "stop here" in str(ex)
    assert seen.get("small_batch_twins") is False
    sys.path.insert(0, ROOT)
    import bench
    assert bench.LEAN_PACKING_FROM_PAIRS == 16 and bench.WORKLOADS["cfg4"][0] >= bench.LEAN_PACKING_FROM_PAIRS > bench.WORKLOADS["cfg3"][0]
