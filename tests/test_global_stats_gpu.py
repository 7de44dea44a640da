"""SURVEY.md 8e mode (ii): global Q2 statistics.  The adapter normalises the reference features over (batch, channel) per
pixel (attention.py:95-103), so data-parallel shards see per-shard statistics unless they exchange them.  With the exchange
(``mvd_engine_reference_encode`` -> merge -> ``mvd_engine_reference_finish``) the sharded forward must reproduce the
UNSHARDED batch -- checked against the CPU oracle run on the whole batch and against the engine's own whole-batch forward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _shard_forward(model, inp, rows, t, local_parts=None, merged=None):
    """One 'rank': engine-level calls on the rows of its shard.  First call (merged is None) returns the local statistics,
    second call finishes the reference with the merged ones and runs the main pass."""
    eng = model._sync_engine()
    lat = inp["lat"][rows].cuda().contiguous()
    text = inp["text"][rows].cuda().contiguous()
    if merged is None:
        return eng.reference_encode(lat, text, len(rows))
    eng.reference_finish(merged)
    ts = torch.full((len(rows),), float(t), device="cuda")
    return eng.forward(inp["sample"][rows].cuda().contiguous(), ts, text, reuse_ref=True, keep_features=True)


def test_sharded_forward_with_global_statistics_equals_unsharded_batch():
    from mvd_amd import distributed as D
    from oracle import mvd as OM
    from tests.parity_util import build_pair, make_inputs, rel_l2
    B, hw, Lt, t = 6, 16, 7, 300
    ocfg, params, full = build_pair("tiny", 0, 96, 48)
    inp = make_inputs(ocfg, B, hw, Lt, 3, 96)
    inp["lat"] = inp["lat"] + torch.linspace(-0.5, 0.5, B)[:, None, None, None]      # shards with visibly different statistics
    want = OM.multiview_unet_forward(params, ocfg, inp["sample"], torch.tensor(t), inp["text"], None, None, inp["lat"],
                                     img_ref_scale=0.3, cam_modulation_strength=0.2)
    shards = [list(range(0, 2)), list(range(2, 6))]                                  # unequal shard sizes
    ranks = [build_pair("tiny", 0, 96, 48)[2] for _ in shards]
    with torch.no_grad():
        whole = full(inp["sample"].cuda(), torch.tensor(t), inp["text"].cuda(), source_image_latents=inp["lat"].cuda()).sample
        local = [m(inp["sample"][r].cuda(), torch.tensor(t), inp["text"][r].cuda(), source_image_latents=inp["lat"][r].cuda()).sample
                 for m, r in zip(ranks, shards)]
        parts = [_shard_forward(m, inp, r, t) for m, r in zip(ranks, shards)]
        assert parts[0].shape == parts[1].shape and parts[0].shape[1] == 3
        merged = D.merge_stat_parts(parts)
        glob = [_shard_forward(m, inp, r, t, merged=merged) for m, r in zip(ranks, shards)]
    torch.cuda.synchronize()
    glob, local = torch.cat(glob), torch.cat(local)
    e_glob, e_loc = rel_l2(glob, want), rel_l2(local, want)
    e_whole, e_gw = rel_l2(whole, want), rel_l2(glob, whole.cpu())
    print(f"vs the oracle on the whole batch: global-stats shards {e_glob:.4f}  local-stats shards {e_loc:.4f}  "
          f"engine on the whole batch {e_whole:.4f};  global-stats shards vs engine on the whole batch {e_gw:.4f}")
    assert torch.isfinite(glob).all()
    # the fp32 oracle is the noise-free yardstick (two bf16 engine runs at different batch sizes differ by about as much as
    # either differs from it): with the exchange the shards are as close to the unsharded reference as the unsharded engine
    # run is -- the stated end-to-end tolerance -- and per-shard statistics are measurably further away
    assert e_glob < 2e-2 and e_glob < 1.25 * e_whole + 2e-3
    assert e_loc > 1.2 * e_glob       # the test bites: replica-local statistics give a different (the DDP) result
    assert e_gw < 2e-2


def test_one_rank_global_mode_matches_the_fused_reference_pass():
    """reference_stats_group = True without a process group: encode -> (local) merge -> finish -> main pass is the ordinary
    forward, bit for bit, and the Q5 cache keeps working."""
    from tests.parity_util import build_pair, make_inputs
    ocfg, params, model = build_pair("tiny", 0, 96, 48)
    inp = make_inputs(ocfg, 3, 16, 7, 5, 96)
    model.fourier_projection = inp["proj"]
    args = (inp["sample"].cuda(), torch.tensor(500), inp["text"].cuda())
    kw = dict(source_camera=inp["src"].cuda(), target_camera=inp["tgt"].cuda(), source_image_latents=inp["lat"].cuda())
    with torch.no_grad():
        want = model(*args, **kw).sample
        model.reference_stats_group = True
        got = model(*args, **kw).sample
        model.cache_reference = True
        again = model(*args, **kw).sample          # same input tensors: both of these reuse the cached reference
        cached = model(*args, **kw).sample
    torch.cuda.synchronize()
    assert torch.equal(got, want)     # one part: the merge reproduces refnorm_kernel's arithmetic bit for bit
    assert torch.equal(again, got) and torch.equal(cached, got)


def test_reference_halves_reject_misuse():
    from mvd_amd import _lib as L
    from tests.parity_util import build_pair, make_inputs
    ocfg, params, model = build_pair("tiny", 0, 96, 48)
    inp = make_inputs(ocfg, 2, 16, 7, 1, 96)
    eng = model._sync_engine()
    with pytest.raises(L.MvdError):
        eng.reference_finish(torch.zeros(4, 2, device="cuda"))                       # nothing pending
    stats = eng.reference_encode(inp["lat"].cuda(), inp["text"].cuda(), 2)
    assert stats.shape[1] == 3 and bool((stats[:, 0] > 0).all()) and bool((stats[:, 2] >= 0).all())
    with pytest.raises(L.MvdError):
        eng.reference_finish(torch.zeros(stats.shape[0] + 1, 2, device="cuda"))     # wrong pixel count
    with pytest.raises(L.MvdError):                                                  # main pass before the second half
        eng.forward(inp["sample"].cuda(), torch.zeros(2, device="cuda"), inp["text"].cuda(), reuse_ref=True, keep_features=True)
