"""N>1 path on CPU: world_size-2 gloo run of the only collective the hot path has (the start-up
weight broadcast) plus the shard / max-over-ranks helpers bench.py uses."""
import os
import socket

import torch
import torch.multiprocessing as mp

from mvd_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(7)
    ref = [torch.randn(1000, generator=g).to(torch.bfloat16), torch.randn(33, 5, generator=g),
           torch.randn(4096, generator=g).to(torch.bfloat16), torch.randn(7, generator=g)]
    mine = [t.clone() if rank == 0 else torch.zeros_like(t) for t in ref]
    ptrs = [t.data_ptr() for t in mine]
    stats = D.broadcast_tensors(mine, src=0, bucket_bytes=1024)       # several buckets per dtype
    ok = all(torch.equal(a, b) for a, b in zip(mine, ref)) and ptrs == [t.data_ptr() for t in mine]
    ok = ok and stats["bytes"] == sum(t.numel() * t.element_size() for t in ref) and stats["buckets"] >= 3
    mx = D.max_over_ranks(float(rank + 1), "cpu")
    ok = ok and mx == float(world)
    D.barrier()
    ret[rank] = ok
    torch.distributed.destroy_process_group()


class _FakeEngine:
    """CPU stand-in with the two attributes/one method of MVDEngine the broadcast path uses."""

    def __init__(self, weights):
        self._weights = weights
        self._arenas = {}

    def consolidate_weights(self):
        if self._arenas:
            return
        flat = {f"{i}/{k}": t for i, d in enumerate(self._weights) for k, t in d.items()}
        self._arenas, views = D.pack_into_arenas(flat)
        for name, v in views.items():
            i, k = name.split("/", 1)
            self._weights[int(i)][k] = v


def _engine_weights(rank):
    g = torch.Generator().manual_seed(11)
    ref = [{"a.w": torch.randn(300, 64, generator=g).to(torch.bfloat16), "a.b": torch.randn(300, generator=g),
            "conv.w": torch.randn(77, 9, generator=g).to(torch.bfloat16)},
           {"enc.w": torch.randn(1000, generator=g).to(torch.bfloat16), "enc.g": torch.randn(5, generator=g)}]
    mine = [{k: (v.clone() if rank == 0 else torch.zeros_like(v)) for k, v in d.items()} for d in ref]
    return ref, mine


def _engine_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env("gloo")
    ref, mine = _engine_weights(rank)
    eng = _FakeEngine(mine)
    stats = D.broadcast_engine_weights(eng, 0, bucket_bytes=4096)        # several slices per arena
    ok = all(torch.equal(eng._weights[i][k], v) for i, d in enumerate(ref) for k, v in d.items())
    # every slot is an aligned VIEW of its dtype's arena: the broadcast touched no staging copy
    for d in eng._weights:
        for t in d.values():
            a = eng._arenas[t.dtype]
            lo = a.data_ptr()
            ok = ok and lo <= t.data_ptr() < lo + a.numel() * a.element_size() and (t.data_ptr() - lo) % D.ARENA_ALIGN == 0
    arena_bytes = sum(a.numel() * a.element_size() for a in eng._arenas.values())
    ok = ok and stats["bytes"] == arena_bytes and stats["buckets"] >= 3 and stats["seconds"] >= 0.0
    D.barrier()
    ret[rank] = ok
    torch.distributed.destroy_process_group()


def test_gloo_engine_weight_broadcast_world2():
    """The engine-weight path of bench.py (``broadcast_engine_weights``) on two CPU-side fake engines."""
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_engine_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _mismatch_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env("gloo")
    _, mine = _engine_weights(rank)
    if rank == 1:
        del mine[1]["enc.w"]          # e.g. this rank de-duplicated its encoder weights and rank 0 did not
    eng = _FakeEngine(mine)
    try:
        D.broadcast_engine_weights(eng, 0, bucket_bytes=4096)
        ret[rank] = "no error"
    except RuntimeError as e:
        ret[rank] = "layout differs" in str(e)
    D.barrier()
    torch.distributed.destroy_process_group()


def test_gloo_weight_broadcast_rejects_mismatched_layouts():
    """Ranks whose packed arenas differ (the rank-0-random / rank-1-zero encoder de-duplication trap of bench.py) get a clear
    error on EVERY rank before any payload moves -- not a transport abort in the middle of the broadcast."""
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_mismatch_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _q2_shard(rank):
    """Reference features of one rank: two 'feature maps' (C 64 / 128) with 6 + 3 pixels, UNEQUAL batches per rank and a
    large common offset (the cancellation case a sum / sum-of-squares exchange would lose)."""
    g = torch.Generator().manual_seed(100 + rank)
    b = 3 + 2 * rank
    return [torch.randn(b, 64, 6, generator=g) * 0.7 + 40.0, torch.randn(b, 128, 3, generator=g) * 2.0 - 5.0]


def _q2_local(feats):
    rows = []
    for f in feats:                                              # [b][C][pixels] -> per pixel (n, mean, M2) over (b, C)
        n = f.shape[0] * f.shape[1]
        x = f.permute(2, 0, 1).reshape(f.shape[2], n)
        mean = x.mean(1)
        rows.append(torch.stack([torch.full_like(mean, float(n)), mean, ((x - mean[:, None]) ** 2).sum(1)], 1))
    return torch.cat(rows)


def _q2_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env("gloo")
    got = D.merge_reference_stats(_q2_local(_q2_shard(rank)))
    # the reference's arithmetic on the UNSHARDED batch (attention.py:95-103: mean / unbiased std over dims (0, 1), clamp, x0.5)
    full = [torch.cat([_q2_shard(r)[i] for r in range(world)]).double() for i in range(2)]
    mean = torch.cat([f.mean(dim=(0, 1)) for f in full])
    k = torch.cat([0.5 / f.std(dim=(0, 1)).clamp_min(1e-6) for f in full])
    ok = torch.allclose(got[:, 0].double(), mean, rtol=1e-6, atol=0) and torch.allclose(got[:, 1].double(), k, rtol=1e-5, atol=0)
    gathered = [torch.empty_like(got) for _ in range(world)]
    torch.distributed.all_gather(gathered, got)
    ok = ok and all(torch.equal(g_, got) for g_ in gathered)           # every rank holds bit-identical statistics
    D.barrier()
    ret[rank] = bool(ok)
    torch.distributed.destroy_process_group()


def test_gloo_global_reference_statistics_world2():
    """SURVEY.md 8e mode (ii): the merged per-pixel (mean, k) of two ranks' shards equal the statistics of the unsharded
    batch, with unequal shard sizes and a large common offset; all ranks end with the same bits."""
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_q2_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_reference_statistics_merge_single_process():
    """Without a process group the merge only finishes the local statistics (n, mean, M2) -> (mean, 0.5 / max(std, 1e-6))."""
    feats = _q2_shard(0)
    got = D.merge_reference_stats(_q2_local(feats))
    mean = torch.cat([f.mean(dim=(0, 1)) for f in feats])
    k = torch.cat([0.5 / f.std(dim=(0, 1)).clamp_min(1e-6) for f in feats])
    assert torch.allclose(got[:, 0], mean, rtol=1e-6) and torch.allclose(got[:, 1], k, rtol=1e-5)
    const = torch.tensor([[8.0, 1.5, 0.0]])                       # zero variance: the clamp, not a division by zero
    assert D.merge_reference_stats(const)[0].tolist() == [1.5, 0.5 / 1e-6]


def test_pack_into_arenas_layout():
    ref, _ = _engine_weights(0)
    flat = {f"{i}/{k}": t for i, d in enumerate(ref) for k, t in d.items()}
    arenas, views = D.pack_into_arenas(flat)
    assert set(arenas) == {torch.bfloat16, torch.float32} and set(views) == set(flat)
    for k, v in views.items():
        assert torch.equal(v, flat[k]) and v.shape == flat[k].shape
        assert (v.data_ptr() - arenas[v.dtype].data_ptr()) % D.ARENA_ALIGN == 0


def test_gloo_weight_broadcast_world2():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_ranges_partition_objects():
    for n, world in ((64, 8), (10, 4), (3, 8), (8, 1)):
        seen = []
        for r in range(world):
            seen += list(D.shard_range(n, r, world))
        assert seen == list(range(n))
        sizes = [len(D.shard_range(n, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_single_process_helpers_are_noops():
    t = [torch.ones(3)]
    assert D.broadcast_tensors(t)["bytes"] == 0
    assert D.max_over_ranks(2.5, "cpu") == 2.5
    D.barrier()
