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
