"""Multi-GPU support for the hot path: one process per GPU, pairs sharded by object, and ONE
collective -- the start-up broadcast of the packed weights from rank 0 (RCCL over xGMI when the
backend is "nccl").  There is no per-forward collective: the denoising path has no cross-pair
data flow (replica-local Q2 statistics)."""
from __future__ import annotations

import os
import socket
import time
from typing import Dict, Iterable, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl", set_device: bool = True) -> tuple:
    """(rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1.  With the RCCL backend the rank's
    GPU is selected BEFORE the process group exists, so the communicator is created on the right device."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if set_device and backend == "nccl" and torch.cuda.is_available():
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl" and torch.cuda.is_available():
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


ARENA_ALIGN = 256      # bytes; every slot of an arena starts on this boundary (the engine needs >= 16)


def pack_into_arenas(tensors: Dict[str, torch.Tensor]) -> tuple:
    """Move ``tensors`` (name -> contiguous tensor, one device) into ONE flat buffer per dtype.

    Returns ``(arenas, views)``: ``arenas`` maps dtype -> flat tensor, ``views`` maps name -> a view of its arena with
    the original shape (256-byte aligned).  A broadcast of the packed weights is then a handful of large in-place
    ``dist.broadcast`` calls on slices of the arenas -- no ``torch.cat`` staging copy and no copy-back."""
    arenas: Dict[torch.dtype, torch.Tensor] = {}
    views: Dict[str, torch.Tensor] = {}
    by_dtype: Dict[torch.dtype, List[str]] = {}
    for name in sorted(tensors):
        by_dtype.setdefault(tensors[name].dtype, []).append(name)
    for dtype, names in by_dtype.items():
        esz = tensors[names[0]].element_size()
        step = ARENA_ALIGN // esz
        offs, total = [], 0
        for n in names:
            offs.append(total)
            total += (tensors[n].numel() + step - 1) // step * step
        arena = torch.zeros(total, dtype=dtype, device=tensors[names[0]].device)
        for n, off in zip(names, offs):
            t = tensors[n]
            v = arena[off:off + t.numel()].view(t.shape)
            v.copy_(t)
            views[n] = v
        arenas[dtype] = arena
    return arenas, views


def broadcast_arenas(arenas: Iterable[torch.Tensor], src: int = 0, bucket_bytes: int = 256 << 20) -> Dict[str, float]:
    """In-place broadcast of flat buffers in slices of ``bucket_bytes`` (a ring broadcast over xGMI is per-link bound,
    so few large messages; slices rather than one call keep the transfers pipelined with each other's completion)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(bytes=0, seconds=0.0, buckets=0)
    arenas = list(arenas)
    # Every rank must hold the same layout: a mismatch (e.g. one rank de-duplicated the encoder weights and another did not)
    # would otherwise surface as a transport abort or a hang in the middle of the data broadcast.  One tiny all-gather first.
    dev = arenas[0].device if arenas else torch.device("cpu")
    sig = torch.zeros(16, dtype=torch.int64, device=dev)
    sig[0] = len(arenas)
    for i, a in enumerate(arenas[:7]):
        sig[1 + 2 * i], sig[2 + 2 * i] = a.numel(), a.element_size()
    sigs = [torch.zeros_like(sig) for _ in range(dist.get_world_size())]
    dist.all_gather(sigs, sig)
    ref = sigs[src].tolist()
    for r, s_ in enumerate(sigs):
        if s_.tolist() != ref:
            raise RuntimeError(f"weight arena layout differs between rank {src} {ref[:1 + 2 * ref[0]]} and rank {r} "
                               f"{s_.tolist()[:1 + 2 * int(s_[0])]}: every rank must pack the same slots "
                               f"(same config, adapter / camera / encoder-sharing choices) before the broadcast")
    t0 = time.perf_counter()
    total = nb = 0
    cuda = False
    for a in arenas:
        cuda = cuda or a.is_cuda
        step = max(1, bucket_bytes // a.element_size())
        for off in range(0, a.numel(), step):
            chunk = a[off:off + step]
            dist.broadcast(chunk, src)
            total += chunk.numel() * chunk.element_size()
            nb += 1
    if cuda:
        torch.cuda.synchronize()
    return dict(bytes=total, seconds=time.perf_counter() - t0, buckets=nb)


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous shard of ``n_items`` objects for ``rank`` (keeps an object's views on one GPU)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def broadcast_tensors(tensors: Sequence[torch.Tensor], src: int = 0, bucket_bytes: int = 256 << 20) -> Dict[str, float]:
    """In-place broadcast of ``tensors`` from ``src``.  Tensors are coalesced into flat buckets per
    dtype (few large transfers: a ring broadcast over xGMI is per-link bound, so large messages)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(bytes=0, seconds=0.0, buckets=0)
    t0 = time.perf_counter()
    total = 0
    nb = 0
    by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, ts in by_dtype.items():
        bucket: List[torch.Tensor] = []
        size = 0

        def flush():
            nonlocal bucket, size, total, nb
            if not bucket:
                return
            flat = torch.cat([b.reshape(-1) for b in bucket])
            dist.broadcast(flat, src)
            off = 0
            for b in bucket:
                n = b.numel()
                b.copy_(flat[off:off + n].view_as(b))
                off += n
            total += flat.numel() * flat.element_size()
            nb += 1
            bucket, size = [], 0

        for t in ts:
            bucket.append(t)
            size += t.numel() * t.element_size()
            if size >= bucket_bytes:
                flush()
        flush()
    if tensors and tensors[0].is_cuda:
        torch.cuda.synchronize()
    return dict(bytes=total, seconds=time.perf_counter() - t0, buckets=nb)


def broadcast_engine_weights(engine, src: int = 0, bucket_bytes: int = 256 << 20) -> Dict[str, float]:
    """Broadcast every packed device weight of ``engine`` (both weight sets) from ``src`` in place.  The engine's
    weights are first consolidated into one arena per dtype (``engine.consolidate_weights()``; a no-op when already
    done), so each bucket is a slice of an arena: nothing is staged or copied back."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(bytes=0, seconds=0.0, buckets=0)
    engine.consolidate_weights()
    return broadcast_arenas(engine._arenas.values(), src, bucket_bytes)


def merge_reference_stats(local: torch.Tensor, group=None) -> torch.Tensor:
    """Global Q2 statistics (SURVEY.md 8e mode ii): merge the ranks' per-pixel ``(n, mean, M2)`` rows
    (``Engine.reference_encode``) into ``(mean, k)`` with ``k = 0.5 / max(std, 1e-6)`` and the UNBIASED std over all ranks'
    ``n`` samples (attention.py:95-103 on the unsharded batch).  ONE all-gather of ``pixels * 12`` bytes (323 KB at 64x64);
    the merge is Chan's parallel update in fp64, in rank order, so every rank computes bit-identical results.  Without a
    process group (or world size 1) it just finishes the local statistics."""
    if local.dim() != 2 or local.shape[1] != 3:
        raise ValueError("merge_reference_stats expects [pixels][3] rows (n, mean, M2)")
    parts = [local]
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        on_host = local.is_cuda and dist.get_backend(group) == "gloo"       # gloo: stage the 323 KB through the host
        src = local.cpu() if on_host else local.contiguous()
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src, group=group)
        parts = [p.to(local.device) for p in parts]
    return merge_stat_parts(parts)


def merge_stat_parts(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """The arithmetic of ``merge_reference_stats`` on the already gathered per-rank ``[pixels][3]`` rows, in list order."""
    n = torch.zeros_like(parts[0][:, 0], dtype=torch.float64)
    mean = torch.zeros_like(n)
    m2 = torch.zeros_like(n)
    for p in parts:
        pn, pm, pq = p[:, 0].double(), p[:, 1].double(), p[:, 2].double()
        tot = n + pn
        delta = pm - mean
        mean = mean + delta * pn / tot
        m2 = m2 + pq + delta * delta * n * pn / tot
        n = tot
    # the last step in fp32, as refnorm_kernel does it (var = M2 / (n - 1); k = 0.5 / max(sqrt(var), 1e-6)): with ONE part
    # the result is then bit-identical to the fused reference pass
    # (n = batch x channels per pixel, >= 64 for every feature map of the engine (channels % 64 == 0), so n - 1 > 0; a
    #  single-sample pixel would give NaN exactly like torch.std(unbiased) in attention.py:99 -- not checked here because
    #  reading n back would put a host sync into every cold forward)
    mean, m2, n = mean.float(), m2.float(), n.float()
    k = 0.5 / (m2 / (n - 1.0)).sqrt().clamp_min(1e-6)
    return torch.stack([mean, k], dim=1).contiguous()


def any_rank(flag: bool, group=None, device=None) -> bool:
    """True on every rank of ``group`` iff ``flag`` is true on at least one of them (one 8-byte all-reduce; no process
    group or world size 1: ``flag`` itself).  Used to make "re-run the reference pass?" a collective decision."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(flag)
    on_host = dist.get_backend(group) == "gloo"
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device="cpu" if on_host or device is None else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item())


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown():
    """Leave the process group in step with the other ranks (barrier, then destroy); never raises -- by the time this runs the
    results are out, and a rank that cannot say goodbye must not turn a finished run into a failed one."""
    if not dist.is_initialized():
        return
    try:
        if dist.get_world_size() > 1:
            dist.barrier()
        dist.destroy_process_group()
    except Exception as ex:          # pragma: no cover
        print(f"warning: process group shutdown: {ex}", flush=True, file=__import__("sys").stderr)


def collective_library() -> str:
    """Name + version of the collective library behind the process group (what carried the weight broadcast): "rccl 2.x.y"
    for backend nccl on ROCm (torch.cuda.nccl.version() IS RCCL's there), "gloo" for the CPU rehearsal backend."""
    if not dist.is_initialized():
        return "none"
    be = dist.get_backend()
    if be != "nccl":
        return str(be)
    try:
        v = torch.cuda.nccl.version()
        v = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:           # pragma: no cover  (a build without the binding)
        v = "?"
    return ("rccl " if getattr(torch.version, "hip", None) else "nccl ") + v


def device_identity(local_rank: int) -> Dict[str, object]:
    """What distinguishes this rank's GPU from the others' on one node: LOCAL_RANK, the device index it bound, and the
    device's UUID / PCI bus id where the runtime exposes them (a CPU-only rehearsal reports the process id instead)."""
    ident: Dict[str, object] = {"local_rank": int(local_rank), "pid": os.getpid(), "host": socket.gethostname()}
    if torch.cuda.is_available():
        idx = torch.cuda.current_device()
        ident["device"] = idx
        try:
            p = torch.cuda.get_device_properties(idx)
            ident["name"] = p.name
            for key in ("uuid", "pci_bus_id", "pci_device_id"):
                if hasattr(p, key):
                    ident[key] = str(getattr(p, key))
        except Exception:       # pragma: no cover
            pass
    return ident


def gather_identities(local_rank: int) -> List[Dict[str, object]]:
    """Every rank's ``device_identity`` (one all_gather_object; world size 1: a one-element list).  Rank 0 prints it as
    ``ranks_seen`` so that whoever launched N ranks can check that N DISTINCT GPUs took part."""
    me = device_identity(local_rank)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [me]
    out: List[object] = [None] * dist.get_world_size()
    dist.all_gather_object(out, me)
    return out       # type: ignore[return-value]


def _usable_uuid(u) -> bool:
    """A UUID the runtime really reported: not missing, not a placeholder, not all zeros."""
    s = str(u if u is not None else "").strip().lower()
    if s in ("", "none", "?", "n/a"):
        return False
    body = s[4:] if s.startswith("gpu-") else s
    return any(c not in "0-: " for c in body)


def distinct_devices(idents: Sequence[Dict[str, object]]) -> int:
    """Number of different GPUs among the gathered identities.  A GPU is (host, UUID) when the runtime reports usable UUIDs
    that are not all the same -- two ranks that reach one physical GPU through different visibility masks / device indices
    then count ONCE, and equal tuples on different hosts count twice.  Where UUIDs are absent or degenerate (every device the
    same one) the key falls back to (host, PCI bus id, PCI device id, bound device index), so a runtime that hands every
    device the same UUID cannot make eight GPUs count as one.  Without a GPU (CPU rehearsal) host + process id stand in."""
    gpu = [d for d in idents if "device" in d]
    uuids = [str(d.get("uuid")) for d in gpu]
    by_uuid = bool(gpu) and all(_usable_uuid(u) for u in uuids) and (len(gpu) == 1 or len(set(uuids)) > 1)
    keys = []
    for d in idents:
        if "device" not in d:
            keys.append(("pid", d.get("host"), d.get("pid")))
        elif by_uuid:
            keys.append(("gpu", d.get("host"), str(d.get("uuid"))))
        else:
            keys.append(("gpu", d.get("host"), d.get("pci_bus_id"), d.get("pci_device_id"), d.get("device")))
    return len(set(keys))


def check_enough_devices(world: int, rehearsal: bool = False):
    """One rank per GPU: refuse to start more ranks than the node has devices (unless rehearsing on one GPU)."""
    if rehearsal or not torch.cuda.is_available():
        return
    n = torch.cuda.device_count()
    if n < world:
        raise RuntimeError(f"WORLD_SIZE={world} ranks but only {n} GPU(s) visible: one process per GPU is the contract "
                           "(set MVD_BENCH_REHEARSAL=1 for a one-GPU rehearsal of the control flow over gloo)")
