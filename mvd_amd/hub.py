"""Offline resolution of a ``pretrained_model_name_or_path`` the way the reference's loaders see it.

The reference builds everything with ``from_pretrained(name, cache_dir=...)`` (mvd_unet.py:46-52, 411-415;
image_encoder.py:18-22; infer.py:33-44 passes ``--base-model stabilityai/stable-diffusion-2-1`` and the cache directory it
also exports as ``HF_HOME``).  This build never fetches: a name is either a diffusers snapshot DIRECTORY or a hub repo id
whose snapshot already sits in a huggingface cache -- ``<cache>/models--<org>--<name>/snapshots/<revision>/`` with
``refs/main`` naming the revision.  Anything else is an error (``MvdError``), never a silently random-initialised model.
"""
from __future__ import annotations

import os
from typing import List, Optional

from ._lib import MvdError


def _cache_roots(cache_dir) -> List[str]:
    """Hub cache directories in the order huggingface_hub consults them (explicit cache_dir first)."""
    roots = []
    if cache_dir:
        roots += [str(cache_dir), os.path.join(str(cache_dir), "hub")]     # (infer.py hands the HF_HOME directory itself)
    for env in ("HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE"):
        if os.environ.get(env):
            roots.append(os.environ[env])
    if os.environ.get("HF_HOME"):
        roots.append(os.path.join(os.environ["HF_HOME"], "hub"))
    roots.append(os.path.join(os.path.expanduser("~"), ".cache", "huggingface", "hub"))
    seen, out = set(), []
    for r in roots:
        if r not in seen:
            seen.add(r)
            out.append(r)
    return out


def _snapshot_in(repo_dir: str, revision: Optional[str]) -> Optional[str]:
    snaps = os.path.join(repo_dir, "snapshots")
    if not os.path.isdir(snaps):
        return None
    rev = revision
    if rev is None:
        ref = os.path.join(repo_dir, "refs", "main")
        if os.path.isfile(ref):
            rev = open(ref).read().strip()
    if rev and os.path.isdir(os.path.join(snaps, rev)):
        return os.path.join(snaps, rev)
    if revision is None:          # no refs/main: a cache with exactly one snapshot is unambiguous
        only = [d for d in sorted(os.listdir(snaps)) if os.path.isdir(os.path.join(snaps, d))]
        if len(only) == 1:
            return os.path.join(snaps, only[0])
    return None


def resolve_snapshot(name_or_path, cache_dir=None, revision: Optional[str] = None, required: bool = True) -> Optional[str]:
    """Directory of the diffusers snapshot ``name_or_path`` stands for, or None for ``None`` (checkpoint-free construction:
    tests, synthetic weights).  Raises ``MvdError`` when a name was given and nothing local answers to it."""
    if name_or_path is None:
        return None
    p = str(name_or_path)
    if os.path.isdir(p):
        return p
    tried = []
    if p.count("/") <= 1 and not p.startswith((".", "/", "~")):
        repo = "models--" + p.replace("/", "--")
        for root in _cache_roots(cache_dir):
            d = os.path.join(root, repo)
            tried.append(d)
            snap = _snapshot_in(d, revision) if os.path.isdir(d) else None
            if snap:
                return snap
    if not required:
        return None
    raise MvdError(
        f"pretrained model {p!r}: not a directory and no cached snapshot found (looked for {', '.join(tried) or 'a local path'}). "
        "This build has no network access and never random-initialises a named model: pass a snapshot directory, populate the "
        "huggingface cache (cache_dir / HF_HOME), or pass None with unet_config= for checkpoint-free construction.")
