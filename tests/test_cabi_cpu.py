"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mvd_hip.h
declares, validates arguments on the host, and the sizing dry-run works without a GPU."""
import ctypes as C
import os
import re

import pytest

from mvd_amd import _lib as L
from mvd_amd.config import UNetConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.lib()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mvd_hip.h")).read()
    declared = set(re.findall(r"\b(mvd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in mvd_hip.h but not exported"
    assert declared == set(L.EXPORTED_SYMBOLS)


def _mk_engine(lib, cfg: UNetConfig):
    c = L.mvd_config_t()
    c.in_channels, c.out_channels, c.num_levels = cfg.in_channels, cfg.out_channels, cfg.num_levels
    for i in range(cfg.num_levels):
        c.block_out_channels[i] = cfg.block_out_channels[i]
        c.num_heads[i] = cfg.num_heads[i]
    c.layers_per_block, c.cross_attention_dim = cfg.layers_per_block, cfg.cross_attention_dim
    c.norm_num_groups, c.norm_eps = cfg.norm_num_groups, cfg.norm_eps
    c.cam_output_dim, c.cam_hidden_dim, c.simple_cam_encoder, c.cam_modulation_strength = 1024, 512, 0, 0.2
    h = C.c_void_p()
    rc = lib.mvd_engine_create(C.byref(c), C.byref(h))
    return rc, h


def test_engine_create_and_sizing_dry_run(lib):
    rc, h = _mk_engine(lib, UNetConfig.sd21())
    assert rc == 0, L.last_error()
    assert lib.mvd_engine_num_features(h) == 16
    ws1 = lib.mvd_engine_workspace_bytes(h, 1, 64, 64, 77, 1)
    ws32 = lib.mvd_engine_workspace_bytes(h, 32, 64, 64, 77, 32)
    assert 0 < ws1 < ws32 < 64 * 2 ** 30, (ws1, ws32)
    # cached K_ref/V_ref: SURVEY 8d quotes ~92 MB per pair
    rc1 = lib.mvd_engine_refcache_bytes(h, 1, 64, 64, 0)
    assert 90e6 < rc1 < 95e6, rc1
    assert lib.mvd_engine_workspace_bytes(h, 1, 60, 64, 77, 0) < 0      # not divisible by 8
    assert "divisible" in L.last_error()
    lib.mvd_engine_destroy(h)


def test_engine_rejects_bad_config(lib):
    bad = UNetConfig(block_out_channels=(96, 128, 128, 128), num_heads=(1, 2, 2, 2))
    rc, _ = _mk_engine(lib, bad)
    assert rc != 0 and "level 0" in L.last_error()


def test_forward_without_workspace_or_weights_fails_loudly(lib):
    rc, h = _mk_engine(lib, UNetConfig.tiny())
    assert rc == 0
    a = L.mvd_forward_args_t()
    a.batch, a.height, a.width, a.text_len = 1, 16, 16, 7
    assert lib.mvd_unet_forward(h, C.byref(a), None) != 0
    assert L.last_error()
    lib.mvd_engine_destroy(h)
