"""packing.pack_ws: the fragment-ordered weight stream of the weight-streaming convolution (mvd_amd/csrc/conv_ws.hip).  The
layout is checked element by element against its definition -- [column tile][round][wave][tap][lane][8] for the convolution,
then [round][wave][lane][8] for the fused 1x1 shortcut, lane (i, h) = (lane & 15, lane >> 4) holding W[16 ct + i][128 rd + 32 wave
+ 8 h + j] of tap 3 ky + kx -- and pack_unet registers the twins exactly where the engine looks for them."""
import torch

from mvd_amd.packing import pack_ws


def test_pack_ws_layout_by_definition():
    g = torch.Generator().manual_seed(0)
    n, c, sc = 48, 256, 384
    w4 = torch.randn(n, c, 3, 3, generator=g).to(torch.bfloat16)
    wsc = torch.randn(n, sc, generator=g).to(torch.bfloat16)
    p = pack_ws(w4, wsc)
    rc, rs = c // 128, sc // 128
    tile = (rc * 9 + rs) * 4 * 512
    assert p.dtype == torch.bfloat16 and p.numel() == (n // 16) * tile == n * (9 * c + sc)
    p = p.reshape(n // 16, tile)
    for ct in range(n // 16):
        conv = p[ct, :rc * 4 * 9 * 512].reshape(rc, 4, 9, 64, 8)
        short = p[ct, rc * 4 * 9 * 512:].reshape(rs, 4, 64, 8)
        for lane in (0, 7, 15, 16, 33, 63):
            i, h = lane & 15, lane >> 4
            for rd in range(rc):
                for wave in range(4):
                    k0 = 128 * rd + 32 * wave + 8 * h
                    for tap in range(9):
                        assert torch.equal(conv[rd, wave, tap, lane], w4[16 * ct + i, k0:k0 + 8, tap // 3, tap % 3])
            for rd in range(rs):
                for wave in range(4):
                    k0 = 128 * rd + 32 * wave + 8 * h
                    assert torch.equal(short[rd, wave, lane], wsc[16 * ct + i, k0:k0 + 8])


def test_pack_unet_registers_ws_twins_for_every_level_but_the_first():
    from mvd_amd.config import UNetConfig
    from mvd_amd.packing import pack_unet
    from oracle import sd21_unet as OU                     # (test infrastructure: seeded weights of the tiny topology)
    cfg = UNetConfig.tiny()
    sd = OU.init_params(OU.UNetConfig.tiny(), seed=3)
    packed = pack_unet(sd, cfg, "cpu", adapter=False)
    first = cfg.block_out_channels[0]
    seen = 0
    split = cfg.resnet_input_split()
    for key, cin, cout in cfg.resnets():
        has1, has2 = f"{key}.conv1.ws" in packed, f"{key}.conv2.ws" in packed
        c0, c1 = split[key]
        assert c0 + c1 == cin and (c1 == 0) == (not key.startswith("up_blocks.")), key
        assert has1 == (cout > first and cout % 128 == 0 and cin % 128 == 0), key
        assert has2 == (cout > first and cout % 128 == 0 and (cin == cout or (c0 % 128 == 0 and c1 % 128 == 0))), key
        if has1:
            assert packed[f"{key}.conv1.ws"].numel() == cout * 9 * cin
        if has2:
            assert packed[f"{key}.conv2.ws"].numel() == cout * (9 * cout + (cin if cin != cout else 0))
        seen += has1 + has2
    assert seen > 0


def test_pack_unet_without_small_batch_twins_keeps_only_what_many_image_batches_read():
    """the copies only a batch-1 forward reads are optional.  Off: no ``.ws`` twin anywhere, LayerNorm-folded
    twins only up to C = 640 (what mvd_gemm_ln_fold_ok takes), every other slot bit-identical."""
    import torch
    from mvd_amd.config import UNetConfig
    from mvd_amd.packing import LN_FOLD_LARGE_BATCH_MAX_C, pack_unet
    from oracle import sd21_unet as OU
    cfg = UNetConfig.tiny()
    sd = OU.init_params(OU.UNetConfig.tiny(), seed=3)
    full = pack_unet(sd, cfg, "cpu", adapter=False)
    lean = pack_unet(sd, cfg, "cpu", adapter=False, small_batch_twins=False)
    assert set(lean) <= set(full)
    dropped = set(full) - set(lean)
    assert dropped and all(k.endswith((".ws", ".wf", ".cf")) for k in dropped), sorted(dropped)[:5]
    assert not any(k.endswith(".ws") for k in lean)
    for key, _feat, C, _heads in cfg.transformers():
        for slot in (f"{key}.attn1.qkv", f"{key}.attn2.q", f"{key}.ff1"):
            assert (f"{slot}.wf" in lean) == (C <= LN_FOLD_LARGE_BATCH_MAX_C), slot
            assert f"{slot}.wf" in full
    for k, t in lean.items():
        assert torch.equal(t, full[k]), k
    nbytes = lambda d: sum(t.numel() * t.element_size() for t in d.values())   # noqa: E731
    assert nbytes(lean) < nbytes(full)


def test_every_packed_ws_twin_is_reachable_from_the_engine():
    """a ``.ws`` twin is packed only where conv_ws.hip's shape predicate (``mvd_conv_ws_applicable``: C % 128, N % 16,
    EACH shortcut source % 128) can say yes for the resnet's real operands -- walked over the SD-2.1 configuration and over one
    whose skip halves are not multiples of 128 (where the concatenated width still is), with the engine's work-item cap
    (``try_ws``: (M / 64) * (N / 16) <= 1000) at the map sizes of a batch-1 forward on 64 x 64 latents."""
    from mvd_amd.config import UNetConfig
    from mvd_amd.packing import ws_twin_shapes_ok
    for cfg in (UNetConfig.sd21(), UNetConfig(block_out_channels=(192, 320, 448, 448), num_heads=(3, 5, 7, 7))):
        first = cfg.block_out_channels[0]
        split = cfg.resnet_input_split()
        level_hw = {c: 64 >> i for i, c in enumerate(cfg.block_out_channels)}
        n_twins = 0
        for key, cin, cout in cfg.resnets():
            c0, c1 = split[key]
            want1 = cout > first and ws_twin_shapes_ok(cin, cout, 0, 0)
            want2 = cout > first and (ws_twin_shapes_ok(cout, cout, 0, 0) if cin == cout else ws_twin_shapes_ok(cout, cout, c0, c1))
            if cin != cout and cin % 128 == 0 and (c0 % 128 or c1 % 128):
                assert not want2, key                      # the case the advisor named: 640 + 320-style splits
            n_twins += want1 + want2
            if want2 and cin != cout:
                assert c0 % 128 == 0 and c1 % 128 == 0, key
        assert n_twins >= 8, n_twins
    # SD-2.1 at one 64 x 64 latent: every twin the packer emits is taken at some level by the engine's work-item cap
    cfg = UNetConfig.sd21()
    for key, cin, cout in cfg.resnets():
        if cout > cfg.block_out_channels[0]:
            side = 64 >> max(i for i, c in enumerate(cfg.block_out_channels) if c == cout)      # the smallest map of that width
            assert (side * side // 64) * (cout // 16) <= 1000, key
