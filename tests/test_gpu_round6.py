"""Round 6, on the MI355X: the tile conv v3 on fp16 operands (csrc/conv_tile3.hpp Tile3Geo<2, WIDE_F16>; BASELINE.json configs[4]):
both staging sources, both destinations, fp32- and fp16-stored caches, the routing entry points; write-through epilogue stores are
covered by every conv test of the earlier rounds (the results are the same bytes)."""
import pytest
import torch
import torch.nn.functional as F

from tests import util  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from sige_amd import hip as h

    h.lib()
    return h


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _masks(res):
    from sige_amd.utils import reduce_mask

    mask = torch.zeros(res, res, dtype=torch.bool)
    mask[8:44, 12:60] = True
    mask[0, 0] = mask[res - 1, res - 1] = True
    return reduce_mask(mask.to(DEV), 6, 4, 1), reduce_mask(mask.to(DEV), 4, 4, 0)


@pytest.mark.parametrize("c1,c2,cout,up", [(128, 0, 128, False), (64, 64, 64, False), (64, 0, 128, True)])
def test_tile_conv3_f16_gather_forms(hip, c1, c2, cout, up):
    """Source 1 on fp16 operands: raw, affine and affine + SiLU staging, to tiles and into a full tensor with residual, out-affine
    and twins.  Criterion 1 (exact products): an fp64 conv of the fp16-ROUNDED tiles and weights -- what the kernel computes up to
    fp32 summation order (swish_fast in the staging path is <= 1e-6 relative: a value next to an fp16 rounding boundary may round
    the other way, so a small fraction of outputs may sit one fp16 step of one operand off).  Criterion 2: the conv_mfma.hpp fp16
    launch of the same call (ConvGeoH: the same roundings)."""
    torch.manual_seed(c1 + c2 + cout + 1)
    B, res = 1, 64
    C = c1 + c2
    src = res // 2 if up else res
    idx, _ = _masks(res)
    x = _cl(torch.randn(B, c1, src, src, device=DEV))
    x2 = _cl(torch.randn(1, c2, src, src, device=DEV)) if c2 else None
    w = torch.randn(cout, C, 3, 3, device=DEV) / (3 * C ** 0.5)
    bias = torch.randn(cout, device=DEV)
    packed = hip.conv_pack_weights(w, 6, 6, (1, 1), "f16")
    assert packed.compute == "f16" and (getattr(packed, "tile3", None) is not None) != (getattr(packed, "_tile3_src", None) is not None)
    full_in = x if x2 is None else torch.cat([x, x2], 1)
    if up:
        full_in = F.interpolate(full_in, scale_factor=2.0, mode="nearest")
    full_in = _cl(full_in)
    scale, shift = torch.randn(1, C, 1, 1, device=DEV), torch.randn(1, C, 1, 1, device=DEV)
    os_, oh_ = torch.randn(cout, device=DEV), torch.randn(cout, device=DEV)

    def run(**kw):
        outs = []
        for flag in (True, False):
            hip.TILE3 = flag
            try:
                outs.append(hip.gather_conv_cl(x, x2, (6, 6), idx, kw.get("sc"), kw.get("sh"), kw.get("act", "identity"), packed, bias,
                                               cout, (3, 3), (1, 1), out_affine=kw.get("oa"), upsample2x=up))
            finally:
                hip.TILE3 = None
        return outs

    n0 = hip.launch_count()
    for act, sc, sh in (("swish", scale, shift), ("identity", scale, shift), ("identity", None, None)):
        tiles = hip.gather_cl(full_in, 6, 6, idx, sc, sh, act)
        want = F.conv2d(tiles.half().double(), w.half().double(), bias.double()).float()
        new, old = run(sc=sc, sh=sh, act=act)
        assert new is not None and old is not None
        bad = (new.double() - want.double()).abs() > 3e-4 * (1.0 + want.double().abs())
        assert float(bad.double().mean()) < 1e-3, float(bad.double().mean())
        torch.testing.assert_close(new, old, rtol=0, atol=2e-3)  # (two kernels' swish_fast may round a staged value to neighbouring halves)
        assert float((new - old).abs().mean()) < 2e-5
        new, old = run(sc=sc, sh=sh, act=act, oa=(os_, oh_, "swish"))
        assert float((new - F.silu(want * os_.view(1, -1, 1, 1) + oh_.view(1, -1, 1, 1))).abs().mean()) < 5e-5
        assert float((new - old).abs().mean()) < 2e-5
    assert hip.launch_count() - n0 >= 12
    residual = _cl(torch.randn(B, cout, res, res, device=DEV))
    ts0, tt0, ts1, tt1 = (torch.randn(cout, device=DEV) for _ in range(4))
    outs = []
    for flag in (True, False):
        hip.TILE3 = flag
        try:
            tw = [(_cl(torch.zeros(B, cout, res, res, device=DEV)), ts0, tt0), (_cl(torch.zeros(B, cout, res, res, device=DEV)), ts1, tt1)]
            o = hip.gather_conv_cl(x, x2, (6, 6), idx, scale, shift, "swish", packed, bias, cout, (3, 3), (1, 1),
                                   full=dict(offset=(1, 1), out_res=(res, res), residual=residual), upsample2x=up, twins=tw,
                                   out=_cl(torch.zeros(B, cout, res, res, device=DEV)))
            outs.append((o, tw[0][0], tw[1][0]))
        finally:
            hip.TILE3 = None
    for a_, b_ in zip(outs[0], outs[1]):
        assert float((a_ - b_).abs().mean()) < 2e-5 and float((a_ - b_).abs().max()) < 5e-3
    assert float(outs[0][0].abs().max()) > 0.1 and float(outs[0][1].abs().max()) > 0.01


@pytest.mark.parametrize("cache_f16", [False, True])
def test_tile_conv3_f16_scatter_gather_to_full(hip, cache_f16):
    """Source 2 on fp16 operands into the persistent full tensor: bias, residual, block residual, a twin -- with the cached tensor
    and the cached shortcut tensor stored in fp32 or in halves (SIGEModel.set_cache_dtype("f16")) -- against the conv_mfma.hpp fp16
    launch (the same operand roundings, raw staging: equal up to fp32 summation order) and, tile by tile, against an fp64 conv of
    the fp16-rounded scatter_gather tiles."""
    torch.manual_seed(11)
    B, C, cout, res = 1, 128, 128, 64
    idx, idx1 = _masks(res)
    smap = hip.get_scatter_map(res, res, 6, 6, 3, 3, 1, 1, 1, 1, idx)
    y32 = _cl(torch.randn(B, C, res, res, device=DEV))
    y = _cl(y32.half()) if cache_f16 else y32
    t4 = _cl(torch.randn(B * idx.shape[0], C, 4, 4, device=DEV))
    w = torch.randn(cout, C, 3, 3, device=DEV) / (3 * C ** 0.5)
    bias = torch.randn(cout, device=DEV)
    packed = hip.conv_pack_weights(w, 6, 6, (1, 1), "f16")
    y1_32 = _cl(torch.randn(B, cout, res, res, device=DEV))
    y1 = _cl(y1_32.half()) if cache_f16 else y1_32
    x1 = _cl(torch.randn(B * idx1.shape[0], cout, 4, 4, device=DEV))
    table1 = hip.tile_table(idx1, (0, 0), (1, 1), (4, 4), (res, res))
    cache_out = _cl(torch.randn(B, cout, res, res, device=DEV))
    ts0, tt0 = torch.randn(cout, device=DEV), torch.randn(cout, device=DEV)
    for block_res in (False, True):
        outs = []
        for flag in (True, False):
            hip.TILE3 = flag
            try:
                out = cache_out.clone(memory_format=torch.preserve_format)
                tw = [(_cl(torch.zeros(B, cout, res, res, device=DEV)), ts0, tt0)]
                # (an fp16 residual only exists as the cached shortcut tensor of a fused ScatterWithBlockResidual)
                r = y1 if (block_res or not cache_f16) else y1_32
                o = hip.scatter_gather_conv_scatter_cl(t4, y, (6, 6), idx, smap, None, None, "identity", packed, bias, cout, (3, 3), (1, 1),
                                                       out, residual=r, x1=x1 if block_res else None,
                                                       table1=table1 if block_res else None, twins=tw)
                assert o is not None
                outs.append((o.clone(), tw[0][0]))
            finally:
                hip.TILE3 = None
        torch.testing.assert_close(outs[0][0], outs[1][0], rtol=0, atol=2e-4)
        torch.testing.assert_close(outs[0][1], outs[1][1], rtol=0, atol=2e-4)
        yv = y.float() if cache_f16 else y
        sg = hip.scatter_gather_cl(t4, _cl(yv), 6, 6, idx, smap)
        conv = F.conv2d(sg.half().double(), w.half().double(), bias.double()).float()
        tab = table1.cpu()
        rv = (y1.float() if (cache_f16 and block_res) else y1_32)
        for n in (0, 1, idx.shape[0] // 2, idx.shape[0] - 2):
            h0, w0 = int(idx[n, 0]) + 1, int(idx[n, 1]) + 1
            if h0 >= res or w0 >= res:
                continue
            h1, w1 = min(h0 + 4, res), min(w0 + 4, res)
            want = conv[n][:, :h1 - h0, :w1 - w0] + rv[0, :, h0:h1, w0:w1]
            if block_res:
                t1 = int(tab[h0 // 4, w0 // 4])
                if t1 >= 0:
                    want = want + (x1[t1][:, :h1 - h0, :w1 - w0] - rv[0, :, h0:h1, w0:w1])
            torch.testing.assert_close(outs[0][0][0, :, h0:h1, w0:w1], want, rtol=0, atol=3e-4)
        cover = torch.zeros(res, res, dtype=torch.bool)
        for h0, w0 in idx.cpu().tolist():
            cover[max(h0 + 1, 0):h0 + 5, max(w0 + 1, 0):w0 + 5] = True
        assert torch.equal(outs[0][0][0][:, ~cover], cache_out[0][:, ~cover])


def test_tile_conv3_f16_router_decides_from_the_tile_count(hip):
    """The routing entry points (sige_hip_gather_conv_nhwc_v3_f16c): below TILE3_MIN_BLOCKS_F16 workgroups the conv_mfma.hpp launch,
    bit for bit; from the threshold on the v3 kernel, bit for bit the forced v3 launch.  A SPARSE tile list (fewer tiles than 4 x 4
    cells) is routed from 128 workgroups on whatever the general threshold (block_conv.hip kTile3F16SparseMin); a dense one is not."""
    torch.manual_seed(5)
    C, res = 128, 64
    idx, _ = _masks(res)
    x = _cl(torch.randn(1, C, res, res, device=DEV))

    def setup(cout, tiles):
        w = torch.randn(cout, C, 3, 3, device=DEV) / (3 * C ** 0.5)
        packed = hip.conv_pack_weights(w, 6, 6, (1, 1), "f16")
        call = lambda: hip.gather_conv_cl(x, None, (6, 6), tiles, None, None, "identity", packed, None, cout, (3, 3), (1, 1))  # noqa: E731
        return call, ((tiles.shape[0] + 1) // 2) * (cout // 64)

    def forced_and_old(call):
        try:
            hip.TILE3 = True
            forced = call().clone()
            hip.TILE3 = False
            old = call().clone()
        finally:
            hip.TILE3 = None
        d = float((forced - old).abs().max())
        assert d < 1e-3, d
        distinct.append(not torch.equal(forced, old))  # (the two kernels sum in different orders: normally not the same bits)
        return forced, old

    distinct = []

    keep = hip.TILE3_MIN_BLOCKS_F16
    try:
        # below the sparse rule's 128 workgroups: the general threshold decides
        call, blocks = setup(64, idx)
        assert blocks < 128
        forced, old = forced_and_old(call)
        hip.TILE3_MIN_BLOCKS_F16 = blocks
        assert torch.equal(call(), forced)
        hip.TILE3_MIN_BLOCKS_F16 = blocks + 1
        assert torch.equal(call(), old)
        # a sparse list with >= 128 workgroups goes to v3 whatever the general threshold ...
        call, blocks = setup(128, idx)
        assert 128 <= blocks and idx.shape[0] * 16 < res * res
        forced, old = forced_and_old(call)
        hip.TILE3_MIN_BLOCKS_F16 = 100000
        assert torch.equal(call(), forced)
        # ... a DENSE list (every tile active: the dense layers of a sparse forward) does not
        dense = hip.all_tiles(res, res, (4, 4), (1, 1), (1, 1), DEV)
        call, blocks = setup(128, dense)
        assert blocks >= 128 and dense.shape[0] * 16 >= res * res
        forced, old = forced_and_old(call)
        assert torch.equal(call(), old)
        hip.TILE3_MIN_BLOCKS_F16 = blocks
        assert torch.equal(call(), forced)
    finally:
        hip.TILE3, hip.TILE3_MIN_BLOCKS_F16 = None, keep
    assert any(distinct), "forced v3 and conv_mfma.hpp gave the same bits in every case: the routing checks above prove nothing"


@pytest.mark.parametrize("compute", ["f32", "f16"])
def test_tile_conv3_scatter_gather_with_cached_affine(hip, compute):
    """Source 2 with the cached GroupNorm affine + SiLU in the staging path (the Stable-Diffusion U-Net's conv2: a per-sample affine,
    CFG batch 2 -- sige_openaimodel.py; scatter_gather.cpp:58-84 applies scale / shift / activation to the re-tiled window, padding
    stays 0): the v3 kernel against conv_mfma.hpp's launch of the same call and, tile by tile, against an fp64 conv of the
    standalone scatter_gather's tiles (pinned to the oracle)."""
    torch.manual_seed(21)
    B, C, cout, res = 2, 128, 128, 64
    idx, idx1 = _masks(res)
    if idx.shape[0] % 2:
        idx = idx[:-1].contiguous()  # (a per-sample affine needs whole tile pairs per image)
    smap = hip.get_scatter_map(res, res, 6, 6, 3, 3, 1, 1, 1, 1, idx)
    y = _cl(torch.randn(B, C, res, res, device=DEV))
    t4 = _cl(torch.randn(B * idx.shape[0], C, 4, 4, device=DEV))
    w = torch.randn(cout, C, 3, 3, device=DEV) / (3 * C ** 0.5)
    bias = torch.randn(cout, device=DEV)
    packed = hip.conv_pack_weights(w, 6, 6, (1, 1), compute)
    scale, shift = torch.randn(B, C, 1, 1, device=DEV), torch.randn(B, C, 1, 1, device=DEV)
    y1 = _cl(torch.randn(B, cout, res, res, device=DEV))
    cache_out = _cl(torch.randn(B, cout, res, res, device=DEV))
    outs = []
    n0 = hip.launch_count()
    for flag in (True, False):
        hip.TILE3 = flag
        try:
            out = cache_out.clone(memory_format=torch.preserve_format)
            o = hip.scatter_gather_conv_scatter_cl(t4, y, (6, 6), idx, smap, scale, shift, "swish", packed, bias, cout, (3, 3), (1, 1), out,
                                                   residual=y1)
            assert o is not None
            outs.append(o.clone())
        finally:
            hip.TILE3 = None
    assert hip.launch_count() == n0 + 2
    tol = 2e-4 if compute == "f32" else 3e-3
    assert float((outs[0] - outs[1]).abs().max()) < tol and float((outs[0] - outs[1]).abs().mean()) < 2e-5
    sg = hip.scatter_gather_cl(t4, y, 6, 6, idx, smap, scale, shift, "swish")
    wd, sgd = (w.double(), sg.double()) if compute == "f32" else (w.half().double(), sg.half().double())
    conv = F.conv2d(sgd, wd, bias.double()).float()
    N = idx.shape[0]
    for b in range(B):
        for n in (0, 1, N // 2, N - 2):
            h0, w0 = int(idx[n, 0]) + 1, int(idx[n, 1]) + 1
            if h0 >= res or w0 >= res:
                continue
            h1, w1 = min(h0 + 4, res), min(w0 + 4, res)
            want = conv[b * N + n][:, :h1 - h0, :w1 - w0] + y1[b, :, h0:h1, w0:w1]
            got = outs[0][b, :, h0:h1, w0:w1]
            if compute == "f32":
                torch.testing.assert_close(got, want, rtol=0, atol=3e-4)
            else:  # (swish_fast vs the standalone kernel's swish: a staged value may round to the neighbouring half)
                assert float((got - want).abs().mean()) < 5e-5 and float((got - want).abs().max()) < 5e-3


# ---- the first conv on the active windows only (sige_hip_conv3x3_small_cin_tiles_nhwc_f32) --------------------------------------------
def test_conv_in_on_the_active_windows_only(hip):
    """The tile-list form of the first conv: inside every 6 x 6 window of the index list (border windows clipped) bit for bit the
    dense launch's values, outside them the buffer untouched."""
    torch.manual_seed(2)
    B, res, cout = 2, 64, 128
    idx, _ = _masks(res)
    x = _cl(torch.randn(B, 3, res, res, device=DEV))
    w, b = torch.randn(cout, 3, 3, 3, device=DEV) / 5, torch.randn(cout, device=DEV)
    dense = hip.conv3x3_small_cin_cl(x, w, b)
    torch.testing.assert_close(dense, F.conv2d(x, w, b, 1, 1), rtol=0, atol=2e-5)
    buf = _cl(torch.full((B, cout, res, res), 7.0, device=DEV))
    n0 = hip.launch_count()
    got = hip.conv3x3_small_cin_cl(x, w, b, tiles=(idx, (6, 6)), out=buf)
    assert got is buf and hip.launch_count() == n0 + 1
    cover = torch.zeros(res, res, dtype=torch.bool)
    for h0, w0 in idx.cpu().tolist():
        cover[max(h0, 0):h0 + 6, max(w0, 0):w0 + 6] = True
    assert cover.any() and not cover.all()
    assert torch.equal(buf[:, :, cover], dense[:, :, cover])
    assert bool((buf[:, :, ~cover] == 7.0).all())


@pytest.mark.selfcheck
def test_ddpm_forward_sparse_conv_in_on_off(hip):
    """The benchmark network with conv_in evaluated on the active windows only and with the dense conv_in the reference runs
    (sige_fused_unet.py:395): the SAME output bit for bit -- a sparse pass reads hs[0] through Gather windows only -- and both within
    the north_star tolerance of the CPU oracle; also after a mask change (the buffer then holds stale values outside the new
    windows: unread)."""
    import bench
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval().to(DEV).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = bench.make_inputs()
    t = torch.zeros(1, device=DEV)
    ratios = (0.15, 0.012)
    _, wants = util.ddpm_cpu_oracle([bench.edit_mask(r) for r in ratios])
    try:
        with util.native_full_pass(), torch.no_grad():
            model.set_mode("full")
            model(_cl(x0.to(DEV)), t)
            for ratio, want in zip(ratios, wants):
                mask = bench.edit_mask(ratio)
                x1 = _cl((x0 + noise * mask).to(DEV))
                model.set_masks(downsample_mask(dilate_mask(mask.to(DEV), 5), 8))
                model.set_mode("sparse")
                outs = {}
                for on in (True, False):
                    model.SPARSE_CONV_IN = on
                    for _ in range(3):
                        model(x1, t)
                    outs[on] = model(x1, t).clone()
                assert getattr(model, "_h0_buf", None) is not None  # (the tile-list form DID run: no silent dense fallback)
                assert torch.equal(outs[True], outs[False]), float((outs[True] - outs[False]).abs().max())
                err = util.record_margin("sparse_conv_in", "ratio %g vs cpu oracle" % ratio, (outs[True].cpu() - want).abs().max(), util.CONV_ATOL)
                assert err <= util.CONV_ATOL, (ratio, err)
    finally:
        model.SPARSE_CONV_IN = True


@pytest.mark.gpu
def test_tuned_token_gemms_table_loads_and_computes_the_same_product():
    """sige_amd/workloads/gemm_tuning.py: the TunableOp table of SD's token GEMMs is accepted by this stack (else skipped: PyTorch
    rejects a table whose validator lines -- library versions, gfx950 -- differ), holds the benchmarked shapes, and a Linear of one
    of them gives the product the default solution gives (every candidate is an fp32 GEMM)."""
    import torch.nn.functional as F
    from sige_amd.workloads import gemm_tuning

    g = torch.Generator().manual_seed(3)
    x = torch.randn(8192, 320, generator=g).cuda()
    w = torch.randn(2560, 320, generator=g).cuda()
    b = torch.randn(2560, generator=g).cuda()
    want = F.linear(x, w, b)
    ok = gemm_tuning.enable_tuned_gemms()
    try:
        if not ok:
            assert not torch.cuda.tunable.is_enabled()
            pytest.skip("PyTorch rejected the table (validators of another stack)")
        assert torch.cuda.tunable.is_enabled() and not torch.cuda.tunable.tuning_is_enabled()
        assert len(torch.cuda.tunable.get_results()) >= 40
        got = F.linear(x, w, b)
    finally:
        gemm_tuning.disable_tuned_gemms()
    assert not torch.cuda.tunable.is_enabled()
    exact = x.double() @ w.double().t() + b.double()
    assert float((got.double() - exact).abs().max()) <= 1e-3 and float((want.double() - exact).abs().max()) <= 1e-3
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-3)
