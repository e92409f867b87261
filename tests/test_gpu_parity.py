"""Parity of the HIP path (through the C ABI, via sige_amd.hip) with the
reference: golden vectors from the real reference, the CPU oracle on larger
seeded inputs, and size-independent properties at BASELINE.json's full sizes.

Tolerances: index tensors and identity-activation copies/adds
bit-exact; swish <= 1e-6 rel; conv-containing paths <= 1e-3 abs (north_star)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import util
from tests.golden_cases import CASES

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from sige_amd import hip as h

    h.lib()  # fails loudly if the extension is not built
    return h


def _close(case, got, want):
    got = got.cpu()
    if case["act"] == "swish":
        torch.testing.assert_close(got, want, rtol=util.SWISH_RTOL, atol=util.SWISH_ATOL)
    else:
        assert torch.equal(got, want)


def test_native_library_is_loaded(hip):
    assert hip.lib().sige_hip_version() == 309
    arch = hip.lib().sige_hip_device_arch()
    assert arch is not None and arch.decode().startswith("gfx950"), arch
    assert "libsige_hip.so" in open("/proc/self/maps").read()


@pytest.mark.parametrize("case", CASES, ids=util.case_ids())
def test_golden_cases(hip, case):
    g = case["geom"]
    d = util.tensors(case, DEV)
    idx_ref = util.ref(case, "idx")
    idx = hip.reduce_mask(d["mask"], g.block, g.block_stride, g.offset)
    assert torch.equal(idx.cpu(), idx_ref)

    got = hip.gather(d["x"], g.block[0], g.block[1], idx, d["scale"], d["shift"], case["act"], case["act_first"])
    _close(case, got, util.ref(case, "gather"))

    tiles_in = util.ref(case, "gather", DEV)
    Cout = case["cout"]
    packed = hip.conv_pack_weights(d["weight"], g.block[0], g.block[1], g.stride)
    if packed is not None:
        conv = hip.block_conv(tiles_in, packed, d["bias"], Cout, g.kernel, g.stride)
        torch.testing.assert_close(conv.cpu(), util.ref(case, "conv"), rtol=0, atol=util.CONV_ATOL)
        torch.testing.assert_close(conv.cpu(), util.ref(case, "conv"), rtol=0, atol=2e-5)  # fp32 MFMA: far inside
    direct = hip.block_conv_direct(tiles_in, d["weight"], d["bias"], g.stride, 1)
    torch.testing.assert_close(direct.cpu(), util.ref(case, "conv"), rtol=0, atol=2e-5)

    tiles = util.ref(case, "conv", DEV)
    args = (g.offset[0], g.offset[1], g.stride[0], g.stride[1], idx)
    Ho, Wo = d["out_res"]
    table = hip.tile_table(idx, g.offset, g.stride, g.out_tile, (Ho, Wo))
    n = idx.shape[0]
    for key, res in (("scatter", None), ("scatter_res", d["residual"]), ("scatter_resc", d["residual_c"])):
        assert torch.equal(hip.scatter(tiles, d["y"], *args, res).cpu(), util.ref(case, key)), key
        assert torch.equal(hip.scatter_fused(tiles, d["y"], table, n, res).cpu(), util.ref(case, key)), key + " fused"

    smap = hip.get_scatter_map(Ho, Wo, *g.block, *g.kernel, *g.offset, *g.stride, idx)
    assert torch.equal(smap.cpu(), util.ref(case, "map"))
    sg = hip.scatter_gather(tiles, d["y"], g.block[0], g.block[1], idx, smap, d["scale2"], d["shift2"], case["act"],
                            case["act_first"])
    _close(case, sg, util.ref(case, "sg"))

    idx1 = hip.reduce_mask(util.shortcut_mask(case, d).to(DEV), (4, 4), (4, 4), (0, 0))
    assert torch.equal(idx1.cpu(), util.ref(case, "idx1"))
    x1 = util.x1_tiles(case, d, idx1.shape[0], DEV)
    want = util.ref(case, "swbr")
    assert torch.equal(hip.scatter_with_block_residual(tiles, d["y"], x1, d["y1"], *args[:4], idx, idx1).cpu(), want)
    t1 = hip.tile_table(idx1, (0, 0), (1, 1), (4, 4), (Ho, Wo))
    fused = hip.scatter_with_block_residual_fused(tiles, d["y"], x1, d["y1"], table, n, t1, idx1.shape[0])
    assert torch.equal(fused.cpu(), want)


def _fixture_names():
    return sorted({k.split("/")[0] for k in util.golden("masks").files})


@pytest.mark.parametrize("name", _fixture_names())
def test_reduce_mask_fixture_masks(hip, name):
    """Device reduce_mask on the reference's own fixture masks (up to 512x1024)."""
    g = util.golden("masks")
    mask = util.unpack(g[name + "/mask"], g[name + "/shape"]).to(DEV)
    for gname, (b, s, p) in {"b6s4p1": (6, 4, 1), "b4s4p0": (4, 4, 0), "b5s4p0": (5, 4, 0), "b5s4p1": (5, 4, 1)}.items():
        got = hip.reduce_mask(mask, (b, b), (s, s), (p, p))
        assert torch.equal(got.cpu(), torch.from_numpy(g["%s/reduce/%s" % (name, gname)])), gname


def _square_mask(ratio, H=256, W=256, top=100, left=90):
    side = int(round((ratio ** 0.5) * H))
    m = torch.zeros(H, W, dtype=torch.bool)
    m[top:top + side, left:left + side] = True
    return m


@pytest.mark.parametrize("C,res,ratio,B", [(128, 64, 0.05, 1), (64, 128, 0.15, 2), (256, 32, 0.3, 1)])
def test_against_oracle_mid_size(hip, C, res, ratio, B):
    """Seeded mid-size inputs: HIP vs the CPU oracle for the whole op chain."""
    from sige_amd.utils import reduce_mask

    torch.manual_seed(C + res)
    mask = _square_mask(ratio, res, res, res // 3, res // 4)
    idx = reduce_mask(mask, 6, 4, 1)
    idx1 = reduce_mask(mask, 4, 4, 0)
    x = torch.randn(B, C, res, res)
    y, y1 = torch.randn(B, C, res, res), torch.randn(B, C, res, res)
    scale, shift = torch.randn(1, C, 1, 1), torch.randn(1, C, 1, 1)
    w, bias = torch.randn(C, C, 3, 3) / (3 * C ** 0.5), torch.randn(C)
    x1 = torch.randn(B * idx1.shape[0], C, 4, 4)
    g_o = oracle.gather(x, 6, 6, idx, scale, shift, "swish", False)
    c_o = oracle.block_conv(g_o, w, bias, 1)
    m_o = oracle.get_scatter_map(res, res, 6, 6, 3, 3, 1, 1, 1, 1, idx)
    sg_o = oracle.scatter_gather(c_o, y, 6, 6, idx, m_o, scale, shift, "swish", False)
    sw_o = oracle.scatter_with_block_residual(c_o, y, x1, y1, 1, 1, 1, 1, idx, idx1)

    to = lambda t: t.to(DEV)  # noqa: E731
    idx_d, idx1_d = to(idx), to(idx1)
    g_h = hip.gather(to(x), 6, 6, idx_d, to(scale), to(shift), "swish", False)
    torch.testing.assert_close(g_h.cpu(), g_o, rtol=util.SWISH_RTOL, atol=util.SWISH_ATOL)
    packed = hip.conv_pack_weights(to(w), 6, 6, (1, 1))
    c_h = hip.block_conv(to(g_o), packed, to(bias), C, (3, 3), (1, 1))
    torch.testing.assert_close(c_h.cpu(), c_o, rtol=0, atol=util.CONV_ATOL)
    m_h = hip.get_scatter_map(res, res, 6, 6, 3, 3, 1, 1, 1, 1, idx_d)
    assert torch.equal(m_h.cpu(), m_o)
    sg_h = hip.scatter_gather(to(c_o), to(y), 6, 6, idx_d, m_h, to(scale), to(shift), "swish", False)
    torch.testing.assert_close(sg_h.cpu(), sg_o, rtol=util.SWISH_RTOL, atol=util.SWISH_ATOL)
    sw_h = hip.scatter_with_block_residual(to(c_o), to(y), to(x1), to(y1), 1, 1, 1, 1, idx_d, idx1_d)
    assert torch.equal(sw_h.cpu(), sw_o)
    t0 = hip.tile_table(idx_d, (1, 1), (1, 1), (4, 4), (res, res))
    t1 = hip.tile_table(idx1_d, (0, 0), (1, 1), (4, 4), (res, res))
    sw_f = hip.scatter_with_block_residual_fused(to(c_o), to(y), to(x1), to(y1), t0, idx.shape[0], t1, idx1.shape[0])
    assert torch.equal(sw_f.cpu(), sw_o)


@pytest.mark.parametrize("cin,cout,k,stride,R", [(128, 128, 3, 1, 6), (256, 128, 3, 1, 6), (128, 256, 1, 1, 4),
                                                (512, 256, 1, 1, 4), (128, 128, 3, 2, 5), (16, 32, 3, 1, 6),
                                                (36, 128, 3, 1, 6), (40, 24, 1, 1, 4), (6, 70, 3, 2, 5)])
def test_block_conv_vs_torch_and_oracle(hip, cin, cout, k, stride, R):
    """MFMA stacked-block conv at DDPM / GauGAN channel counts, odd T, channel
    counts that are not multiples of the 32-wide chunks."""
    torch.manual_seed(cin * cout + k)
    for T in (1, 7, 124):
        x = torch.randn(T, cin, R, R, device=DEV)
        w = torch.randn(cout, cin, k, k, device=DEV) / (k * cin ** 0.5)
        b = torch.randn(cout, device=DEV)
        packed = hip.conv_pack_weights(w, R, R, (stride, stride))
        assert packed is not None
        got = hip.block_conv(x, packed, b, cout, (k, k), (stride, stride))
        want = oracle.block_conv(x.cpu(), w.cpu(), b.cpu(), stride)
        torch.testing.assert_close(got.cpu(), want, rtol=0, atol=util.CONV_ATOL)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride).float()
        torch.testing.assert_close(got, ref, rtol=0, atol=1e-4)
        nob = hip.block_conv(x, packed, None, cout, (k, k), (stride, stride))
        torch.testing.assert_close(nob, got - b.view(1, -1, 1, 1), rtol=0, atol=1e-5)


@pytest.mark.parametrize("C,cout,res,ratio,B", [(128, 128, 64, 0.05, 1), (64, 96, 128, 0.15, 2), (256, 256, 32, 0.3, 1),
                                                 (48, 40, 64, 0.02, 1), (128, 128, 256, 0.15, 1)])
def test_fused_gather_conv_equals_two_kernels(hip, C, cout, res, ratio, B):
    """gather_conv / scatter_gather_conv == gather / scatter_gather followed by
    block_conv for the three tile geometries: bit for bit without activation (same
    staging values, same MFMA order); within 1e-5 with SiLU (the fused staging
    path evaluates it with v_exp_f32 / v_rcp_f32, ~1e-6 relative)."""
    from sige_amd.utils import reduce_mask

    torch.manual_seed(C + res + B)
    mask = _square_mask(ratio, res, res, res // 3, res // 4).to(DEV)
    mask[0, 0] = mask[res - 1, res - 1] = True  # border tiles: zero fill inside the fused prologue
    x = torch.randn(B, C, res, res, device=DEV)
    y = torch.randn(B, C, res, res, device=DEV)
    bias = torch.randn(cout, device=DEV)
    for k, s, blk, off, scale_shape in ((3, 1, 6, 1, (1, C, 1, 1)), (1, 1, 4, 0, None), (3, 2, 5, 0, (B, C, 1, 1)),
                                        (3, 1, 6, 1, None)):
        idx = reduce_mask(mask, blk, 4, off)
        w = torch.randn(cout, C, k, k, device=DEV) / (k * C ** 0.5)
        packed = hip.conv_pack_weights(w, blk, blk, (s, s))
        scale = None if scale_shape is None else torch.randn(*scale_shape, device=DEV)
        shift = None if scale_shape is None else torch.randn(*scale_shape, device=DEV)
        act = "swish" if scale_shape is not None else "identity"
        tiles = hip.gather(x, blk, blk, idx, scale, shift, act, False)
        two = hip.block_conv(tiles, packed, bias, cout, (k, k), (s, s))
        one = hip.gather_conv(x, (blk, blk), idx, scale, shift, act, packed, bias, cout, (k, k), (s, s))
        if act == "identity":
            assert torch.equal(one, two), (k, s, (one - two).abs().max().item())
        else:
            torch.testing.assert_close(one, two, rtol=0, atol=1e-5)
        ref = torch.nn.functional.conv2d(tiles.double(), w.double(), bias.double(), s).float()
        torch.testing.assert_close(one, ref, rtol=0, atol=1e-4)
        if k == 3 and s == 1:
            smap = hip.get_scatter_map(res, res, 6, 6, 3, 3, 1, 1, 1, 1, idx)
            t4 = torch.randn(B * idx.shape[0], C, 4, 4, device=DEV)
            sg = hip.scatter_gather(t4, y, 6, 6, idx, smap, scale, shift, act, False)
            two = hip.block_conv(sg, packed, bias, cout, (3, 3), (1, 1))
            one = hip.scatter_gather_conv(t4, y, (6, 6), idx, smap, scale, shift, act, packed, bias, cout, (3, 3),
                                          (1, 1))
            if act == "identity":
                assert torch.equal(one, two)
            else:
                torch.testing.assert_close(one, two, rtol=0, atol=1e-5)


@pytest.mark.parametrize("mt,nb", [(16, 1), (16, 2), (32, 1), (32, 2)])
@pytest.mark.parametrize("cin,c2,cout,k,stride,blk,off", [(128, 0, 128, 3, 1, 6, 1), (64, 40, 200, 3, 1, 6, 1), (96, 0, 64, 1, 1, 4, 0),
                                                          (256, 44, 40, 1, 1, 4, 0), (64, 0, 72, 3, 2, 5, 0), (70, 0, 24, 3, 1, 6, 1)])
def test_conv_every_output_block_shape(hip, mt, nb, cin, c2, cout, k, stride, blk, off, tuning):
    """Every (pixels x channels) output block of the MFMA kernel, pinned with
    sige_hip_block_conv_force_tile, for the tile / gather / scatter_gather / NCHW
    forms: channel counts that are not multiples of the chunk, a fused torch.cat (per
    image, split on a chunk boundary), B = 2 with a per-batch affine, border tiles."""
    from sige_amd.utils import reduce_mask

    torch.manual_seed(cin + cout + mt + nb)
    B, res = 2, 48
    C = cin + c2
    mask = _square_mask(0.2, res, res, res // 3, res // 4).to(DEV)
    mask[0, 0] = mask[res - 1, res - 1] = True
    x = torch.randn(B, C, res, res, device=DEV)
    y = torch.randn(B, C, res, res, device=DEV)
    w = torch.randn(cout, C, k, k, device=DEV) / (k * C ** 0.5)
    bias = torch.randn(cout, device=DEV)
    scale, shift = torch.randn(B, C, 1, 1, device=DEV), torch.randn(B, C, 1, 1, device=DEV)
    idx = reduce_mask(mask, blk, 4, off)
    packed = hip.conv_pack_weights(w, blk, blk, (stride, stride))
    conv = lambda t: torch.nn.functional.conv2d(t.double(), w.double(), bias.double(), stride).float()  # noqa: E731
    hip.conv_force_tile(mt, nb)
    try:
        for act, sc, sh in (("swish", scale, shift), ("identity", scale, shift), ("identity", None, None)):
            tiles = hip.gather(x, blk, blk, idx, sc, sh, act, False)
            torch.testing.assert_close(hip.block_conv(tiles, packed, bias, cout, (k, k), (stride, stride)), conv(tiles),
                                       rtol=0, atol=1e-4)
            one = hip.gather_conv(x, (blk, blk), idx, sc, sh, act, packed, bias, cout, (k, k), (stride, stride))
            if one is None:  # per-batch affine with workgroups straddling images (stride-2 tiles): two-kernel form
                assert stride == 2 and sc is not None
            else:
                torch.testing.assert_close(one, conv(tiles), rtol=0, atol=1e-4)
            if c2:  # the same conv with the channels coming from two tensors, written into an NCHW tensor
                assert hip.cat_fusable(1, cin, (k, k))
                ho = res if stride == 1 else res // 2
                o = 4 if stride == 1 else 2
                t_out = conv(tiles).reshape(B, idx.shape[0], cout, o, o)
                for b in range(B):
                    residual = torch.randn(1, cout, ho, ho, device=DEV)
                    out = hip.gather_conv_nchw(x[b:b + 1, :cin].contiguous(), x[b:b + 1, cin:].contiguous(), (blk, blk), idx,
                                               None if sc is None else sc[b:b + 1], None if sh is None else sh[b:b + 1], act,
                                               packed, bias, cout, (k, k), (stride, stride), (off, off), (ho, ho), residual)
                    for n in (0, idx.shape[0] // 2, idx.shape[0] - 1):
                        h0, w0 = (int(idx[n, 0]) + off) // stride, (int(idx[n, 1]) + off) // stride
                        h1, w1 = min(h0 + o, ho), min(w0 + o, ho)
                        want = t_out[b, n][:, :h1 - h0, :w1 - w0] + residual[0, :, h0:h1, w0:w1]
                        torch.testing.assert_close(out[0, :, h0:h1, w0:w1], want, rtol=0, atol=1e-4)
            if k == 3 and stride == 1:
                smap = hip.get_scatter_map(res, res, 6, 6, 3, 3, 1, 1, 1, 1, idx)
                t4 = torch.randn(B * idx.shape[0], C, 4, 4, device=DEV)
                sg = hip.scatter_gather(t4, y, 6, 6, idx, smap, sc, sh, act, False)
                one = hip.scatter_gather_conv(t4, y, (6, 6), idx, smap, sc, sh, act, packed, bias, cout, (3, 3), (1, 1))
                assert one is not None
                torch.testing.assert_close(one, conv(sg), rtol=0, atol=1e-4)
    finally:
        hip.conv_force_tile(0, 0)


def test_deferred_fusion_in_modules():
    """Module level: with fusion on, Gather returns DeferredTiles and the ResBlock
    output equals the unfused run (deferred.FUSION = False) to 1e-5 (the fused staging path's
    SiLU uses v_exp_f32 / v_rcp_f32)."""
    from sige_amd.nn import deferred
    from sige_amd.utils import dilate_mask
    from tests.test_host_logic import ResNet

    torch.manual_seed(4)
    net = ResNet(64, 128).to(DEV).eval()
    blk = net.block
    blk.s1, blk.t1 = torch.randn(1, 64, 1, 1, device=DEV), torch.randn(1, 64, 1, 1, device=DEV)
    blk.s2, blk.t2 = torch.randn(1, 128, 1, 1, device=DEV), torch.randn(1, 128, 1, 1, device=DEV)
    orig = torch.randn(1, 64, 64, 64, device=DEV)
    mask = torch.zeros(64, 64, dtype=torch.bool, device=DEV)
    mask[20:31, 12:40] = True
    edited = orig + torch.randn_like(orig) * mask
    with torch.no_grad():
        net.set_mode("full")
        dense = net(edited)
        net(orig)
        net.set_mode("sparse")
        net.set_masks({(64, 64): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))})
        assert isinstance(blk.main_gather(edited, blk.s1, blk.t1), deferred.DeferredTiles)
        fused = net(edited)
        deferred.FUSION = False
        try:
            assert not isinstance(blk.main_gather(edited, blk.s1, blk.t1), deferred.DeferredTiles)
            unfused = net(edited)
        finally:
            deferred.FUSION = True
    torch.testing.assert_close(fused, unfused, rtol=0, atol=1e-5)
    torch.testing.assert_close(fused, dense, rtol=0, atol=util.CONV_ATOL)


@pytest.mark.oracle_parity  # (the reference computes this row with the torch op compared here)
@pytest.mark.parametrize("res,c1,c2,cout,k,stride", [(32, 256, 0, 256, 3, 1), (16, 512, 512, 512, 3, 1), (8, 512, 256, 512, 1, 1),
                                                     (32, 96, 40, 72, 3, 1), (16, 512, 0, 1536, 1, 1), (32, 256, 0, 256, 3, 2),
                                                     (256, 128, 0, 3, 3, 1), (18, 64, 0, 64, 3, 1)])
def test_dense_fused_conv_vs_torch(res, c1, c2, cout, k, stride):
    """Dense layers as all-tiles gather-conv: conv(swish(cat(x,x2)*s+t)) + residual == torch."""
    from torch import nn

    from sige_amd.nn.dense import fused_conv2d

    torch.manual_seed(res + c1 + k)
    B = 1
    conv = nn.Conv2d(c1 + c2, cout, k, stride, 0 if stride == 2 else k // 2).to(DEV)
    x = torch.randn(B, c1, res, res, device=DEV)
    x2 = torch.randn(B, c2, res, res, device=DEV) if c2 else None
    s, t = torch.randn(1, c1 + c2, 1, 1, device=DEV), torch.randn(1, c1 + c2, 1, 1, device=DEV)
    ro = res if stride == 1 else res // 2
    residual = torch.randn(B, cout, ro, ro, device=DEV)
    with torch.no_grad():
        got = fused_conv2d(conv, x, s, t, "swish", x2=x2, residual=residual, pad_bottom_right=stride == 2)
        h = x if x2 is None else torch.cat([x, x2], 1)
        h = torch.nn.functional.silu(h * s + t)
        if stride == 2:
            h = torch.nn.functional.pad(h, (0, 1, 0, 1))
        want = torch.nn.functional.conv2d(h.double(), conv.weight.double(), conv.bias.double(), stride,
                                          conv.padding).float() + residual
        torch.testing.assert_close(got, want, rtol=0, atol=2e-4)
        plain = fused_conv2d(conv, x, x2=x2, pad_bottom_right=stride == 2)
        h = x if x2 is None else torch.cat([x, x2], 1)
        if stride == 2:
            h = torch.nn.functional.pad(h, (0, 1, 0, 1))
        torch.testing.assert_close(plain, conv(h), rtol=0, atol=2e-4)


@pytest.mark.oracle_parity  # (the reference computes this row with the torch op compared here)
@pytest.mark.parametrize("B,C,hw", [(1, 512, 16), (1, 512, 8), (2, 64, 12), (1, 48, 32)])
def test_attention_vs_torch(hip, B, C, hw):
    """AttnBlock core (bmm -> softmax -> bmm on NCHW q, k, v) against torch in fp64."""
    torch.manual_seed(C + hw)
    qkv = torch.randn(B, 3 * C, hw, hw, device=DEV)
    assert hip.attention_supported(C, hw * hw)
    got = hip.attention(qkv, C ** -0.5)
    q, k, v = qkv.double().reshape(B, 3, C, hw * hw).unbind(1)
    attn = torch.softmax(torch.bmm(q.transpose(1, 2), k) * (C ** -0.5), dim=2)
    want = torch.bmm(v, attn.transpose(1, 2)).reshape(B, C, hw, hw).float()
    torch.testing.assert_close(got, want, rtol=0, atol=2e-5)


@pytest.mark.oracle_parity  # (the reference computes this row with the torch op compared here)
@pytest.mark.parametrize("shape,groups", [((1, 128, 256, 256), 32), ((2, 64, 17, 23), 32), ((1, 512, 8, 8), 32)])
def test_group_norm_affine_vs_torch(hip, shape, groups):
    torch.manual_seed(shape[1])
    x = torch.randn(*shape, device=DEV) * 3 + 1.5
    gamma, beta = torch.randn(shape[1], device=DEV), torch.randn(shape[1], device=DEV)
    scale, shift = hip.group_norm_affine(x, groups, 1e-6, gamma, beta)
    want = torch.nn.functional.group_norm(x.double(), groups, gamma.double(), beta.double(), 1e-6).float()
    torch.testing.assert_close(x * scale + shift, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("ratio", [0.012, 0.15])
def test_ddpm_unet_gpu_vs_oracle_backend(ratio):
    """Model level (BASELINE configs[1] shape, ch 32 to keep the CPU side short): the
    sparse forward on the GPU (HIP kernels, fused paths) equals the same network on the
    CPU with the oracle as native backend, within 1e-3."""
    from sige_amd import runtime
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig(ch=32)).eval()
    x0 = torch.randn(1, 3, 256, 256)
    mask = _square_mask(ratio)
    x1 = x0 + torch.randn(1, 3, 256, 256) * mask
    t = torch.zeros(1)

    def run(device):
        model.to(device)
        model.clear_cache()
        with torch.no_grad():
            model.set_mode("full")
            full = model(x0.to(device), t.to(device))
            model.set_masks(downsample_mask(dilate_mask(mask.to(device), 5), 8))
            model.set_mode("sparse")
            return full.cpu(), model(x1.to(device), t.to(device)).cpu()

    runtime.register_backend("cpu", oracle)
    try:
        full_c, sparse_c = run("cpu")
    finally:
        runtime.unregister_backend("cpu")
    full_g, sparse_g = run(DEV)
    torch.testing.assert_close(full_g, full_c, rtol=0, atol=util.CONV_ATOL)
    torch.testing.assert_close(sparse_g, sparse_c, rtol=0, atol=util.CONV_ATOL)
    assert (sparse_c - full_c).abs().max() > 1e-2


def test_block_conv_direct_groups(hip):
    torch.manual_seed(5)
    x = torch.randn(9, 24, 6, 6, device=DEV)
    w = torch.randn(24, 1, 3, 3, device=DEV)
    b = torch.randn(24, device=DEV)
    got = hip.block_conv_direct(x, w, b, (1, 1), 24)  # depthwise (GAN-Compression SIGESeparableConv2d)
    want = oracle.block_conv(x.cpu(), w.cpu(), b.cpu(), 1, groups=24)
    torch.testing.assert_close(got.cpu(), want, rtol=0, atol=1e-5)


# ---- BASELINE.json full sizes: size-independent properties -----------------
@pytest.mark.parametrize("ratio", [0.012, 0.05, 0.15])
@pytest.mark.parametrize("C", [128, 256])
def test_full_size_properties(hip, ratio, C):
    """DDPM-256 op shapes (x [1,C,256,256], square edit):
    round trips, idempotence and linearity that hold for any size."""
    from sige_amd.utils import reduce_mask

    torch.manual_seed(int(ratio * 1000) + C)
    mask = _square_mask(ratio).to(DEV)
    idx6 = reduce_mask(mask, 6, 4, 1)
    idx4 = reduce_mask(mask, 4, 4, 0)
    expected = {0.012: (72, 56), 0.05: (240, 225), 0.15: (676, 650)}[ratio]  # N6 / N4
    assert (idx6.shape[0], idx4.shape[0]) == expected
    x = torch.randn(1, C, 256, 256, device=DEV)
    # (1) gather(4x4 tiles) -> scatter into x itself is the identity (encode -> decode round trip)
    t4 = hip.gather(x, 4, 4, idx4)
    assert torch.equal(hip.scatter(t4, x, 0, 0, 1, 1, idx4, None), x)
    # (2) interior of the 6x6 gather == the 4x4 tile at origin+1 (halo consistency)
    t6 = hip.gather(x, 6, 6, idx6)
    inner = hip.gather(x, 4, 4, idx6 + 1)
    assert torch.equal(t6[:, :, 1:5, 1:5], inner)
    # (3) scatter into zeros then gather back returns the tiles (decode -> encode)
    zeros = torch.zeros_like(x)
    tbl = hip.tile_table(idx6, (1, 1), (1, 1), (4, 4), (256, 256))
    s = hip.scatter_fused(inner, zeros, tbl, idx6.shape[0], None)
    assert torch.equal(hip.gather(s, 4, 4, idx6 + 1), inner)
    assert torch.equal(s, hip.scatter(inner, zeros, 1, 1, 1, 1, idx6, None))  # fused == two-pass
    # (4) everything outside the active tiles is untouched; checksum of checksums
    covered = torch.zeros(256, 256, dtype=torch.bool, device=DEV)
    for h, w in (idx6 + 1).tolist():
        covered[h:h + 4, w:w + 4] = True
    y = torch.randn_like(x)
    out = hip.scatter_fused(inner, y, tbl, idx6.shape[0], None)
    assert torch.equal(out[:, :, ~covered], y[:, :, ~covered])
    assert torch.equal(out[:, :, covered], x[:, :, covered])
    # (5) linearity of scatter in the residual: scatter(t, y, res) - scatter(t, y) == res on covered pixels
    res = torch.randn_like(x)
    d = hip.scatter_fused(inner, y, tbl, idx6.shape[0], res) - out
    torch.testing.assert_close(d[:, :, covered], res[:, :, covered], rtol=0, atol=1e-5)
    assert torch.count_nonzero(d[:, :, ~covered]) == 0
    # (6) scatter_gather with an all-(-1) map is gather on y; with the real map it equals gather(scatter(...))
    smap = hip.get_scatter_map(256, 256, 6, 6, 3, 3, 1, 1, 1, 1, idx6)
    sg = hip.scatter_gather(inner, y, 6, 6, idx6, smap)
    assert torch.equal(sg, hip.gather(out, 6, 6, idx6))
    none = torch.full_like(smap, -1)
    assert torch.equal(hip.scatter_gather(inner, y, 6, 6, idx6, none), hip.gather(y, 6, 6, idx6))
    # (7) swish-affine gather == torch elementwise on the gathered tiles (zeros stay zeros)
    sc, sh = torch.randn(1, C, 1, 1, device=DEV), torch.randn(1, C, 1, 1, device=DEV)
    ga = hip.gather(x, 6, 6, idx6, sc, sh, "swish", False)
    ref = torch.nn.functional.silu(t6 * sc + sh)
    inside = hip.gather(torch.ones_like(x), 6, 6, idx6) > 0
    torch.testing.assert_close(ga, torch.where(inside, ref, torch.zeros_like(ref)), rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("cin,cout", [(16, 32), (64, 64)])
def test_example_py_on_gpu(cin, cout):
    """BASELINE config 1 (example.py): sparse == dense within atol 1e-4 (example.py:95)."""
    from tests.test_host_logic import ExampleModel

    g = util.golden("masks")
    mask = util.unpack(g["assets_mask/mask"], g["assets_mask/shape"]).to(DEV)
    torch.manual_seed(0)
    orig = torch.randn(1, cin, 256, 256, device=DEV)
    edited = orig + torch.randn(1, cin, 256, 256, device=DEV) * mask[None, None]
    model = ExampleModel(cin, cout).to(DEV).eval()
    with torch.no_grad():
        model.set_mode("full")
        std = model(edited)
        model(orig)
        model.set_mode("sparse")
        model.set_masks({(256, 256): mask})
        sp = model(edited)
    assert model.m.gather.active_indices.shape[0] == 783
    assert np.array_equal(model.m.gather.active_indices.cpu().numpy(), util.golden("example")["c16_32/idx"])
    assert torch.isclose(std, sp, atol=1e-4).all(), (std - sp).abs().max().item()


def test_example_golden_output_on_gpu():
    """Same inputs as tests/golden/make_golden.py: the GPU sparse output equals the
    REFERENCE's sparse output (sige.nn + sige/cpu) within 1e-3."""
    from tests.test_host_logic import ExampleModel, _example_inputs

    mask, orig, edited, w, b = _example_inputs()
    model = ExampleModel(16, 32).to(DEV).eval()
    with torch.no_grad():
        model.m.conv.weight.copy_(w)
        model.m.conv.bias.copy_(b)
        model.set_mode("full")
        model(orig.to(DEV))
        model.set_mode("sparse")
        model.set_masks({(256, 256): mask.to(DEV)})
        sp = model(edited.to(DEV)).cpu()
    ex = util.golden("example")
    torch.testing.assert_close(sp[:, ::4, ::3, ::3], torch.from_numpy(ex["c16_32/sparse_sub"]), rtol=0, atol=util.CONV_ATOL)
    assert abs(sp.double().sum().item() - float(ex["c16_32/sparse_sum"])) < 1.0


def test_resblock_gpu_vs_oracle_backend():
    """Module-level parity: the same ResBlock (weights, caches, masks) run through
    the HIP backend on the GPU and through the CPU oracle backend."""
    from sige_amd import runtime
    from sige_amd.utils import dilate_mask
    from tests.test_host_logic import ResNet

    torch.manual_seed(3)
    net = ResNet(32, 64).eval()
    blk = net.block
    blk.s1, blk.t1 = torch.randn(1, 32, 1, 1), torch.randn(1, 32, 1, 1)
    blk.s2, blk.t2 = torch.randn(1, 64, 1, 1), torch.randn(1, 64, 1, 1)
    orig = torch.randn(1, 32, 64, 64)
    mask = torch.zeros(64, 64, dtype=torch.bool)
    mask[20:31, 12:40] = True
    mask[63, 0] = True
    edited = orig + torch.randn_like(orig) * mask
    masks = {(64, 64): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))}

    def run(device):
        net.to(device)
        for n in ("s1", "t1", "s2", "t2"):
            setattr(blk, n, getattr(blk, n).to(device))
        with torch.no_grad():
            net.set_mode("full")
            dense = net(edited.to(device))
            net(orig.to(device))
            net.set_mode("sparse")
            net.set_masks({k: v.to(device) for k, v in masks.items()})
            return dense.cpu(), net(edited.to(device)).cpu()

    runtime.register_backend("cpu", oracle)
    try:
        dense_cpu, sparse_cpu = run("cpu")
    finally:
        runtime.unregister_backend("cpu")
    dense_gpu, sparse_gpu = run(DEV)
    torch.testing.assert_close(sparse_gpu, sparse_cpu, rtol=0, atol=util.CONV_ATOL)
    torch.testing.assert_close(sparse_gpu, dense_gpu, rtol=0, atol=util.CONV_ATOL)


def test_hipgraph_capture_replay(hip):
    """Every entry point is capturable (no allocation / sync inside the library)."""
    from sige_amd.utils import reduce_mask

    torch.manual_seed(9)
    mask = _square_mask(0.05).to(DEV)
    idx = reduce_mask(mask, 6, 4, 1)
    x = torch.randn(1, 64, 256, 256, device=DEV)
    y = torch.randn(1, 64, 256, 256, device=DEV)
    w = torch.randn(64, 64, 3, 3, device=DEV) / 24
    packed = hip.conv_pack_weights(w, 6, 6, (1, 1))
    tbl = hip.tile_table(idx, (1, 1), (1, 1), (4, 4), (256, 256))

    def step():
        t = hip.gather(x, 6, 6, idx, None, None, "swish", False)
        c = hip.block_conv(t, packed, None, 64, (3, 3), (1, 1))
        return hip.scatter_fused(c, y, tbl, idx.shape[0], x)

    eager = step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
        with torch.cuda.graph(graph, stream=side):
            out = step()
    torch.cuda.current_stream().wait_stream(side)
    x.mul_(2.0)  # new inputs, same buffers
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, step())
    assert not torch.equal(out, eager)


def test_errors_are_loud(hip):
    x = torch.randn(1, 4, 8, 8, device=DEV)
    idx = torch.zeros(1, 2, dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError, match="Unknown activation"):
        hip.gather(x, 6, 6, idx, None, None, "relu", False)
    with pytest.raises(RuntimeError, match="invalid argument"):
        hip.gather(x, 6, 6, idx, torch.randn(1, 3, 1, 1, device=DEV), None)  # not broadcastable
    with pytest.raises(NotImplementedError):
        hip.gather(x.half(), 6, 6, idx)
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        hip.gather(x.cpu(), 6, 6, idx)
    empty = hip.gather(x, 6, 6, idx[:0])  # N = 0 is legal
    assert empty.shape == (0, 4, 6, 6)
    assert torch.equal(hip.scatter(torch.zeros(0, 4, 4, 4, device=DEV), x, 1, 1, 1, 1, idx[:0], None), x)
