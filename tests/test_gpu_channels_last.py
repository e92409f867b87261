"""GPU parity of the channels-last (NHWC) entry points against the NCHW ones (which
tests/test_gpu_parity.py pins to the oracle and the golden vectors): same values in,
same values out -- bit for bit for the data-movement ops, 1e-5 for the convs."""
import pytest
import torch

from tests import util  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
CL = torch.channels_last


@pytest.fixture(scope="module")
def hip():
    from sige_amd import hip as h

    h.lib()
    return h


def _mask(res, ratio=0.2):
    side = int(round((ratio ** 0.5) * res))
    m = torch.zeros(res, res, dtype=torch.bool)
    m[res // 3:res // 3 + side, res // 4:res // 4 + side] = True
    m[0, 0] = m[res - 1, res - 1] = True
    return m.to(DEV)


def _cl(t):
    return t.contiguous(memory_format=CL)


@pytest.mark.parametrize("B,C,res,blk,off", [(1, 128, 64, 6, 1), (2, 40, 48, 4, 0), (1, 64, 32, 5, 0)])
@pytest.mark.parametrize("act,aff", [("swish", "bc"), ("identity", "c"), ("identity", None)])
def test_gather_and_scatter_gather_cl(hip, B, C, res, blk, off, act, aff):
    from sige_amd.utils import reduce_mask

    torch.manual_seed(B + C + res)
    x, y = torch.randn(B, C, res, res, device=DEV), torch.randn(B, C, res, res, device=DEV)
    idx = reduce_mask(_mask(res), blk, 4, off)
    sc = sh = None
    if aff:
        shp = (B if aff == "bc" else 1, C, 1, 1)
        sc, sh = torch.randn(*shp, device=DEV), torch.randn(*shp, device=DEV)
    want = hip.gather(x, blk, blk, idx, sc, sh, act, False)
    got = hip.gather_cl(_cl(x), blk, blk, idx, sc, sh, act)
    assert hip.is_cl(got)
    assert torch.equal(got.contiguous(), want)
    if blk == 6:
        smap = hip.get_scatter_map(res, res, 6, 6, 3, 3, 1, 1, 1, 1, idx)
        t4 = torch.randn(B * idx.shape[0], C, 4, 4, device=DEV)
        want = hip.scatter_gather(t4, y, 6, 6, idx, smap, sc, sh, act, False)
        got = hip.scatter_gather_cl(_cl(t4), _cl(y), 6, 6, idx, smap, sc, sh, act)
        assert torch.equal(got.contiguous(), want)


@pytest.mark.parametrize("B,C,res", [(1, 128, 64), (2, 36, 52)])
def test_scatter_cl_full_and_in_place(hip, B, C, res):
    from sige_amd.utils import reduce_mask

    torch.manual_seed(C)
    mask = _mask(res)
    idx0, idx1 = reduce_mask(mask, 6, 4, 1), reduce_mask(mask, 4, 4, 0)
    n0, n1 = idx0.shape[0], idx1.shape[0]
    x0, x1 = torch.randn(B * n0, C, 4, 4, device=DEV), torch.randn(B * n1, C, 4, 4, device=DEV)
    y0, y1, res_t = (torch.randn(B, C, res, res, device=DEV) for _ in range(3))
    t0 = hip.tile_table(idx0, (1, 1), (1, 1), (4, 4), (res, res))
    t1 = hip.tile_table(idx1, (0, 0), (1, 1), (4, 4), (res, res))
    for residual in (None, res_t):
        want = hip.scatter_fused(x0, y0, t0, n0, residual)
        got = hip.scatter_cl(_cl(x0), _cl(y0), (1, 1), (1, 1), idx0, t0, None if residual is None else _cl(residual))
        assert hip.is_cl(got) and torch.equal(got.contiguous(), want)
        buf = _cl(y0).clone(memory_format=torch.preserve_format)
        out = hip.scatter_cl(_cl(x0), _cl(y0), (1, 1), (1, 1), idx0, t0, None if residual is None else _cl(residual), out=buf)
        assert out.data_ptr() == buf.data_ptr() and torch.equal(out.contiguous(), want)
    want = hip.scatter_with_block_residual_fused(x0, y0, x1, y1, t0, n0, t1, n1)
    got = hip.scatter_with_block_residual_cl(_cl(x0), _cl(y0), _cl(x1), _cl(y1), (1, 1), (1, 1), idx0, t0, idx1, t1)
    assert torch.equal(got.contiguous(), want)
    buf = _cl(y0).clone(memory_format=torch.preserve_format)
    out = hip.scatter_with_block_residual_cl(_cl(x0), _cl(y0), _cl(x1), _cl(y1), (1, 1), (1, 1), idx0, t0, idx1, t1, out=buf)
    assert torch.equal(out.contiguous(), want)
    # a second in-place call with other tiles on the same mask overwrites exactly the covered pixels
    x0b = torch.randn_like(x0)
    want = hip.scatter_with_block_residual_fused(x0b, y0, x1, y1, t0, n0, t1, n1)
    out = hip.scatter_with_block_residual_cl(_cl(x0b), _cl(y0), _cl(x1), _cl(y1), (1, 1), (1, 1), idx0, t0, idx1, t1, out=buf)
    assert torch.equal(out.contiguous(), want)


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("mt,nb", [(0, 0), (16, 1), (16, 2), (32, 1), (32, 2)])
@pytest.mark.parametrize("cin,c2,cout,k,stride,blk,off", [(128, 0, 128, 3, 1, 6, 1), (64, 40, 200, 3, 1, 6, 1), (96, 0, 64, 1, 1, 4, 0),
                                                          (256, 44, 40, 1, 1, 4, 0), (64, 0, 72, 3, 2, 5, 0), (68, 0, 24, 3, 1, 6, 1),
                                                          (128, 64, 64, 3, 1, 6, 1), (512, 256, 48, 1, 1, 4, 0)])
def test_conv_cl_vs_nchw(hip, waves, mt, nb, cin, c2, cout, k, stride, blk, off, tuning):
    from sige_amd.utils import reduce_mask

    torch.manual_seed(cin + cout + mt + nb)
    B, res = 2, 48
    C = cin + c2
    x, y = torch.randn(B, C, res, res, device=DEV), torch.randn(B, C, res, res, device=DEV)
    w = torch.randn(cout, C, k, k, device=DEV) / (k * C ** 0.5)
    bias = torch.randn(cout, device=DEV)
    scale, shift = torch.randn(1, C, 1, 1, device=DEV), torch.randn(1, C, 1, 1, device=DEV)
    idx = reduce_mask(_mask(res), blk, 4, off)
    packed = hip.conv_pack_weights(w, blk, blk, (stride, stride))
    conv = lambda t: torch.nn.functional.conv2d(t.double(), w.double(), bias.double(), stride).float()  # noqa: E731
    hip.conv_force_tile(mt, nb)
    hip.conv_force_waves(waves)
    try:
        for act, sc, sh in (("swish", scale, shift), ("identity", scale, shift), ("identity", None, None)):
            tiles = hip.gather(x, blk, blk, idx, sc, sh, act, False)
            want = conv(tiles)
            got = hip.block_conv_cl(_cl(tiles), packed, bias, cout, (k, k), (stride, stride))
            assert hip.is_cl(got)
            torch.testing.assert_close(got.contiguous(), want, rtol=0, atol=1e-4)
            got = hip.gather_conv_cl(_cl(x), None, (blk, blk), idx, sc, sh, act, packed, bias, cout, (k, k), (stride, stride))
            torch.testing.assert_close(got.contiguous(), want, rtol=0, atol=1e-4)
            if c2:  # channels from two tensors, straight into a full tensor (per image)
                ho = res if stride == 1 else res // 2
                o = 4 if stride == 1 else 2
                t_out = want.reshape(B, idx.shape[0], cout, o, o)
                for b in range(B):
                    residual = torch.randn(1, cout, ho, ho, device=DEV)
                    out = hip.gather_conv_cl(_cl(x[b:b + 1, :cin]), _cl(x[b:b + 1, cin:]), (blk, blk), idx, sc, sh, act, packed, bias,
                                             cout, (k, k), (stride, stride),
                                             full=dict(offset=(off, off), out_res=(ho, ho), residual=_cl(residual)))
                    assert out is not None and hip.is_cl(out)
                    for n in (0, idx.shape[0] // 2, idx.shape[0] - 1):
                        h0, w0 = (int(idx[n, 0]) + off) // stride, (int(idx[n, 1]) + off) // stride
                        h1, w1 = min(h0 + o, ho), min(w0 + o, ho)
                        exp = t_out[b, n][:, :h1 - h0, :w1 - w0] + residual[0, :, h0:h1, w0:w1]
                        torch.testing.assert_close(out[0, :, h0:h1, w0:w1], exp, rtol=0, atol=1e-4)
            if k == 3 and stride == 1:
                smap = hip.get_scatter_map(res, res, 6, 6, 3, 3, 1, 1, 1, 1, idx)
                t4 = torch.randn(B * idx.shape[0], C, 4, 4, device=DEV)
                sg = hip.scatter_gather(t4, y, 6, 6, idx, smap, sc, sh, act, False)
                got = hip.scatter_gather_conv_cl(_cl(t4), _cl(y), (6, 6), idx, smap, sc, sh, act, packed, bias, cout, (3, 3), (1, 1))
                torch.testing.assert_close(got.contiguous(), conv(sg), rtol=0, atol=1e-4)
    finally:
        hip.conv_force_tile(0, 0)
        hip.conv_force_waves(0)


@pytest.mark.oracle_parity  # (the reference computes this row with the torch op compared here)
@pytest.mark.parametrize("shape,groups", [((1, 128, 256, 256), 32), ((2, 64, 17, 23), 16), ((1, 512, 8, 8), 32)])
def test_group_norm_affine_cl(hip, shape, groups):
    torch.manual_seed(shape[1])
    x = torch.randn(*shape, device=DEV) * 3 + 1.5
    gamma, beta = torch.randn(shape[1], device=DEV), torch.randn(shape[1], device=DEV)
    r = hip.group_norm_affine_cl(_cl(x), groups, 1e-6, gamma, beta)
    assert r is not None
    scale, shift = r
    want = torch.nn.functional.group_norm(x.double(), groups, gamma.double(), beta.double(), 1e-6).float()
    torch.testing.assert_close(x * scale + shift, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,C,hw", [(1, 512, 16), (1, 512, 8), (2, 64, 12)])
def test_attention_cl(hip, B, C, hw):
    torch.manual_seed(C + hw)
    qkv = torch.randn(B, 3 * C, hw, hw, device=DEV)
    got = hip.attention_cl(_cl(qkv), C ** -0.5)
    assert got is not None and hip.is_cl(got)
    q, k, v = qkv.double().reshape(B, 3, C, hw * hw).unbind(1)
    attn = torch.softmax(torch.bmm(q.transpose(1, 2), k) * (C ** -0.5), dim=2)
    want = torch.bmm(v, attn.transpose(1, 2)).reshape(B, C, hw, hw).float()
    torch.testing.assert_close(got.contiguous(), want, rtol=0, atol=2e-5)


@pytest.mark.parametrize("inplace", [False, True])
def test_ddpm_unet_channels_last_equals_nchw(inplace):
    """Model level: the DDPM-256 workload run channels-last (and with the persistent in-place
    scatter buffers) gives the NCHW run's sparse output, for two different edits on one mask."""
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig(ch=32)).to(DEV).eval()
    x0 = torch.randn(1, 3, 256, 256, device=DEV)
    mask = torch.zeros(256, 256, dtype=torch.bool, device=DEV)
    mask[100:140, 90:150] = True
    edits = [x0 + torch.randn(1, 3, 256, 256, device=DEV) * mask for _ in range(2)]
    t = torch.zeros(1, device=DEV)
    masks = downsample_mask(dilate_mask(mask, 5), 8)
    outs = {}
    with torch.no_grad():
        for layout in ("nchw", "nhwc"):
            fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
            model.to(memory_format=fmt)
            model.clear_cache()
            model.set_scatter_inplace(inplace and layout == "nhwc")
            model.set_mode("full")
            model(x0.contiguous(memory_format=fmt), t)
            model.set_masks(masks)
            model.set_mode("sparse")
            # (the first sparse forward after a full pass registers the activated twins of the conv1 inputs and activates for
            #  itself; from the second on the producers' twins are read: equal to ~1e-5, not to the bit)
            model(edits[1].contiguous(memory_format=fmt), t)
            outs[layout] = [model(e.contiguous(memory_format=fmt), t).contiguous().clone() for e in edits]
            if layout == "nhwc":  # again, first edit: the in-place buffers must not remember the second one
                again = model(edits[0].contiguous(memory_format=fmt), t).contiguous()
                torch.testing.assert_close(again, outs[layout][0], rtol=0, atol=0)
        model.to(memory_format=torch.contiguous_format)
    # (the NCHW run itself is pinned to the oracle backend by tests/test_gpu_parity.py)
    for a, b in zip(outs["nchw"], outs["nhwc"]):
        torch.testing.assert_close(a, b, rtol=0, atol=1e-4)
    assert (outs["nhwc"][0] - outs["nhwc"][1]).abs().max() > 1e-3  # the two edits really differ


@pytest.mark.parametrize("B,C,H,W,cout,act", [(1, 128, 256, 256, 3, "swish"), (2, 36, 19, 45, 4, "identity"), (1, 64, 8, 8, 1, "swish"),
                                              (2, 128, 37, 21, 3, "swish"), (1, 64, 33, 16, 2, "identity"), (3, 128, 16, 50, 1, "swish")])
def test_conv3x3_small_cout_cl(hip, B, C, H, W, cout, act, tuning):
    torch.manual_seed(C + H)
    x = torch.randn(B, C, H, W, device=DEV)
    w = torch.randn(cout, C, 3, 3, device=DEV) / (3 * C ** 0.5)
    bias = torch.randn(cout, device=DEV)
    sc, sh = torch.randn(B, C, 1, 1, device=DEV), torch.randn(B, C, 1, 1, device=DEV)
    got = hip.conv3x3_small_cout_cl(_cl(x), w, bias, sc, sh, act)
    assert got is not None
    h = x.double() * sc.double() + sh.double()
    if act == "swish":
        h = torch.nn.functional.silu(h)
    want = torch.nn.functional.conv2d(h, w.double(), bias.double(), 1, 1).float()
    torch.testing.assert_close(got.contiguous(), want, rtol=0, atol=1e-4)
    plain = hip.conv3x3_small_cout_cl(_cl(x), w, None)
    torch.testing.assert_close(plain.contiguous(), torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1).float(), rtol=0, atol=1e-4)
    # the tap-GEMM (MFMA) kernel and the scalar-weight kernel are two evaluations of the same sums
    hip.conv3x3_small_cout_force_scalar(True)
    try:
        scalar = hip.conv3x3_small_cout_cl(_cl(x), w, bias, sc, sh, act)
    finally:
        hip.conv3x3_small_cout_force_scalar(False)
    torch.testing.assert_close(scalar.contiguous(), want, rtol=0, atol=1e-4)
    torch.testing.assert_close(scalar, got, rtol=0, atol=2e-5)


@pytest.mark.parametrize("B,C,H,W,cout", [(1, 3, 256, 256, 128), (2, 3, 19, 45, 64), (1, 1, 8, 8, 32), (2, 2, 33, 16, 128), (1, 3, 5, 7, 128)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_conv3x3_small_cin_cl(hip, B, C, H, W, cout, layout):
    """The U-Net's first layer as one thin-GEMM launch: any input layout in, channels-last out."""
    torch.manual_seed(C + H + cout)
    x = torch.randn(B, C, H, W, device=DEV)
    w = torch.randn(cout, C, 3, 3, device=DEV) / (3 * C ** 0.5)
    bias = torch.randn(cout, device=DEV)
    xin = _cl(x) if layout == "nhwc" else x
    got = hip.conv3x3_small_cin_cl(xin, w, bias)
    assert got is not None and got.is_contiguous(memory_format=torch.channels_last)
    want = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), 1, 1).float()
    torch.testing.assert_close(got.contiguous(), want, rtol=0, atol=1e-5)
    nobias = hip.conv3x3_small_cin_cl(xin, w, None)
    torch.testing.assert_close(nobias.contiguous(), torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1).float(), rtol=0, atol=1e-5)
    assert hip.conv3x3_small_cin_cl(xin, torch.randn(48, C, 3, 3, device=DEV), None) is None  # unsupported width -> caller's plain conv


def test_input_conv2d_matches_the_plain_conv(hip):
    from sige_amd.nn.dense import input_conv2d

    torch.manual_seed(3)
    conv = torch.nn.Conv2d(3, 128, 3, 1, 1).to(DEV).to(memory_format=torch.channels_last)
    x = torch.randn(1, 3, 64, 48, device=DEV)
    with torch.no_grad():
        want = conv(x)
        got = input_conv2d(conv, _cl(x))
        assert got.is_contiguous(memory_format=torch.channels_last)
        torch.testing.assert_close(got, want, rtol=0, atol=1e-5)
        torch.testing.assert_close(input_conv2d(conv, x), want, rtol=0, atol=0)  # NCHW input: the plain conv itself


@pytest.mark.parametrize("c1,c2,k", [(128, 64, 3), (256, 128, 1), (100, 28, 3)])
def test_lazy_cat_feeds_fused_gather(hip, c1, c2, k):
    """A deferred torch.cat consumed by Gather -> SIGEConv2d (two-pointer kernel input, or the
    materialising fallback when the split is not on a chunk boundary) equals the eager cat."""
    from sige_amd.nn import Gather, SIGEConv2d, deferred
    from sige_amd.utils import dilate_mask

    torch.manual_seed(c1 + c2)
    res = 64
    conv = SIGEConv2d(c1 + c2, 96, k, 1, k // 2).to(DEV).eval()
    gather = Gather(conv, 6 if k == 3 else 4, activation_name="swish" if k == 3 else "identity").to(DEV)
    a, b = _cl(torch.randn(1, c1, res, res, device=DEV)), _cl(torch.randn(1, c2, res, res, device=DEV))
    scale = torch.randn(1, c1 + c2, 1, 1, device=DEV) if k == 3 else None
    shift = torch.randn(1, c1 + c2, 1, 1, device=DEV) if k == 3 else None
    mask = torch.zeros(res, res, dtype=torch.bool, device=DEV)
    mask[20:33, 10:40] = True
    with torch.no_grad():
        for m in (gather, conv):
            m.set_mode("full")
        conv(gather(torch.cat([a, b], 1)))
        gather.set_mask({(res, res): dilate_mask(mask, 2)}, {}, 1)
        for m in (gather, conv):
            m.set_mode("sparse")
        lazy = deferred.lazy_cat(a, b)
        assert isinstance(lazy, deferred.LazyCat)
        got = conv(gather(lazy, scale, shift))
        want = conv(gather(torch.cat([a, b], 1), scale, shift))
    torch.testing.assert_close(got.contiguous(), want.contiguous(), rtol=0, atol=1e-5)
    assert lazy.spec is not None or (c1 % 32) != 0  # the cat never ran when the split sits on a chunk boundary


@pytest.mark.parametrize("res,c1,c2,cout,k,stride", [(8, 512, 0, 512, 3, 1), (8, 512, 512, 512, 3, 1), (8, 512, 256, 512, 1, 1),
                                                     (16, 512, 0, 512, 3, 1), (32, 256, 256, 256, 3, 1), (16, 256, 0, 256, 3, 2),
                                                     (8, 64, 0, 48, 3, 1)])
def test_dense_fused_conv_cl(hip, res, c1, c2, cout, k, stride):
    """Dense layers, channels-last: conv(swish(cat(x,x2)*s+t)) + residual == torch, including the
    8x8 layers that take the cross-workgroup K split (workspace + deterministic second pass)."""
    from torch import nn

    from sige_amd.nn.dense import fused_conv2d

    torch.manual_seed(res + c1 + k)
    conv = nn.Conv2d(c1 + c2, cout, k, stride, 0 if stride == 2 else k // 2).to(DEV)
    x = _cl(torch.randn(1, c1, res, res, device=DEV))
    x2 = _cl(torch.randn(1, c2, res, res, device=DEV)) if c2 else None
    s, t = torch.randn(1, c1 + c2, 1, 1, device=DEV), torch.randn(1, c1 + c2, 1, 1, device=DEV)
    ro = res if stride == 1 else res // 2
    residual = _cl(torch.randn(1, cout, ro, ro, device=DEV))
    with torch.no_grad():
        got = fused_conv2d(conv, x, s, t, "swish", x2=x2, residual=residual, pad_bottom_right=stride == 2)
        assert hip.is_cl(got) or got.shape[2] == 1
        h = x if x2 is None else torch.cat([x, x2], 1)
        h = torch.nn.functional.silu(h * s + t)
        if stride == 2:
            h = torch.nn.functional.pad(h, (0, 1, 0, 1))
        want = torch.nn.functional.conv2d(h.double(), conv.weight.double(), conv.bias.double(), stride,
                                          conv.padding).float() + residual
        torch.testing.assert_close(got.contiguous(), want.contiguous(), rtol=0, atol=2e-4)
        again = fused_conv2d(conv, x, s, t, "swish", x2=x2, residual=residual, pad_bottom_right=stride == 2)
        assert torch.equal(again, got)  # the K split is deterministic


@pytest.mark.parametrize("res,cin,cout", [(16, 128, 96), (8, 512, 256)])
def test_epilogue_affine_activation(hip, res, cin, cout):
    """out_affine: the consumer's affine + SiLU in the producer's epilogue (also through the K split)."""
    from torch import nn

    from sige_amd.nn.dense import fused_conv2d

    torch.manual_seed(res + cin)
    conv = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    x = _cl(torch.randn(1, cin, res, res, device=DEV))
    residual = _cl(torch.randn(1, cout, res, res, device=DEV))
    os_, oh_ = torch.randn(1, cout, 1, 1, device=DEV), torch.randn(1, cout, 1, 1, device=DEV)
    with torch.no_grad():
        got = fused_conv2d(conv, x, residual=residual, out_affine=(os_, oh_, "swish"))
        want = torch.nn.functional.silu((conv(x.contiguous()) + residual) * os_ + oh_)
    torch.testing.assert_close(got.contiguous(), want.contiguous(), rtol=0, atol=2e-4)


def test_gather_conv_cl_fused_upsample(hip):
    """upsample2x: gathering (h/2, w/2) of the half-resolution tensor == gathering its nearest x2 upsampling."""
    from sige_amd.utils import reduce_mask

    torch.manual_seed(5)
    C, cout, lo = 64, 48, 24
    x = torch.randn(1, C, lo, lo, device=DEV)
    up = torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
    w = torch.randn(cout, C, 3, 3, device=DEV) / (3 * C ** 0.5)
    bias = torch.randn(cout, device=DEV)
    idx = reduce_mask(_mask(2 * lo), 6, 4, 1)
    packed = hip.conv_pack_weights(w, 6, 6, (1, 1))
    want = hip.gather_conv_cl(_cl(up), None, (6, 6), idx, None, None, "identity", packed, bias, cout, (3, 3), (1, 1))
    got = hip.gather_conv_cl(_cl(x), None, (6, 6), idx, None, None, "identity", packed, bias, cout, (3, 3), (1, 1), upsample2x=True)
    assert torch.equal(got, want)


def test_conv_scatter_fusion_in_modules(hip):
    """Scatter.forward_fused / ScatterWithBlockResidual.forward_fused (the conv's epilogue writes the persistent
    output) equal the two-module form, for a ResBlock with and without a shortcut conv."""
    from sige_amd.nn import Gather, Scatter, ScatterGather, ScatterWithBlockResidual, SIGEConv2d, SIGEModel, SIGEModule
    from sige_amd.utils import dilate_mask

    class Block(SIGEModule):
        def __init__(self, cin, cout):
            super().__init__()
            self.conv1, self.conv2 = SIGEConv2d(cin, cout, 3, 1, 1), SIGEConv2d(cout, cout, 3, 1, 1)
            self.main_gather = Gather(self.conv1, 6, activation_name="swish")
            self.scatter_gather = ScatterGather(self.main_gather, activation_name="swish")
            self.nin = None
            if cin != cout:
                self.nin = SIGEConv2d(cin, cout, 1, 1, 0)
                self.shortcut_gather = Gather(self.nin, 4)
                self.scatter = ScatterWithBlockResidual(self.main_gather, self.shortcut_gather)
            else:
                self.scatter = Scatter(self.main_gather)
            self.fuse = False

        def forward(self, x):
            sc = x if self.nin is None else self.nin(self.shortcut_gather(x))
            if self.mode == "full":
                h = torch.nn.functional.silu(self.main_gather(x) * self.s1 + self.t1)
                h = torch.nn.functional.silu(self.scatter_gather(self.conv1(h)) * self.s2 + self.t2)
                return self.scatter(self.conv2(h), sc)
            h = self.conv1(self.main_gather(x, self.s1, self.t1))
            tiles = self.scatter_gather(h, self.s2, self.t2)
            if self.fuse:
                return self.scatter.forward_fused(self.conv2, tiles, sc)
            return self.scatter(self.conv2(tiles), sc)

    class Net(SIGEModel):
        def __init__(self, cin, cout):
            super().__init__()
            self.block = Block(cin, cout)

        def forward(self, x):
            return self.block(x)

    torch.manual_seed(11)
    for cin, cout in ((64, 64), (64, 128)):
        net = Net(cin, cout).to(DEV).eval().to(memory_format=CL)
        blk = net.block
        blk.s1, blk.t1 = torch.randn(1, cin, 1, 1, device=DEV), torch.randn(1, cin, 1, 1, device=DEV)
        blk.s2, blk.t2 = torch.randn(1, cout, 1, 1, device=DEV), torch.randn(1, cout, 1, 1, device=DEV)
        orig = _cl(torch.randn(1, cin, 64, 64, device=DEV))
        mask = torch.zeros(64, 64, dtype=torch.bool, device=DEV)
        mask[20:31, 12:40] = True
        mask[0, 0] = True
        edits = [_cl(orig + torch.randn_like(orig) * mask) for _ in range(2)]
        with torch.no_grad():
            net.set_mode("full")
            net(orig)
            net.set_mode("sparse")
            net.set_masks({(64, 64): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))})
            want = [net(e).contiguous().clone() for e in edits]
            net.set_scatter_inplace(True)
            blk.fuse = True
            got = [net(e).contiguous().clone() for e in edits]
            again = net(edits[0]).contiguous()
        for g, w in zip(got, want):
            torch.testing.assert_close(g, w, rtol=0, atol=1e-5)
        torch.testing.assert_close(again, want[0], rtol=0, atol=1e-5)


def test_empty_mask_sparse_forward_is_the_cached_result():
    """N = 0 active tiles (the reference demo short-circuits this case, diffusion_demo/runner.py:146-147): every
    fused / in-place kernel must be a no-op and the sparse forward must return the original image's output."""
    from sige_amd.utils import downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig(ch=32)).to(DEV).eval().to(memory_format=CL)
    model.set_scatter_inplace(True)
    x0 = _cl(torch.randn(1, 3, 256, 256, device=DEV))
    t = torch.zeros(1, device=DEV)
    with torch.no_grad():
        model.set_mode("full")
        full = model(x0, t)
        model.set_masks(downsample_mask(torch.zeros(256, 256, dtype=torch.bool, device=DEV), 8))
        model.set_mode("sparse")
        sparse = model(x0, t)
    # (the dense remainder and the output norm are recomputed, so equality is to rounding, not bit-wise)
    torch.testing.assert_close(sparse.contiguous(), full.contiguous(), rtol=0, atol=2e-4)
