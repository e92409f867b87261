"""GPU tests added in round 2: device-side mask pipeline against the reference's golden masks (SURVEY.md 8 rows a2 /
f4), the benchmarked artefact at its own size against the oracle, in-place scatter buffers across cache changes."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from oracle import oracle  # noqa: E402
from tests import util  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from sige_amd import hip as h

    h.lib()
    return h


def _fixture_names():
    return sorted({k.split("/")[0] for k in util.golden("masks").files})


# ---- a2 / f4: the mask helpers on DEVICE masks, bit-exact against the reference's outputs ----------------------
@pytest.mark.parametrize("name", _fixture_names())
def test_device_mask_helpers_bit_exact(hip, name):
    """dilate_mask / downsample_mask / reduce_mask of a mask that lives on the GPU (HIP kernels: one launch for the
    whole pyramid) against the goldens the real reference produced (tests/golden/make_golden.py)."""
    from sige_amd.utils import dilate_mask, downsample_mask, reduce_mask

    g = util.golden("masks")
    shape = g[name + "/shape"]
    mask = util.unpack(g[name + "/mask"], shape).to(DEV)
    for dil in (1, 2, 5):
        got = dilate_mask(mask, dil)
        assert got.is_cuda and got.dtype == torch.bool
        assert torch.equal(got.cpu(), util.unpack(g["%s/dilate/%d" % (name, dil)], shape))
    assert torch.equal(dilate_mask(mask, (2, 0)).cpu(), oracle.dilate_mask(mask.cpu(), (2, 0)))  # rectangular dilation
    for min_res, dil in ((8, 1), (8, 2), (4, 1)):
        pyr = downsample_mask(mask, min_res=min_res, dilation=dil)
        assert len(pyr) == len([k for k in g.files if k.startswith("%s/pyramid/%d_%d/" % (name, min_res, dil))])
        for (h, w), pm in pyr.items():
            assert pm.is_cuda and pm.dtype == torch.bool and tuple(pm.shape) == (h, w)
            assert torch.equal(pm.cpu(), util.unpack(g["%s/pyramid/%d_%d/%dx%d" % (name, min_res, dil, h, w)], (h, w))), (h, w)
    # the diffusion runner's recipe (diffusion/runner.py:157-165): dilate 5, pyramid down to 8, then the index lists
    pyr = downsample_mask(dilate_mask(mask, 5), min_res=8)
    for (h, w), pm in pyr.items():
        assert torch.equal(pm.cpu(), util.unpack(g["%s/ddpm/%dx%d" % (name, h, w)], (h, w)))
        key = "%s/ddpm_reduce_b6/%dx%d" % (name, h, w)
        if key in g.files:
            assert torch.equal(reduce_mask(pm, 6, 4, 1).cpu(), torch.from_numpy(g[key]))


def test_device_difference_mask(hip):
    from sige_amd.utils import compute_difference_mask

    torch.manual_seed(0)
    a = torch.randn(1, 3, 96, 80)
    b = a.clone()
    b[0, :, 20:40, 10:33] += torch.randn(3, 20, 23) * 0.05
    b[0, 1, 70, 70] += 0.0201
    b[0, 2, 71, 71] += 0.0199
    want = torch.any((torch.abs(a - b) > 2e-2)[0], 0)
    for fmt in (torch.contiguous_format, torch.channels_last):
        got = compute_difference_mask(a.to(DEV).contiguous(memory_format=fmt), b.to(DEV).contiguous(memory_format=fmt))
        assert got.is_cuda and got.dtype == torch.bool and torch.equal(got.cpu(), want)
    assert torch.equal(compute_difference_mask(a[0].to(DEV), b[0].to(DEV)).cpu(), want)
    assert torch.equal(compute_difference_mask(a[0, 1].to(DEV), b[0, 1].to(DEV), eps=0.03).cpu(), (a[0, 1] - b[0, 1]).abs() > 0.03)


def test_set_masks_builds_every_index_list_with_one_sync(hip):
    """SIGEModel.set_masks on GPU masks: the batched compaction gives the same lists as one reduce_mask per geometry."""
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig(ch=32)).to(DEV).eval()
    mask = torch.zeros(256, 256, dtype=torch.bool)
    mask[100:128, 90:118] = True
    with torch.no_grad():
        model.set_mode("full")
        model(torch.randn(1, 3, 256, 256, device=DEV), torch.zeros(1, device=DEV))
    masks = downsample_mask(dilate_mask(mask.to(DEV), 5), 8)
    model.set_masks(masks)
    from sige_amd.nn import Gather

    cpu_masks = oracle.downsample_mask(oracle.dilate_mask(mask, 5), 8)
    seen = 0
    for m in model.modules():
        if isinstance(m, Gather):
            want = oracle.reduce_mask(cpu_masks[tuple(m.input_res)], m.block_size, m.block_stride, m.offset)
            assert torch.equal(m.active_indices.cpu(), want)
            seen += 1
    assert seen > 30


# ---- the benchmarked artefact at ITS size: ch 128, channels-last, in-place scatter buffers, hipGraph replay --------
@pytest.mark.parametrize("ratio", [0.012, 0.15])
def test_benchmarked_forward_vs_oracle_at_full_size(hip, ratio):
    """bench.py's headline configuration (BASELINE.json configs[1]) -- DDPM-256 ch 128, torch.channels_last, persistent
    in-place scatter outputs, conv -> scatter epilogue fusion, producer-side activation, cross-workgroup K split,
    hipGraph replay -- against the same network on the CPU with the oracle as native backend, same weights / inputs /
    masks; north_star tolerance 1e-3."""
    import bench
    from sige_amd import runtime
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval()
    x0, noise = bench.make_inputs()
    mask = bench.edit_mask(ratio)
    x1 = x0 + noise * mask
    t = torch.zeros(1)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    oracle.set_num_threads(min(32, os.cpu_count() or 1))
    runtime.register_backend("cpu", oracle)
    try:
        with torch.no_grad():
            model.set_mode("full")
            full_c = model(x0, t)
            model.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
            model.set_mode("sparse")
            sparse_c = model(x1, t)
    finally:
        runtime.unregister_backend("cpu")
    model.clear_cache()
    model = model.to(DEV).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    cl = lambda a: a.to(DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    with torch.no_grad():
        model.set_mode("full")
        full_g = model(cl(x0), t.to(DEV))
        model.set_masks(downsample_mask(dilate_mask(mask.to(DEV), 5), 8))
        model.set_mode("sparse")
        model(cl(x1), t.to(DEV))  # (registers the activated twins of the conv1 inputs; from the next forward on they are read)
        eager = model(cl(x1), t.to(DEV)).clone()
        g, out = bench.capture(model, cl(x1), t.to(DEV))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
    torch.testing.assert_close(full_g.cpu(), full_c, rtol=0, atol=util.CONV_ATOL)
    torch.testing.assert_close(out.cpu(), sparse_c, rtol=0, atol=util.CONV_ATOL)
    assert torch.equal(out, eager)  # replaying the graph == the eager forward, bit for bit
    assert (sparse_c - full_c).abs().max() > 1e-2


# ---- in-place scatter buffers survive cache changes (stale buffer after full(A) .. full(C)) -------------
def test_inplace_scatter_buffer_follows_the_cache(hip):
    from sige_amd.utils import dilate_mask
    from tests.test_host_logic import ResNet

    torch.manual_seed(2)
    net = ResNet(64, 128).to(DEV).to(memory_format=torch.channels_last).eval()
    blk = net.block
    blk.s1, blk.t1 = torch.randn(1, 64, 1, 1, device=DEV), torch.randn(1, 64, 1, 1, device=DEV)
    blk.s2, blk.t2 = torch.randn(1, 128, 1, 1, device=DEV), torch.randn(1, 128, 1, 1, device=DEV)
    mask = torch.zeros(64, 64, dtype=torch.bool, device=DEV)
    mask[20:31, 12:40] = True
    masks = {(64, 64): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))}
    cl = lambda a: a.contiguous(memory_format=torch.channels_last)  # noqa: E731
    imgs = [cl(torch.randn(1, 64, 64, 64, device=DEV)) for _ in range(3)]
    edit = lambda x: cl(x + torch.randn_like(x) * mask)  # noqa: E731

    def sparse(inplace, edited):
        net.set_scatter_inplace(inplace)
        net.set_mode("sparse")
        return net(edited).clone()

    def diff(a, b):
        return float((a - b).abs().max())

    with torch.no_grad():
        net.set_mode("full")
        net(imgs[0])
        net.set_masks(masks)  # ONCE: the masks (and their timestamp) never change below, only the caches do
        for k in range(6):  # full(A) -> sparse -> full(B) -> sparse ...: cache addresses get recycled by the allocator
            orig = imgs[k % 3]
            net.set_mode("full")
            net(orig)
            e = edit(orig)
            want = sparse(False, e)
            got = sparse(True, e)
            assert torch.equal(got, want), (k, diff(got, want))
            assert torch.equal(sparse(True, e), want), k  # and again on the now existing buffer
        # in-place + the unfused module chain (Gather and conv as two kernels)
        from sige_amd.nn import deferred

        deferred.FUSION = False
        try:
            got = sparse(True, e)
            want2 = sparse(False, e)
        finally:
            deferred.FUSION = True
        assert torch.equal(got, want2), diff(got, want2)
        torch.testing.assert_close(want2, want, rtol=0, atol=1e-5)


def test_sparse_update_with_inplace_buffers(hip):
    """sparse_update refreshes the cache in place (sige/nn/scatter.py:59-60); the persistent output must follow."""
    from sige_amd.utils import dilate_mask
    from tests.test_host_logic import ResNet

    torch.manual_seed(3)
    net = ResNet(32, 32).to(DEV).to(memory_format=torch.channels_last).eval()
    blk = net.block
    blk.s1, blk.t1, blk.s2, blk.t2 = (torch.randn(1, 32, 1, 1, device=DEV) for _ in range(4))
    mask = torch.zeros(32, 32, dtype=torch.bool, device=DEV)
    mask[9:15, 5:19] = True
    masks = {(32, 32): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))}
    cl = lambda a: a.contiguous(memory_format=torch.channels_last)  # noqa: E731
    orig = cl(torch.randn(1, 32, 32, 32, device=DEV))
    e1 = cl(orig + torch.randn_like(orig) * mask)
    e2 = cl(e1 + torch.randn_like(orig) * mask)
    outs = {}
    with torch.no_grad():
        for inplace in (False, True):
            net.clear_cache()
            net.set_scatter_inplace(inplace)
            net.set_sparse_update(False)
            net.set_mode("full")
            net(orig)
            net.set_mode("sparse")
            net.set_masks(masks)
            net.set_sparse_update(True)
            a = net(e1).clone()       # refreshes the caches with e1's activations
            net.set_sparse_update(False)
            b = net(e2).clone()       # must build on the refreshed cache
            outs[inplace] = (a, b)
        net.set_sparse_update(False)
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])


def test_dilated_tile_conv(hip):
    """The direct kernel handles dilation (SIGEConv2d no longer has a torch conv behind it)."""
    torch.manual_seed(7)
    x = torch.randn(5, 12, 9, 9, device=DEV)
    w = torch.randn(8, 12, 3, 3, device=DEV)
    b = torch.randn(8, device=DEV)
    got = hip.block_conv_direct(x, w, b, (1, 1), 1, (2, 2))
    want = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 0, 2).float()
    torch.testing.assert_close(got, want, rtol=0, atol=1e-4)


def test_launch_counter(hip):
    x = torch.randn(1, 8, 16, 16, device=DEV)
    idx = torch.tensor([[0, 0], [4, 4]], dtype=torch.int32, device=DEV)
    n0 = hip.launch_count()
    hip.gather(x, 6, 6, idx)
    assert hip.launch_count() == n0 + 1


# ---- f16 compute (BASELINE.json configs[4]): fp16 operands on the fp16 matrix cores, fp32 accumulation -------------
from sige_amd.tolerance import F16_ATOL, F16_RTOL  # noqa: E402  (the ONE stated criterion of the f16 path: sige_amd/tolerance.py)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("k,cin,cout,T,mt,nb", [(3, 128, 128, 124, 0, 0), (3, 256, 128, 40, 16, 1), (3, 64, 256, 7, 32, 1),
                                                 (3, 192, 64, 33, 16, 2), (3, 128, 128, 70, 32, 2), (1, 256, 128, 56, 0, 0),
                                                 (1, 128, 256, 9, 32, 1), (1, 384, 128, 30, 16, 2), (3, 36, 64, 5, 0, 0)])
def test_f16_compute_block_conv_exact_products(hip, k, cin, cout, T, mt, nb, tuning):
    """The f16-compute tile conv equals an fp64 conv of the fp16-ROUNDED operands to fp32 summation accuracy (products
    of two fp16 values are exact in fp32), and the fp32 oracle within the stated f16 tolerance."""
    torch.manual_seed(T + cin)
    R = 6 if k == 3 else 4
    x = _cl(torch.randn(T, cin, R, R, device=DEV))
    w = torch.randn(cout, cin, k, k, device=DEV) / (k * cin ** 0.5)
    b = torch.randn(cout, device=DEV)
    packed = hip.conv_pack_weights(w, R, R, (1, 1), "f16")
    assert packed.compute == "f16"
    hip.conv_force_tile(mt, nb)
    try:
        got = hip.block_conv_cl(x, packed, b, cout, (k, k), (1, 1))
    finally:
        hip.conv_force_tile(0, 0)
    assert got is not None and hip.is_cl(got)
    xh, wh = x.half().double(), w.half().double()
    want_h = torch.nn.functional.conv2d(xh, wh, b.double()).float()
    torch.testing.assert_close(got, want_h, rtol=0, atol=2e-5)
    want = oracle.block_conv(x.contiguous().cpu(), w.cpu(), b.cpu(), 1)
    torch.testing.assert_close(got.cpu(), want, rtol=F16_RTOL, atol=F16_ATOL)


def test_f16_compute_stride2_falls_back_to_fp32_packing(hip):
    w = torch.randn(64, 64, 3, 3, device=DEV)
    assert hip.conv_pack_weights(w, 5, 5, (2, 2), "f16").compute == "f32"


@pytest.mark.parametrize("act", ["swish", "identity"])
def test_f16_compute_fused_gather_and_scatter_gather(hip, act):
    """gather -> conv and scatter_gather -> conv (-> scatter) with f16 compute against the same fused kernels in fp32."""
    from sige_amd.utils import reduce_mask

    torch.manual_seed(5)
    C, Co, H = 128, 128, 64
    mask = torch.zeros(H, H, dtype=torch.bool, device=DEV)
    mask[20:41, 12:40] = True
    mask[0, 0] = mask[63, 63] = True
    idx = reduce_mask(mask, 6, 4, 1)
    x, y = _cl(torch.randn(1, C, H, H, device=DEV)), _cl(torch.randn(1, C, H, H, device=DEV))
    sc, sh = torch.randn(1, C, 1, 1, device=DEV) * 0.5 + 1, torch.randn(1, C, 1, 1, device=DEV) * 0.2
    sc_, sh_ = (sc, sh) if act == "swish" else (None, None)
    w = torch.randn(Co, C, 3, 3, device=DEV) / (3 * C ** 0.5)
    b = torch.randn(Co, device=DEV)
    p32, p16 = hip.conv_pack_weights(w, 6, 6, (1, 1)), hip.conv_pack_weights(w, 6, 6, (1, 1), "f16")
    a32 = hip.gather_conv_cl(x, None, (6, 6), idx, sc_, sh_, act, p32, b, Co, (3, 3), (1, 1))
    a16 = hip.gather_conv_cl(x, None, (6, 6), idx, sc_, sh_, act, p16, b, Co, (3, 3), (1, 1))
    torch.testing.assert_close(a16, a32, rtol=F16_RTOL, atol=F16_ATOL)
    assert (a16 - a32).abs().max() > 0  # really another arithmetic
    # two-pointer input (fused torch.cat) + out_affine epilogue, written into a full tensor (a dense layer)
    xa, xb = _cl(torch.randn(1, 64, 32, 32, device=DEV)), _cl(torch.randn(1, 64, 32, 32, device=DEV))
    all_idx = hip.all_tiles(32, 32, (4, 4), (1, 1), (1, 1), DEV)
    res = _cl(torch.randn(1, Co, 32, 32, device=DEV))
    oa = (torch.randn(Co, device=DEV) * 0.3 + 1, torch.randn(Co, device=DEV) * 0.1, "swish")
    kw = dict(full=dict(offset=(1, 1), out_res=(32, 32), residual=res), out_affine=oa)
    d32 = hip.gather_conv_cl(xa, xb, (6, 6), all_idx, sc_, sh_, act, p32, b, Co, (3, 3), (1, 1), **kw)
    d16 = hip.gather_conv_cl(xa, xb, (6, 6), all_idx, sc_, sh_, act, p16, b, Co, (3, 3), (1, 1), **kw)
    torch.testing.assert_close(d16, d32, rtol=F16_RTOL, atol=F16_ATOL)
    # scatter_gather -> conv, tiles and fused scatter
    smap = hip.get_scatter_map(H, H, 6, 6, 3, 3, 1, 1, 1, 1, idx)
    t4 = _cl(torch.randn(idx.shape[0], C, 4, 4, device=DEV))
    s32 = hip.scatter_gather_conv_cl(t4, y, (6, 6), idx, smap, sc_, sh_, act, p32, b, Co, (3, 3), (1, 1))
    s16 = hip.scatter_gather_conv_cl(t4, y, (6, 6), idx, smap, sc_, sh_, act, p16, b, Co, (3, 3), (1, 1))
    torch.testing.assert_close(s16, s32, rtol=F16_RTOL, atol=F16_ATOL)
    o32, o16 = y.clone(memory_format=torch.preserve_format), y.clone(memory_format=torch.preserve_format)
    r = _cl(torch.randn(1, Co, H, H, device=DEV))
    hip.scatter_gather_conv_scatter_cl(t4, y, (6, 6), idx, smap, sc_, sh_, act, p32, b, Co, (3, 3), (1, 1), o32, residual=r)
    hip.scatter_gather_conv_scatter_cl(t4, y, (6, 6), idx, smap, sc_, sh_, act, p16, b, Co, (3, 3), (1, 1), o16, residual=r)
    torch.testing.assert_close(o16, o32, rtol=F16_RTOL, atol=F16_ATOL)


def test_f16_compute_ddpm_forward_vs_fp32_oracle(hip):
    """BASELINE.json configs[4] at its own size: the DDPM-256 sparse forward (ch 128, channels-last, in-place buffers) with
    f16-compute convs against the fp32 CPU oracle network; stated tolerance 2e-2 abs / 1e-2 rel."""
    import bench
    from sige_amd import runtime
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval()
    x0, noise = bench.make_inputs()
    mask = bench.edit_mask(0.05)
    x1 = x0 + noise * mask
    t = torch.zeros(1)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    oracle.set_num_threads(min(32, os.cpu_count() or 1))
    runtime.register_backend("cpu", oracle)
    try:
        with torch.no_grad():
            model.set_mode("full")
            model(x0, t)
            model.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
            model.set_mode("sparse")
            sparse_c = model(x1, t)
    finally:
        runtime.unregister_backend("cpu")
    model.clear_cache()
    model = model.to(DEV).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    cl = lambda a: a.to(DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    with torch.no_grad():
        model.set_mode("full")
        model(cl(x0), t.to(DEV))
        model.set_masks(downsample_mask(dilate_mask(mask.to(DEV), 5), 8))
        model.set_mode("sparse")
        f32 = model(cl(x1), t.to(DEV)).clone()
        model.set_compute_dtype("f16")
        f16 = model(cl(x1), t.to(DEV)).clone()
        model.set_compute_dtype("f32")
    torch.testing.assert_close(f32.cpu(), sparse_c, rtol=0, atol=util.CONV_ATOL)
    torch.testing.assert_close(f16.cpu(), sparse_c, rtol=F16_RTOL, atol=F16_ATOL)
    assert (f16 - f32).abs().max() > 1e-6


# ---- horizontal fusion: a residual block's 1x1 shortcut launched inside the kernel of its conv1 ---------------------
def _pair_case(hip, res, c1, c2, cout, T, full, seed=0, residual=False, compute="f32"):
    """(run_shortcut, run_conv1) of a residual block at `res`: both gather from the same (optionally concatenated) input."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    cin = c1 + c2
    x, x2 = _cl(r(1, c1, res, res)), (_cl(r(1, c2, res, res)) if c2 else None)
    w3, b3, w1, b1 = r(cout, cin, 3, 3) / (3 * cin ** 0.5), r(cout), r(cout, cin, 1, 1) / cin ** 0.5, r(cout)
    sc, sh, os_, oh_ = r(1, cin, 1, 1), r(1, cin, 1, 1), r(cout), r(cout)
    p3, p1 = hip.conv_pack_weights(w3, 6, 6, (1, 1), compute), hip.conv_pack_weights(w1, 4, 4, (1, 1), compute)
    if full:
        i6, i4 = hip.all_tiles(res, res, (4, 4), (1, 1), (1, 1), DEV), hip.all_tiles(res, res, (4, 4), (1, 1), (0, 0), DEV)
        f6 = dict(offset=(1, 1), out_res=(res, res), residual=_cl(r(1, cout, res, res)) if residual else None)
        f4 = dict(offset=(0, 0), out_res=(res, res), residual=None)
    else:
        n = res // 4
        cells = torch.randperm(n * n, generator=g)[:T]
        i4 = torch.stack([cells // n * 4, cells % n * 4], 1).int().to(DEV)
        i6, f6, f4 = (i4 - 1).contiguous(), None, None
    shortcut = lambda: hip.gather_conv_cl(x, x2, (4, 4), i4, None, None, "identity", p1, b1, cout, (1, 1), (1, 1), full=f4)  # noqa: E731
    conv1 = lambda: hip.gather_conv_cl(x, x2, (6, 6), i6, sc, sh, "swish", p3, b3, cout, (3, 3), (1, 1), full=f6,  # noqa: E731
                                       out_affine=(os_, oh_, "swish"))
    return shortcut, conv1


@pytest.mark.parametrize("res,c1,c2,cout,T,full", [
    (64, 128, 0, 256, 18, False),    # SIGE block, few tiles
    (256, 128, 128, 128, 106, False),  # SIGE up-path block: fused torch.cat, many tiles
    (128, 256, 128, 128, 40, False),
    (32, 256, 256, 256, 0, True),    # dense remainder at 32x32
    (16, 512, 512, 512, 0, True),    # 16x16: 8-wave workgroups
    (8, 512, 512, 512, 0, True),     # 8x8: conv1 is K-split across workgroups (second pass), the shortcut rides along
    (16, 256, 0, 512, 0, True),
])
def test_conv_pair_equals_separate_launches(hip, res, c1, c2, cout, T, full):
    """sige_hip_conv_pair_begin / _end: the pair kernel runs the same two workgroup programs, so conv1 is bit-identical
    to its own launch; the shortcut may pick another output block inside a pair (fp32 summation order only)."""
    shortcut, conv1 = _pair_case(hip, res, c1, c2, cout, T, full)
    want_s, want_c = shortcut(), conv1()
    n0, f0 = hip.launch_count(), hip.conv_pairs_fused()
    with hip.conv_pair(want_s):
        got_s = shortcut()
        got_c = conv1()
    torch.cuda.synchronize()
    assert hip.conv_pairs_fused() == f0 + 1, "no pair kernel for this combination"
    assert hip.launch_count() == n0 + 1  # (8x8: the K split of conv1 is finished inside the launch, too)
    assert torch.equal(got_c, want_c)
    torch.testing.assert_close(got_s, want_s, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("res,c1,c2,cout,T,full", [(64, 128, 0, 256, 18, False), (256, 128, 128, 128, 106, False),
                                                   (32, 256, 256, 256, 0, True), (16, 512, 512, 512, 0, True)])
def test_conv_pair_f16_compute(hip, res, c1, c2, cout, T, full):
    """The same pairing for the f16-compute kernels (ConvGeoH)."""
    shortcut, conv1 = _pair_case(hip, res, c1, c2, cout, T, full, compute="f16")
    want_s, want_c = shortcut(), conv1()
    n0, f0 = hip.launch_count(), hip.conv_pairs_fused()
    with hip.conv_pair(want_s):
        got_s = shortcut()
        got_c = conv1()
    torch.cuda.synchronize()
    assert hip.conv_pairs_fused() == f0 + 1 and hip.launch_count() == n0 + 1
    assert torch.equal(got_c, want_c)
    torch.testing.assert_close(got_s, want_s, rtol=1e-5, atol=1e-5)


def test_conv_pair_leftovers_are_launched(hip):
    """A held shortcut with no partner is launched when the block ends, or in front of the next launch that is not a
    conv1 -- order and results as without pairing."""
    shortcut, conv1 = _pair_case(hip, 64, 128, 0, 256, 18, False, seed=1)
    want_s, want_c = shortcut(), conv1()
    f0, n0 = hip.conv_pairs_fused(), hip.launch_count()
    with hip.conv_pair(want_s):
        got_s = shortcut()              # held ...
    torch.cuda.synchronize()            # ... and launched by pair_end
    assert torch.equal(got_s, want_s) and hip.launch_count() == n0 + 1
    with hip.conv_pair(want_s):
        got_c = conv1()                 # a conv1 with nothing held: a plain launch
        got_s2 = shortcut()             # held, launched at the end
    torch.cuda.synchronize()
    assert torch.equal(got_c, want_c) and torch.equal(got_s2, want_s)
    with hip.conv_pair(want_s):
        a = shortcut()                  # held
        b = shortcut()                  # not a partner: the first one goes out on its own, this one is held in turn
        c = conv1()                     # pairs with the second
    torch.cuda.synchronize()
    assert torch.equal(a, want_s) and torch.equal(c, want_c)
    torch.testing.assert_close(b, want_s, rtol=1e-5, atol=1e-5)
    assert hip.conv_pairs_fused() == f0 + 1
    assert hip.launch_count() == n0 + 1 + 2 + 2
    # outside a `with` nothing is held
    assert torch.equal(shortcut(), want_s)


@pytest.mark.selfcheck
def test_ddpm_forward_paired_vs_unpaired(hip):
    """The benchmark network with and without the shortcut pairing: 20 launches fewer, and BOTH forms within the north_star
    tolerance of the CPU oracle's sparse forward (tests/util.py ddpm_cpu_oracle; reference natives
    /root/reference/sige/cpu/gather.cpp:4-58, scatter_gather.cpp:5-56).  The two forms differ in the fp32 summation order of the
    1x1 shortcuts, which the random-weight network amplifies: their mutual difference is recorded, and bounded by the same 1e-3."""
    import bench
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet, ResBlock

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval().to(DEV).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = bench.make_inputs()
    mask = bench.edit_mask(0.012)
    _, (want,) = util.ddpm_cpu_oracle([mask])
    x0, x1, t = _cl(x0.to(DEV)), _cl((x0 + noise * mask).to(DEV)), torch.zeros(1, device=DEV)
    outs, launches = [], []
    with util.native_full_pass(), torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        model.set_masks(downsample_mask(dilate_mask(mask.to(DEV), 5), 8))
        model.set_mode("sparse")
        model(x1, t)  # (registers the activated twins; the NEXT forward builds their persistent buffers -- library launches
        model(x1, t)  #  since round 4 (scatter._fill), which the launch counts below must not include)
        for pair in (False, True):
            for m in model.modules():
                if isinstance(m, ResBlock):
                    m.pair = pair
            model(x1, t)
            n0, f0 = hip.launch_count(), hip.conv_pairs_fused()
            outs.append(model(x1, t).clone())
            launches.append((hip.launch_count() - n0, hip.conv_pairs_fused() - f0))
    torch.cuda.synchronize()
    for name, o in zip(("unpaired", "paired"), outs):
        err = util.record_margin("paired_vs_unpaired", name + " vs cpu oracle", (o.cpu() - want).abs().max(), util.CONV_ATOL)
        assert err <= util.CONV_ATOL, (name, err)
    diff = util.record_margin("paired_vs_unpaired", "paired vs unpaired", (outs[1] - outs[0]).abs().max(), util.SELF_ATOL)
    assert diff <= util.SELF_ATOL, diff
    assert launches[0][1] == 0 and launches[1][1] >= 15, launches
    assert launches[1][0] == launches[0][0] - launches[1][1], launches


@pytest.mark.parametrize("res,c1,c2,cout,residual", [(8, 512, 512, 512, False), (8, 512, 0, 512, True), (16, 512, 512, 512, True),
                                                      (8, 512, 0, 256, True)])
def test_ksplit_finished_inside_the_launch(hip, res, c1, c2, cout, residual, tuning):
    """Cross-workgroup K split: the last workgroup of an output block adds the partial copies up in split order and runs the
    epilogue -- bit-identical to the second-pass kernel, one launch instead of two; the tickets are back at zero afterwards
    and no workgroup reads a stale partial sum (two different inputs alternate over the same workspace memory, eagerly and
    in graph replays)."""
    convs = [_pair_case(hip, res, c1, c2, cout, 0, True, seed=3 + i, residual=residual)[1] for i in range(2)]
    try:
        hip.conv_force_ksplit(4)
        hip.conv_force_ksplit_pass(True)
        n0 = hip.launch_count()
        want = [c() for c in convs]
        assert hip.launch_count() == n0 + 4
        assert not torch.equal(want[0], want[1])
        hip.conv_force_ksplit_pass(False)
        n0 = hip.launch_count()
        got = [convs[i % 2]() for i in range(60)]
        assert hip.launch_count() == n0 + 60
        torch.cuda.synchronize()
        assert all(torch.equal(o, want[i % 2]) for i, o in enumerate(got))
        del got
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            convs[0]()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                outs = [convs[i % 2]() for i in range(6)]
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(o, want[i % 2]) for i, o in enumerate(outs))
    finally:
        hip.conv_force_ksplit(0)
        hip.conv_force_ksplit_pass(False)


# ---- NCHW gather, grouped row form (8 consecutive tiles per workgroup) ----------------------------------------------
@pytest.mark.parametrize("bsize,B,C,res,act,first", [(6, 1, 256, 128, "swish", False), (6, 2, 160, 96, "identity", False),
                                                      (4, 1, 256, 128, "identity", False), (5, 1, 264, 128, "swish", True)])
def test_grouped_nchw_gather_bit_exact(hip, bsize, B, C, res, act, first, tuning):
    """Enough tiles for the grouped form (merged cache-line requests for neighbouring tiles): bit-identical to the one-tile
    row form and to the oracle, including tiles over the image border, a ragged last group and a ragged channel chunk."""
    g = torch.Generator().manual_seed(bsize * 100 + C)
    stride = 4 if bsize != 5 else 4
    n_side = res // stride
    keep = torch.rand(n_side, n_side, generator=g) < 0.75          # most tiles active: long horizontal runs, some holes
    cells = keep.nonzero()
    idx = (cells * stride - 1).int().contiguous()                   # origins -1, 3, ...: first row / column start outside the image
    x = torch.randn(B, C, res, res, generator=g)
    scale, shift = (torch.randn(1, C, 1, 1, generator=g), torch.randn(1, C, 1, 1, generator=g)) if act == "swish" else (None, None)
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    try:
        hip.gather_force_rows(True)
        rows = hip.gather(d(x), bsize, bsize, d(idx), d(scale), d(shift), act, first)
        hip.gather_force_rows(False)
        grouped = hip.gather(d(x), bsize, bsize, d(idx), d(scale), d(shift), act, first)
    finally:
        hip.gather_force_rows(False)
    torch.cuda.synchronize()
    assert torch.equal(grouped, rows)
    want = oracle.gather(x, bsize, bsize, idx, scale, shift, act, first)
    torch.testing.assert_close(grouped.cpu(), want, rtol=1e-6, atol=1e-6)


# ---- producer-side activation of the conv1 inputs (activated twins; WIP: first validation pending) -------------------
def test_twin_epilogue_of_a_dense_conv(hip):
    """gather -> conv -> full tensor with two twins: twin_k = SiLU(scale_k * result + shift_k), the primary output unchanged."""
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    C, Co, res = 128, 256, 16
    x, res_t = _cl(r(1, C, res, res)), _cl(r(1, Co, res, res))
    w, b = r(Co, C, 3, 3) / (3 * C ** 0.5), r(Co)
    p = hip.conv_pack_weights(w, 6, 6, (1, 1))
    idx = hip.all_tiles(res, res, (4, 4), (1, 1), (1, 1), DEV)
    full = dict(offset=(1, 1), out_res=(res, res), residual=res_t)
    want = hip.gather_conv_cl(x, None, (6, 6), idx, None, None, "identity", p, b, Co, (3, 3), (1, 1), full=full)
    tw = [(_cl(torch.empty(1, Co, res, res, device=DEV)), r(Co), r(Co)) for _ in range(2)]
    got = hip.gather_conv_cl(x, None, (6, 6), idx, None, None, "identity", p, b, Co, (3, 3), (1, 1), full=full, twins=tw)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    for buf, sc, sh in tw:
        torch.testing.assert_close(buf, torch.nn.functional.silu(want * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), rtol=1e-5, atol=1e-5)


@pytest.mark.selfcheck
def test_ddpm_forward_twins_vs_no_twins(hip):
    """The benchmark network with conv1 inputs activated by their producers (cfg.conv1_twins) and the same network with conv1
    activating in its staging path: BOTH within the north_star tolerance (1e-3) of the CPU oracle's sparse forward of the same
    edit (tests/util.py ddpm_cpu_oracle -- the reference's natives, /root/reference/sige/cpu/gather.cpp:4-58,
    scatter.cpp:4-68, scatter_gather.cpp:5-56), same launch count, and the twins are really used (most residual blocks found
    both of theirs).  Also after a mask change and after a new full pass (a stale twin is a cached activation of another image:
    an error of 1e-2 and more, not of 1e-4).  The two HIP forms differ from each other by fp32 rounding (swish in the staging
    path vs in the producer's epilogue) amplified by the random-weight network -- 1.3e-4 on one box, 7e-3 on an input with other
    statistics: their mutual difference is recorded (gpurun_out/test_margins.jsonl), not a criterion tighter than the row's."""
    import bench
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet, ResBlock

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval().to(DEV).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = bench.make_inputs()
    t = torch.zeros(1, device=DEV)
    blocks = [m for m in model.modules() if isinstance(m, ResBlock)]
    ratios = (0.012, 0.15, 0.05)  # (mask changes: the persistent twins are rebuilt from the cache)
    _, wants = util.ddpm_cpu_oracle([bench.edit_mask(r) for r in ratios])
    _, (want_flip,) = util.ddpm_cpu_oracle([bench.edit_mask(0.012)], flip=True)

    def run(ratio, twins, original=None):
        mask = bench.edit_mask(ratio)
        x1 = _cl(((x0 if original is None else original) + noise * mask).to(DEV))  # (= the original outside the mask)
        for b in blocks:
            b.use_twins = twins
            b._drop_twin_links()
        model.set_masks(downsample_mask(dilate_mask(mask.to(DEV), 5), 8))
        model.set_mode("sparse")
        for _ in range(3):  # (forward 1 registers the consumers, from forward 2 on the twins exist)
            out = model(x1, t)
        n0 = hip.launch_count()
        out = model(x1, t).clone()
        return out, hip.launch_count() - n0

    def check(tag, ref, got, want):
        for name, o in (("no twins", ref), ("twins", got)):
            err = util.record_margin("twins_vs_no_twins", "%s, %s vs cpu oracle" % (tag, name), (o.cpu() - want).abs().max(),
                                     util.CONV_ATOL)
            assert err <= util.CONV_ATOL, (tag, name, err)
        diff = util.record_margin("twins_vs_no_twins", "%s, twins vs no twins" % tag, (got - ref).abs().max(), util.SELF_ATOL)
        assert diff <= util.SELF_ATOL, (tag, diff)

    with util.native_full_pass(), torch.no_grad():
        model.set_mode("full")
        model(_cl(x0.to(DEV)), t)
        ref_first = None
        for ratio, want in zip(ratios, wants):
            ref, n_ref = run(ratio, False)
            ref_first = ref if ref_first is None else ref_first
            got, n_got = run(ratio, True)
            check("ratio %g" % ratio, ref, got, want)
            assert n_got == n_ref
            linked = sum(1 for b in blocks if b._twin_links)
            assert linked >= 20, linked
        # a new original image (the mirror image): new caches, new affines -> the old twins must not be used
        x0b = x0.flip(-1).contiguous()
        model.set_mode("full")
        model(_cl(x0b.to(DEV)), t)
        ref, _ = run(0.012, False, original=x0b)
        got, _ = run(0.012, True, original=x0b)
        check("mirrored original", ref, got, want_flip)
        assert (got - ref_first).abs().max() > 1e-2  # (and it IS a different result than for the first original)
