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

    def sparse(inplace, orig, edited):
        net.set_scatter_inplace(inplace)
        net.set_mode("full")
        net(orig)
        net.set_mode("sparse")
        net.set_masks(masks)
        return net(edited).clone()

    with torch.no_grad():
        for k in range(6):  # full(A) -> sparse -> full(B) -> sparse ... with the SAME masks: addresses get recycled
            orig = imgs[k % 3]
            e = edit(orig)
            want = sparse(False, orig, e)
            got = sparse(True, orig, e)
            assert torch.equal(got, want), k
        # in-place + the block-residual fallback branch (more shortcut tiles than main tiles cannot happen for masks
        # derived from one mask; force the unfused path instead)
        from sige_amd.nn import deferred

        deferred.FUSION = False
        try:
            got = sparse(True, imgs[0], e)
            want = sparse(False, imgs[0], e)
        finally:
            deferred.FUSION = True
        assert torch.equal(got, want)


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
