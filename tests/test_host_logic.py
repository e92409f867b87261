"""Host-side logic of sige_amd (module API, mask helpers, caches) on CPU.

The GPU library cannot run here, so these tests register the CPU ORACLE as the
"cpu" backend of the runtime registry -- test infrastructure standing in for
the native functions, exactly where the reference would use `sige.cpu`.  The
product default has no "cpu" backend (see test_no_cpu_backend_by_default)."""
import warnings

import numpy as np
import pytest
import torch
from torch import nn

import sige_amd
from oracle import oracle
from sige_amd import runtime
from sige_amd.nn import (Gather, Scatter, ScatterGather, ScatterWithBlockResidual, SIGEConv2d, SIGEModel,
                         SIGEModule)
from sige_amd.utils import compute_difference_mask, dilate_mask, downsample_mask, reduce_mask
from tests import util
from tests.golden_cases import CASES


@pytest.fixture()
def cpu_oracle_backend():
    runtime.register_backend("cpu", oracle)
    yield
    runtime.unregister_backend("cpu")


# ----------------------------------------------------------------- masks ----
def _fixture_names():
    return sorted({k.split("/")[0] for k in util.golden("masks").files})


@pytest.mark.parametrize("name", _fixture_names())
def test_mask_helpers_bit_exact(name):
    g = util.golden("masks")
    shape = g[name + "/shape"]
    mask = util.unpack(g[name + "/mask"], shape)
    for gname, (b, s, p) in {"b6s4p1": (6, 4, 1), "b4s4p0": (4, 4, 0), "b5s4p0": (5, 4, 0), "b5s4p1": (5, 4, 1)}.items():
        got = reduce_mask(mask, b, s, p)
        assert got.dtype == torch.int32 and got.is_contiguous()
        assert torch.equal(got, torch.from_numpy(g["%s/reduce/%s" % (name, gname)]))
    for dil in (1, 2, 5):
        assert torch.equal(dilate_mask(mask, dil), util.unpack(g["%s/dilate/%d" % (name, dil)], shape))
        assert np.array_equal(dilate_mask(mask.numpy(), dil), util.unpack(g["%s/dilate/%d" % (name, dil)], shape).numpy())
    for min_res, dil in ((8, 1), (8, 2), (4, 1)):
        pyr = downsample_mask(mask, min_res=min_res, dilation=dil)
        assert len(pyr) == len([k for k in g.files if k.startswith("%s/pyramid/%d_%d/" % (name, min_res, dil))])
        for (h, w), pm in pyr.items():
            assert torch.equal(pm, util.unpack(g["%s/pyramid/%d_%d/%dx%d" % (name, min_res, dil, h, w)], (h, w)))
    for (h, w), pm in downsample_mask(dilate_mask(mask, 5), min_res=8).items():
        assert torch.equal(pm, util.unpack(g["%s/ddpm/%dx%d" % (name, h, w)], (h, w)))


@pytest.mark.parametrize("case", CASES, ids=util.case_ids())
def test_reduce_mask_cases(case):
    g = case["geom"]
    d = util.tensors(case)
    assert torch.equal(reduce_mask(d["mask"], g.block, g.block_stride, g.offset), util.ref(case, "idx"))


def test_reduce_mask_edge_cases():
    assert reduce_mask(torch.zeros(8, 8, dtype=torch.bool), None, 4, 1) is None
    empty = reduce_mask(torch.zeros(16, 16, dtype=torch.bool), 6, 4, 1)
    assert empty.shape == (0, 2) and empty.dtype == torch.int32  # all-false mask -> int32 [0,2]
    full = reduce_mask(torch.ones(256, 256, dtype=torch.bool), 6, 4, 1)
    assert full.shape == (65 * 65 - 65 - 64, 2) or full.shape[0] <= 65 * 65  # bottom/right candidates past the image drop out
    assert full[0].tolist() == [-1, -1]


def test_compute_difference_mask():
    a = torch.zeros(1, 3, 8, 8)
    b = a.clone()
    b[0, 1, 2, 3] = 0.5
    b[0, 2, 5, 5] = 0.01
    m = compute_difference_mask(a, b)
    assert m.shape == (8, 8) and m.sum() == 1 and m[2, 3]
    assert torch.equal(compute_difference_mask(a[0], b[0]), m)
    assert compute_difference_mask(a[0, 1], b[0, 1]).sum() == 1


# ------------------------------------------------------------- geometry ----
@pytest.mark.parametrize("case", CASES, ids=util.case_ids())
def test_gather_geometry(case):
    g = case["geom"]
    conv = nn.Conv2d(4, 4, g.kernel, g.stride, g.padding)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ga = Gather(conv, (6, 6) if g.block == (5, 5) else g.block)
    assert ga.block_size == g.block and ga.block_stride == g.block_stride and ga.offset == g.offset
    assert ga.out_tile == g.out_tile
    assert bool(w) == (g.block == (5, 5))  # stride-2: "Change the block size from (6, 6) to (5, 5)"


# --------------------------------------------------------------- modules ----
class ExampleModule(SIGEModule):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = SIGEConv2d(cin, cout, 3, 1, 1, bias=True)
        self.gather = Gather(self.conv, block_size=6)
        self.scatter = Scatter(self.gather)

    def forward(self, x):
        return self.scatter(self.conv(self.gather(x)))


class ExampleModel(SIGEModel):
    def __init__(self, cin, cout):
        super().__init__()
        self.m = ExampleModule(cin, cout)

    def forward(self, x):
        return self.m(x)


def _example_inputs():
    g = util.golden("masks")
    mask = util.unpack(g["assets_mask/mask"], g["assets_mask/shape"])
    rs = np.random.RandomState(4242)
    orig = rs.standard_normal((1, 16, 256, 256)).astype(np.float32)
    noise = rs.standard_normal((1, 16, 256, 256)).astype(np.float32)
    w = (rs.standard_normal((32, 16, 3, 3)) / np.sqrt(9 * 16)).astype(np.float32)
    b = rs.standard_normal((32,)).astype(np.float32)
    edited = orig + noise * mask.numpy()[None, None]
    return mask, torch.from_numpy(orig), torch.from_numpy(edited), torch.from_numpy(w), torch.from_numpy(b)


def test_example_py_matches_reference(cpu_oracle_backend):
    """example.py: Gather -> 3x3 conv -> Scatter equals the dense conv
    within atol 1e-4 (example.py:95) and equals the reference's sparse output."""
    mask, orig, edited, w, b = _example_inputs()
    model = ExampleModel(16, 32).eval()
    with torch.no_grad():
        model.m.conv.weight.copy_(w)
        model.m.conv.bias.copy_(b)
        model.set_mode("full")
        std = model(edited)
        model(orig)
        model.set_mode("sparse")
        model.set_masks({(256, 256): mask})
        sp = model(edited)
    ex = util.golden("example")
    assert np.array_equal(model.m.gather.active_indices.numpy(), ex["c16_32/idx"])
    assert torch.isclose(std, sp, atol=1e-4).all()
    torch.testing.assert_close(sp[:, ::4, ::3, ::3], torch.from_numpy(ex["c16_32/sparse_sub"]), rtol=0, atol=1e-6)
    assert abs(sp.double().sum().item() - float(ex["c16_32/sparse_sum"])) < 1e-2


def test_no_cpu_backend_by_default():
    """The product has no CPU path: a sparse op on a CPU tensor fails loudly."""
    mask, orig, edited, w, b = _example_inputs()
    model = ExampleModel(16, 32).eval()
    with torch.no_grad():
        model.set_mode("full")
        model(orig)
        model.set_mode("sparse")
        model.set_masks({(256, 256): mask})
        with pytest.raises(RuntimeError, match="no native backend"):
            model(edited)


def test_dtype_and_dim_checks(cpu_oracle_backend):
    conv = SIGEConv2d(4, 4, 3, 1, 1)
    g = Gather(conv, 6)
    with pytest.raises(NotImplementedError, match="does not support dtype"):
        g(torch.zeros(1, 4, 8, 8, dtype=torch.float16))
    with pytest.raises(NotImplementedError, match="does not support input with dim"):
        g(torch.zeros(4, 8, 8))
    g.set_mode("bogus")
    with pytest.raises(NotImplementedError, match="Unknown mode"):
        g(torch.zeros(1, 4, 8, 8))


class ResBlock(SIGEModule):
    """The op pattern of SIGEFusedResnetBlock.sparse_forward
    (diffusion/models/ddpm_arch/sige_fused_unet.py:100-131) with a 1x1 shortcut."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = SIGEConv2d(cin, cout, 3, 1, 1)
        self.conv2 = SIGEConv2d(cout, cout, 3, 1, 1)
        self.nin = SIGEConv2d(cin, cout, 1, 1, 0)
        self.main_gather = Gather(self.conv1, 6, activation_name="swish")
        self.scatter_gather = ScatterGather(self.main_gather, activation_name="swish")
        self.shortcut_gather = Gather(self.nin, 4)
        self.scatter = ScatterWithBlockResidual(self.main_gather, self.shortcut_gather)
        self.s1 = self.t1 = self.s2 = self.t2 = None

    def forward(self, x):
        if self.mode == "full":
            sc = self.nin(self.shortcut_gather(x))
            h = self.main_gather(x)
            h = torch.nn.functional.silu(h * self.s1 + self.t1)
            h = self.scatter_gather(self.conv1(h))
            h = torch.nn.functional.silu(h * self.s2 + self.t2)
            return self.scatter(self.conv2(h), sc)
        sc = self.nin(self.shortcut_gather(x))
        h = self.conv1(self.main_gather(x, self.s1, self.t1))
        h = self.conv2(self.scatter_gather(h, self.s2, self.t2))
        return self.scatter(h, sc)


class ResNet(SIGEModel):
    def __init__(self, cin, cout):
        super().__init__()
        self.block = ResBlock(cin, cout)

    def forward(self, x):
        return self.block(x)


def test_resblock_sparse_equals_dense_on_edited_region(cpu_oracle_backend):
    """With per-channel affine (no data-dependent statistics) the sparse forward
    must reproduce the dense forward of the edited input everywhere."""
    torch.manual_seed(0)
    net = ResNet(6, 8).eval()
    blk = net.block
    blk.s1, blk.t1 = torch.randn(1, 6, 1, 1), torch.randn(1, 6, 1, 1)
    blk.s2, blk.t2 = torch.randn(1, 8, 1, 1), torch.randn(1, 8, 1, 1)
    orig = torch.randn(1, 6, 40, 44)
    mask = torch.zeros(40, 44, dtype=torch.bool)
    mask[10:19, 7:23] = True
    mask[0, 0] = mask[39, 43] = True
    edited = orig + torch.randn_like(orig) * mask
    with torch.no_grad():
        net.set_mode("full")
        dense = net(edited)
        base = net(orig)
        net.set_mode("sparse")
        # two stacked 3x3 convs: square receptive field of radius 2 (dilate_mask's own
        # structuring element is plus-shaped, so dilate rows then columns)
        net.set_masks({(40, 44): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))})
        sparse = net(edited)
    torch.testing.assert_close(sparse, dense, rtol=0, atol=1e-4)
    assert not torch.allclose(base, dense, atol=1e-3)
    # caches untouched by the sparse pass (out-of-place semantics, scatter.py:59-60)
    assert torch.equal(blk.scatter.original_outputs[0], base)
    # shared-cache memoisation: both modules see the same index tensor object
    assert blk.scatter_gather.gather.module.active_indices is blk.main_gather.active_indices
    assert blk.scatter_gather.scatter_map.shape == (40, 44, 3)


def test_sparse_update_refreshes_caches(cpu_oracle_backend):
    torch.manual_seed(1)
    net = ResNet(4, 4).eval()
    blk = net.block
    blk.s1, blk.t1, blk.s2, blk.t2 = (torch.randn(1, 4, 1, 1) for _ in range(4))
    orig = torch.randn(1, 4, 24, 24)
    mask = torch.zeros(24, 24, dtype=torch.bool)
    mask[5:9, 5:9] = True
    edited = orig + mask * 1.0
    with torch.no_grad():
        net.set_mode("full")
        net(orig)
        dense = net(edited)
        net(orig)
        net.set_mode("sparse")
        net.set_masks({(24, 24): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))})
        net.set_sparse_update(True)
        out = net(edited)
    torch.testing.assert_close(out, dense, rtol=0, atol=1e-4)
    assert torch.equal(blk.scatter.original_outputs[0], out)  # cache now holds the edited result
    torch.testing.assert_close(blk.scatter.original_residuals[0], blk.nin(edited), rtol=0, atol=1e-5)


def test_profile_mode_shapes(cpu_oracle_backend):
    torch.manual_seed(2)
    net = ResNet(4, 6).eval()
    blk = net.block
    blk.s1, blk.t1 = torch.randn(1, 4, 1, 1), torch.randn(1, 4, 1, 1)
    blk.s2, blk.t2 = torch.randn(1, 6, 1, 1), torch.randn(1, 6, 1, 1)
    x = torch.randn(1, 4, 16, 16)
    mask = torch.zeros(16, 16, dtype=torch.bool)
    mask[4:6, 4:6] = True
    with torch.no_grad():
        net.set_mode("full")
        net(x)
        net.set_masks({(16, 16): mask})
        net.set_mode("profile")
        out = net(x)
    assert out.shape == (1, 6, 16, 16)


def test_cache_id_and_clear_cache(cpu_oracle_backend):
    net = ExampleModel(4, 4).eval()
    x = torch.randn(1, 4, 16, 16)
    with torch.no_grad():
        net.set_mode("full")
        net.set_cache_id(3)
        net(x)
    assert list(net.m.scatter.original_outputs) == [3]
    net.clear_cache()
    assert net.m.scatter.original_outputs == {}


def test_compat_install_aliases():
    import sys

    from sige_amd import compat

    saved = {k: v for k, v in sys.modules.items() if k == "sige" or k.startswith("sige.")}
    try:
        compat.install()
        import sige
        from sige.nn import Gather as G2
        from sige.utils import reduce_mask as r2

        assert G2 is Gather and r2 is reduce_mask and sige.__version__ == sige_amd.__version__
    finally:
        for k in [k for k in sys.modules if k == "sige" or k.startswith("sige.")]:
            del sys.modules[k]
        sys.modules.update(saved)


# ------------------------------------------------------- deferred tiles ----
def test_deferred_tiles_behave_like_tensors():
    from sige_amd.nn.deferred import DeferredTiles

    calls = []

    def thunk():
        calls.append(1)
        return torch.arange(24, dtype=torch.float32).reshape(2, 3, 2, 2)

    t = DeferredTiles((2, 3, 2, 2), torch.float32, torch.device("cpu"), thunk, dict(kind="gather"))
    assert t.shape == (2, 3, 2, 2) and t.dim() == 4 and t.dtype == torch.float32 and t.size(1) == 3
    assert t.contiguous() is t and not calls and t.spec is not None  # metadata never materialises
    y = t * 2  # any real op does, exactly once
    assert calls == [1] and t.spec is None and type(y) is torch.Tensor and y[1, 2, 1, 1] == 46
    assert torch.equal(torch.cat([t, t])[2:], t.materialize()) and calls == [1]
    assert torch.equal(t.view(2, -1), t.materialize().view(2, -1))


def test_forced_deferral_matches_eager(cpu_oracle_backend, monkeypatch):
    """Gather / ScatterGather return DeferredTiles; convs, scatters and arbitrary
    torch ops downstream see exactly the eager values."""
    from sige_amd.nn import deferred

    torch.manual_seed(0)
    net = ResNet(6, 8).eval()
    blk = net.block
    blk.s1, blk.t1 = torch.randn(1, 6, 1, 1), torch.randn(1, 6, 1, 1)
    blk.s2, blk.t2 = torch.randn(1, 8, 1, 1), torch.randn(1, 8, 1, 1)
    orig = torch.randn(1, 6, 40, 44)
    mask = torch.zeros(40, 44, dtype=torch.bool)
    mask[10:19, 7:23] = True
    edited = orig + torch.randn_like(orig) * mask
    with torch.no_grad():
        net.set_mode("full")
        net(orig)
        net.set_mode("sparse")
        net.set_masks({(40, 44): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))})
        eager = net(edited)
        monkeypatch.setattr(deferred, "FORCE_ON_CPU", True)
        tiles = blk.main_gather(edited, blk.s1, blk.t1)
        assert isinstance(tiles, deferred.DeferredTiles) and tiles.spec["kind"] == "gather"
        lazy = net(edited)
        assert torch.equal(lazy, eager)
        # spatially varying affine or sparse_update: no deferral
        full_scale = torch.randn(1, 6, 40, 44)
        assert not isinstance(blk.main_gather(edited, full_scale, None), deferred.DeferredTiles)
        net.set_sparse_update(True)
        assert not isinstance(blk.main_gather(edited, blk.s1, blk.t1), deferred.DeferredTiles)


def test_mi355x_first_options_reduce_to_the_reference_form(cpu_oracle_backend):
    """The options that are not in the reference -- producer-side activation (`out_affine` +
    `cache_activated` / `preactivated`), conv -> scatter fusion (`forward_fused`), deferred `lazy_cat`,
    -- must give the plain module chain's values when no fused kernel applies (here: CPU tensors on the
    oracle backend); the `upsample2x` gather flag is refused there."""
    from sige_amd.nn import deferred

    torch.manual_seed(3)
    net = ResNet(8, 8).eval()
    blk = net.block
    blk.s1, blk.t1, blk.s2, blk.t2 = (torch.randn(1, 8, 1, 1) for _ in range(4))
    orig = torch.randn(1, 8, 32, 32)
    mask = torch.zeros(32, 32, dtype=torch.bool)
    mask[9:15, 5:19] = True
    edited = orig + torch.randn_like(orig) * mask
    with torch.no_grad():
        net.set_mode("full")
        net(orig)
        blk.scatter_gather.cache_activated(blk.s2, blk.t2)
        net.set_mode("sparse")
        net.set_masks({(32, 32): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))})
        want = net(edited)
        # the same block with every option on
        net.set_scatter_inplace(True)
        sc = blk.nin(blk.shortcut_gather(edited))
        h = blk.conv1(blk.main_gather(edited, blk.s1, blk.t1), out_affine=(blk.s2, blk.t2, "swish"))
        got = blk.scatter.forward_fused(blk.conv2, blk.scatter_gather(h, preactivated=True), sc)
        torch.testing.assert_close(got, want, rtol=0, atol=1e-5)
        # lazy cat: eager on CPU, and the Gather accepts either
        a, b = edited[:, :5], edited[:, 5:]
        cat = deferred.lazy_cat(a, b)
        assert not isinstance(cat, deferred.LazyCat) and torch.equal(cat, edited)
        # upsample2x gather == gather of the nearest-upsampled tensor
        conv = SIGEConv2d(8, 4, 3, 1, 1).eval()
        g = Gather(conv, 6)
        lo = torch.randn(1, 8, 16, 16)
        up = torch.nn.functional.interpolate(lo, scale_factor=2.0, mode="nearest")
        for m in (g, conv):
            m.set_mode("full")
        conv(g(up))
        g.set_mask({(32, 32): mask}, {}, 1)
        for m in (g, conv):
            m.set_mode("sparse")
        # off the fused GPU path the flag is refused (the model upsamples itself, as the reference's does)
        assert not g.fuses_upsample(lo)
        with pytest.raises(ValueError, match="upsample in the model"):
            g(lo, upsample2x=True)


def test_input_conv2d_and_plain_weight_off_gpu():
    """The first-layer helper is the plain conv anywhere but on a channels-last GPU image; a channels-last
    model's permuted weight is converted once per weight version."""
    from torch import nn

    from sige_amd.nn.dense import _plain_weight, input_conv2d

    torch.manual_seed(0)
    conv = nn.Conv2d(3, 8, 3, 1, 1)
    x = torch.randn(1, 3, 12, 10)
    with torch.no_grad():
        assert torch.equal(input_conv2d(conv, x), conv(x))
        xcl = x.contiguous(memory_format=torch.channels_last)
        assert torch.equal(input_conv2d(conv, xcl), conv(xcl))
    assert _plain_weight(conv) is conv.weight  # already dense: no copy
    conv = conv.to(memory_format=torch.channels_last)
    w1 = _plain_weight(conv)
    assert w1.is_contiguous() and torch.equal(w1, conv.weight) and _plain_weight(conv) is w1
    with torch.no_grad():
        conv.weight.mul_(2.0)
    w2 = _plain_weight(conv)
    assert w2 is not w1 and torch.equal(w2, conv.weight)


def test_packed_weights_keep_their_compute_tag():
    """A copy of f16-packed weights must still be dispatched to the f16 kernels (bench.py clones the operands of the calls it
    re-times; read as fp32-packed the buffer is half as long as the kernel expects)."""
    from sige_amd.hip import PackedWeights

    p = torch.zeros(8).as_subclass(PackedWeights)
    p.compute = "f16"
    for q in (p.clone(), p.detach(), p.contiguous(), p.to(torch.float32)):
        assert isinstance(q, PackedWeights) and q.compute == "f16"
    assert torch.zeros(8).as_subclass(PackedWeights).clone().compute == "f32"


def test_paired_convs_is_a_noop_off_the_gpu():
    """sige_amd.nn.paired_convs: the horizontal fusion of [shortcut conv, conv1] exists on the GPU only; on CPU tensors
    (and deferred cats of them) the context does nothing and never touches the native library."""
    from sige_amd.nn import paired_convs
    from sige_amd.nn.deferred import lazy_cat

    x = torch.randn(1, 4, 8, 8)
    with paired_convs(x) as ctx:
        assert ctx is None
    with paired_convs(lazy_cat(x, x), enabled=True) as ctx:
        assert ctx is None
    with paired_convs(x, enabled=False) as ctx:
        assert ctx is None


def test_conv1_twins_host_logic_on_the_oracle_backend():
    """Producer-side activation of the conv1 inputs (DDPMConfig.conv1_twins): registration on the first sparse forward,
    persistent twins across mask changes, invalidation by a new full pass -- with the twins EMULATED in torch where the CPU
    path cannot write them from a launch (scatter.EMULATE_TWINS), so that the whole state machine runs without a GPU.  The
    network with twins must equal the network without to fp32 rounding in every situation."""
    from oracle import oracle
    from sige_amd import runtime
    from sige_amd.nn import scatter
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet, ResBlock

    torch.manual_seed(0)
    cfg = DDPMConfig(ch=32, ch_mult=(1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(16,), resolution=64, sparse_threshold=32, groups=8)
    model = DDPMSparseUNet(cfg).eval()
    model.set_scatter_inplace(True)
    blocks = [m for m in model.modules() if isinstance(m, ResBlock)]
    g = torch.Generator().manual_seed(1)
    x0, noise, t = torch.randn(1, 3, 64, 64, generator=g), torch.randn(1, 3, 64, 64, generator=g), torch.zeros(1)

    def run(k, twins, original):
        m = torch.zeros(64, 64, dtype=torch.bool)
        m[10 + k:22 + k, 14:30 + k] = True
        for b in blocks:
            b.use_twins = twins
            b._drop_twin_links()
        model.set_masks(downsample_mask(dilate_mask(m, 5), 8))
        model.set_mode("sparse")
        for _ in range(3):  # (forward 1 registers, from forward 2 on the twins exist)
            out = model(original + noise * m, t)
        return out.clone()

    runtime.register_backend("cpu", oracle)
    scatter.EMULATE_TWINS = True
    try:
        with torch.no_grad():
            model.set_mode("full")
            model(x0, t)
            first = None
            for k in (0, 6, 3):
                ref, got = run(k, False, x0), run(k, True, x0)
                assert (ref - got).abs().max() < 1e-4
                assert sum(1 for b in blocks if b._twin_links) >= 12
                first = ref if first is None else first
            x0b = x0.flip(-1).contiguous()
            model.set_mode("full")
            model(x0b, t)
            ref, got = run(0, False, x0b), run(0, True, x0b)
            assert (ref - got).abs().max() < 1e-4
            assert (ref - first).abs().max() > 1e-2
            # a consumer that keeps an OLD twin after the producer's cache changed would show up here: make one stale on purpose
            stale = next(b for b in blocks if b._twin_links)
            key, prod = next(iter(stale._twin_links.items()))
            assert key[2] == stale._aff_gen  # (the link carries the generation of the affine it was made for)
            # one cache per denoising step (cache_id, sige/nn/scatter.py:59-60): two originals under ids 0 / 1, forwards
            # ALTERNATING between them -- a twin registered with cache 1's affine must never serve a forward under cache 0
            # (found on the GPU in round 3: the registration key did not carry the cache id)
            for cid, orig in ((0, x0), (1, x0b)):
                model.set_cache_id(cid)
                model.set_mode("full")
                model(orig, t)
            m = torch.zeros(64, 64, dtype=torch.bool)
            m[12:30, 20:40] = True
            model.set_masks(downsample_mask(dilate_mask(m, 5), 8))
            model.set_mode("sparse")
            want = {}
            for b in blocks:
                b.use_twins = False
            for cid, orig in ((0, x0), (1, x0b)):
                model.set_cache_id(cid)
                want[cid] = model(orig + noise * m, t).clone()
            for b in blocks:
                b.use_twins = True
            for cid in (1, 0, 1, 0, 0, 1):
                model.set_cache_id(cid)
                got = model((x0, x0b)[cid] + noise * m, t)
                assert (got - want[cid]).abs().max() < 1e-4, cid
            assert any(k[3] == 0 for b in blocks for k in b._twin_links) and any(k[3] == 1 for b in blocks for k in b._twin_links)
            model.set_cache_id(0)
    finally:
        scatter.EMULATE_TWINS = False
        runtime.unregister_backend("cpu")


def test_twins_survive_sparse_then_pack_then_new_cache():
    """activated twins are registered with VIEWS of the consumer's cached affine.  parallel.pack_caches moves
    every cached tensor -- the affines too -- into one flat buffer; a sparse forward that ran BEFORE the pack had left
    registrations pointing at the old affine tensors, and after a broadcast of another image's cache the producers kept writing
    SiLU(old_scale * x + old_shift).  Order under test: full(A) -> sparse x3 -> pack -> flat <- caches of B -> refresh -> sparse."""
    from oracle import oracle
    from sige_amd import parallel, runtime
    from sige_amd.nn import scatter
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet, ResBlock

    cfg = DDPMConfig(ch=32, ch_mult=(1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(16,), resolution=64, sparse_threshold=32, groups=8)
    g = torch.Generator().manual_seed(3)
    xa, xb, noise = (torch.randn(1, 3, 64, 64, generator=g) for _ in range(3))
    t = torch.zeros(1)
    m = torch.zeros(64, 64, dtype=torch.bool)
    m[18:34, 12:30] = True
    masks = downsample_mask(dilate_mask(m, 5), 8)

    def build():
        torch.manual_seed(0)
        net = DDPMSparseUNet(cfg).eval()
        net.set_scatter_inplace(True)
        return net

    runtime.register_backend("cpu", oracle)
    scatter.EMULATE_TWINS = True
    try:
        with torch.no_grad():
            # what rank 0 would send: the packed cache of image B
            src = build()
            src.set_mode("full")
            src(xb, t)
            flat_b = parallel.pack_caches(src).clone()
            src.set_masks(masks)
            src.set_mode("sparse")
            for b in src.modules():
                if isinstance(b, ResBlock):
                    b.use_twins = False
            want = src(xb + noise * m, t).clone()

            net = build()
            net.set_mode("full")
            net(xa, t)
            net.set_masks(masks)
            net.set_mode("sparse")
            for _ in range(3):
                net(xa + noise * m, t)
            blocks = [b for b in net.modules() if isinstance(b, ResBlock)]
            assert sum(1 for b in blocks if b._twin_links) >= 8
            flat = parallel.pack_caches(net)
            assert not any(b._twin_links for b in blocks)  # (the pack dropped them: the affines moved)
            flat.copy_(flat_b)  # (the broadcast)
            parallel.refresh_derived(net)
            for _ in range(3):
                got = net(xb + noise * m, t)
                assert (got - want).abs().max() < 1e-4
            assert sum(1 for b in blocks if b._twin_links) >= 8

            # the same protection without pack_caches: ANY re-pointing of a block's cached affine (here: fresh tensors with
            # other values) is noticed at the consumer's next forward -- the links carry the address of the affine they were
            # made with
            blk = next(b for b in blocks if b._twin_links)
            s1, t1, s2, t2 = blk.affine[0]
            blk.affine[0] = ((s1 * 1.5).contiguous(), (t1 + 0.25).contiguous(), s2, t2)
            for b in blocks:
                b.use_twins = False
            ref2 = net(xb + noise * m, t).clone()
            for b in blocks:
                b.use_twins = True
            for _ in range(3):
                assert (net(xb + noise * m, t) - ref2).abs().max() < 1e-4
    finally:
        scatter.EMULATE_TWINS = False
        runtime.unregister_backend("cpu")


def test_pipelined_refresh_waits_for_the_consumers_affine():
    """distribute_cache_pipelined refreshes a module as soon as the chunk holding its own caches has landed;
    a Scatter that keeps activated twins rebuilds them with the CONSUMER's affine, which sits later in the packed buffer.  The
    refresh schedule must place such a module no earlier than the chunk holding that affine."""
    from oracle import oracle
    from sige_amd import parallel, runtime
    from sige_amd.nn import scatter
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    cfg = DDPMConfig(ch=32, ch_mult=(1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(16,), resolution=64, sparse_threshold=32, groups=8)
    torch.manual_seed(0)
    net = DDPMSparseUNet(cfg).eval()
    net.set_scatter_inplace(True)
    g = torch.Generator().manual_seed(3)
    x, t = torch.randn(1, 3, 64, 64, generator=g), torch.zeros(1)
    m = torch.zeros(64, 64, dtype=torch.bool)
    m[18:34, 12:30] = True
    runtime.register_backend("cpu", oracle)
    scatter.EMULATE_TWINS = True
    try:
        with torch.no_grad():
            net.set_mode("full")
            net(x, t)
            flat = parallel.pack_caches(net)
            net.set_masks(downsample_mask(dilate_mask(m, 5), 8))
            net.set_mode("sparse")
            net(x, t)  # (consumers register with their producers, against the packed affines)
    finally:
        scatter.EMULATE_TWINS = False
        runtime.unregister_backend("cpu")
    _, layout = net.__dict__["_sige_cache_layout"]
    bounds, ready = parallel._refresh_schedule(net, layout, flat.numel(), 2, 8)
    chunk_of = {id(mod): k for k, mods in enumerate(ready) for mod in mods}
    own_end = {}
    for mod, _, e, _ in layout:
        own_end[id(mod)] = max(own_end.get(id(mod), 0), e)
    checked = later = 0
    for mod in net.modules():
        regs = getattr(getattr(mod, "twins", None), "regs", None)
        for key in (regs or ()):
            k_aff = next(i for i, (lo, hi) in enumerate(bounds) if own_end[key[0]] <= hi)
            assert chunk_of[id(mod)] >= k_aff
            k_own = next(i for i, (lo, hi) in enumerate(bounds) if own_end[id(mod)] <= hi)
            checked += 1
            later += chunk_of[id(mod)] > k_own
    assert checked >= 4 and later >= 1, (checked, later)  # (skip connections: some consumers' affines do sit in a later chunk)


def test_fp16_stored_cache_host_logic_on_the_oracle_backend():
    """SIGEModel.set_cache_dtype("f16"): the modules STORE their caches as fp16, every consumer reads them
    through `from_cache` off the GPU (on the GPU: the "_f16" / "_c16" kernels); pack_caches puts fp16 caches and fp32 affines
    into ONE fp16 buffer; the sparse output stays within the f16 criterion of the fp32-cache output and equals, to fp16 rounding of
    the activated copies, the fp32-cache model whose caches were rounded by hand."""
    from oracle import oracle
    from sige_amd import parallel, runtime, tolerance
    from sige_amd.nn import Scatter, ScatterGather, ScatterWithBlockResidual
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    cfg = DDPMConfig(ch=32, ch_mult=(1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(16,), resolution=64, sparse_threshold=32, groups=8)
    g = torch.Generator().manual_seed(7)
    x0, noise = torch.randn(1, 3, 64, 64, generator=g), torch.randn(1, 3, 64, 64, generator=g)
    t = torch.zeros(1)
    m = torch.zeros(64, 64, dtype=torch.bool)
    m[20:36, 10:34] = True
    masks = downsample_mask(dilate_mask(m, 5), 8)

    def build(cache_dtype):
        torch.manual_seed(0)
        net = DDPMSparseUNet(cfg).eval()
        net.set_scatter_inplace(True)
        net.set_cache_dtype(cache_dtype)
        net.set_mode("full")
        net(x0, t)
        return net

    def sparse(net):
        net.set_masks(masks)
        net.set_mode("sparse")
        net(x0 + noise * m, t)
        return net(x0 + noise * m, t).clone()

    runtime.register_backend("cpu", oracle)
    try:
        with torch.no_grad():
            ref = build("f32")
            want32 = sparse(ref)
            net = build("f16")
            caches = [c for mod in net.modules() if isinstance(mod, (Scatter, ScatterGather, ScatterWithBlockResidual))
                      for d in (mod.original_outputs, getattr(mod, "original_residuals", {}), getattr(mod, "activated_outputs", {}))
                      for c in d.values()]
            assert caches and all(c.dtype == torch.float16 for c in caches)
            got = sparse(net)
            chk = tolerance.f16_check(got, want32)
            assert chk["ok"] and 0.0 < chk["max_abs"] < 2e-2, chk
            # the fp32-cache model with its caches rounded by hand: the same values up to the second rounding of the activated copies
            for mod in ref.modules():
                for name in ("original_outputs", "original_residuals"):
                    for c in getattr(mod, name, {}).values():
                        c.copy_(c.half().float())
            parallel.refresh_derived(ref)
            by_hand = sparse(ref)
            assert float((got - by_hand).abs().max()) < 0.25 * max(chk["max_abs"], 1e-6) + 2e-4
            # one flat fp16 buffer for fp16 caches + fp32 affines; views alias it; results unchanged
            flat = parallel.pack_caches(net)
            assert flat.dtype == torch.float16
            n32 = sum(parallel._get(sl).numel() for sl in parallel.cache_slots(net) if parallel._get(sl).dtype == torch.float32)
            n16 = sum(parallel._get(sl).numel() for sl in parallel.cache_slots(net) if parallel._get(sl).dtype == torch.float16)
            assert n32 > 0 and n16 > 100 * n32 and flat.numel() >= n16 + 2 * n32
            assert torch.equal(sparse(net), got)
            before = parallel.checksum(flat)
            flat2 = flat.clone()
            flat.zero_()
            assert parallel.checksum(flat) == 0 != before
            flat.copy_(flat2)  # ("the broadcast")
            parallel.refresh_derived(net)
            assert parallel.checksum(flat) == before and torch.equal(sparse(net), got)
    finally:
        runtime.unregister_backend("cpu")


def test_stacked_edits_bookkeeping_off_the_gpu(cpu_oracle_backend):
    """sige_amd/stacked.py: the tall-image views, the stacked mask pyramid, stacking / unstacking of a model's caches (values
    repeated E times along H, Gather / Scatter resolutions scaled, everything restored afterwards).  The seam semantics live in
    the HIP kernels (a -m gpu test); off the GPU the library refuses the mode rather than bleeding across seams."""
    from sige_amd import stacked
    from sige_amd.nn import Gather, Scatter, ScatterGather, ScatterWithBlockResidual

    x = torch.randn(3, 8, 4, 6).contiguous(memory_format=torch.channels_last)
    tl = stacked.tall(x)
    assert tuple(tl.shape) == (1, 8, 12, 6) and tl.data_ptr() == x.data_ptr() and torch.equal(tl[0, :, 4:8], x[1])
    back = stacked.untall(tl, 3)
    assert torch.equal(back, x) and back.data_ptr() == x.data_ptr()
    pyr = [{(8, 8): torch.rand(8, 8) > 0.5, (4, 4): torch.rand(4, 4) > 0.5} for _ in range(3)]
    st = stacked.stack_masks(pyr)
    assert set(st) == {(24, 8), (12, 4)} and torch.equal(st[(24, 8)][8:16], pyr[1][(8, 8)])

    torch.manual_seed(0)
    net = ResNet(8, 12).eval()
    blk = net.block
    blk.s1, blk.t1 = torch.randn(1, 8, 1, 1), torch.randn(1, 8, 1, 1)
    blk.s2, blk.t2 = torch.randn(1, 12, 1, 1), torch.randn(1, 12, 1, 1)
    orig = torch.randn(1, 8, 32, 32)
    with torch.no_grad():
        net.set_mode("full")
        net(orig)
    before = {(id(m), n, k): v.clone() for m in net.modules() for n in ("original_outputs", "original_residuals")
              for k, v in getattr(m, n, {}).items()}
    res_before = {id(m): tuple(m.input_res) for m in net.modules() if isinstance(m, Gather)}
    stacked.stack_caches(net, 3)
    for m in net.modules():
        if isinstance(m, (Scatter, ScatterGather, ScatterWithBlockResidual)):
            for n in ("original_outputs", "original_residuals"):
                for k, v in getattr(m, n, {}).items():
                    assert v.shape[2] == 96 and torch.equal(v[:, :, 32:64], before[(id(m), n, k)])
        if isinstance(m, Gather):
            assert tuple(m.input_res) == (3 * res_before[id(m)][0], res_before[id(m)][1])
    with pytest.raises(RuntimeError, match="stacked already"):
        stacked.stack_caches(net, 2)
    stacked.unstack_caches(net)
    for m in net.modules():
        for n in ("original_outputs", "original_residuals"):
            for k, v in getattr(m, n, {}).items():
                assert torch.equal(v, before[(id(m), n, k)])
        if isinstance(m, Gather):
            assert tuple(m.input_res) == res_before[id(m)]


def test_twin_buffers_follow_cache_and_masks():
    """scatter._TwinBuffers (the persistent activated twins of a Scatter module's in-place output): built from the cache,
    rebuilt when the mask stamp or the cache generation changes, refreshed in place when the cache is rewritten in place,
    at most two registrations."""
    from sige_amd.nn.scatter import _TwinBuffers

    tb = _TwinBuffers()
    cache = torch.randn(1, 8, 4, 4).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.randn(8), torch.randn(8)
    want = lambda c: torch.nn.functional.silu(c * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))  # noqa: E731
    assert tb.register("a", sc, sh) and tb.register("b", sc * 2, sh) and not tb.register("c", sc, sh)
    (ka, ba, _, _), (kb, bb, _, _) = tb.launch_args(0, cache, stamp=1)
    assert (ka, kb) == ("a", "b")
    torch.testing.assert_close(ba, want(cache))
    assert tb.launch_args(0, cache, stamp=1)[0][1] is ba                      # same mask, same cache: the same buffer
    ba.add_(1.0)                                                              # (a launch wrote tiles into it)
    b2 = tb.launch_args(0, cache, stamp=2)[0][1]                              # new masks: rebuilt from the cache, IN PLACE
    assert b2 is ba                                                           # (a launch plan / graph may hold its address)
    torch.testing.assert_close(b2, want(cache))
    tb.invalidate(0)                                                          # the cache was replaced (full pass)
    cache2 = torch.randn_like(cache)
    b3 = tb.launch_args(0, cache2, stamp=2)[0][1]
    torch.testing.assert_close(b3, want(cache2))
    cache2.mul_(0.5)                                                          # rewritten in place (a collective): same address
    tb.refresh({0: cache2})
    assert tb.launch_args(0, cache2, stamp=2)[0][1] is b3
    torch.testing.assert_close(b3, want(cache2))
    tb.unregister("a")
    assert [k for k, *_ in tb.launch_args(0, cache2, stamp=2)] == ["b"] and tb.register("c", sc, sh)


def test_full_conv2d_off_the_gpu_is_the_torch_expression():
    """dense.full_conv2d (the full pass's conv) anywhere but on a channels-last GPU tensor with a non-fp32 compute dtype is the
    reference's own torch expression -- also for the options the library's full pass fuses into the launch (second half of a
    cat, nearest x2 upsampling, residual); `stats` is then simply not produced."""
    from torch.nn import functional as F

    from sige_amd import hip
    from sige_amd.nn.dense import fast_full_pass, full_conv2d, group_norm_affine

    torch.manual_seed(3)
    conv = nn.Conv2d(12, 8, 3, 1, 1)
    x, x2 = torch.randn(1, 8, 10, 6), torch.randn(1, 4, 10, 6)
    s, t = torch.randn(1, 12, 1, 1), torch.randn(1, 12, 1, 1)
    res = torch.randn(1, 8, 20, 12)
    assert not fast_full_pass(conv)
    with torch.no_grad():
        got = full_conv2d(conv, x, s, t, "swish", residual=res, x2=x2, upsample2x=True, stats=True)
        h = F.interpolate(torch.cat([x, x2], 1), scale_factor=2.0, mode="nearest")
        want = conv(F.silu(h * s + t)) + res
    assert torch.equal(got, want)
    assert hip.channel_stats(got) is None
    # the GroupNorm of a cat / of x + bias without statistics: the same affine as torch's GroupNorm
    norm = nn.GroupNorm(4, 12)
    with torch.no_grad():
        norm.weight.normal_()
        norm.bias.normal_()
        cb = torch.randn(12)
        sc, sh = group_norm_affine(x, norm, cb, x2=x2, make_stats=True)
        both = torch.cat([x, x2], 1)
        torch.testing.assert_close(both * sc + sh, norm(both + cb.view(1, -1, 1, 1)), rtol=1e-5, atol=1e-5)


def test_block_residual_scatter_takes_a_precomputed_sum(cpu_oracle_backend):
    """ScatterWithBlockResidual.forward(x, residual, x_is_sum=True) (full mode: the conv's epilogue already added the residual)
    caches exactly what forward(main, residual) caches (sige/nn/scatter.py:89-93), and is refused outside full mode."""
    from sige_amd.nn import Gather, ScatterWithBlockResidual, SIGEConv2d

    torch.manual_seed(4)
    conv, short = SIGEConv2d(4, 4, 3, 1, 1), SIGEConv2d(4, 4, 1, 1, 0)
    g0, g1 = Gather(conv, 6), Gather(short, 4)
    a, b = ScatterWithBlockResidual(g0, g1), ScatterWithBlockResidual(g0, g1)
    main, resid = torch.randn(1, 4, 16, 16), torch.randn(1, 4, 16, 16)
    for m in (g0, g1, a, b):
        m.set_mode("full")
    out_a = a(main, resid)
    out_b = b(main + resid, resid, x_is_sum=True)
    assert torch.equal(out_a, out_b)
    assert torch.equal(a.original_outputs[0], b.original_outputs[0]) and torch.equal(a.original_residuals[0], b.original_residuals[0])
    b.set_mode("sparse")
    with pytest.raises(AssertionError):
        b(main, resid, x_is_sum=True)


def test_gaugan_stacked_mode_refuses_off_the_gpu():
    """The SPADE generator's stacked mode exists on the all-library GPU path only (every op seam-aware, per pixel or tile-local):
    anywhere else -- CPU tensors, the torch module chain -- the forward raises instead of running ops that would read across the
    seams of the tall image."""
    from sige_amd.workloads.gaugan_spade import SPADEConfig, SpadeGenerator

    model = SpadeGenerator(SPADEConfig(ngf=8, crop_size=64)).eval()
    seg = torch.zeros(2, 36, 32, 64)
    model.set_mode("sparse")
    for m in model.modules():
        if hasattr(m, "edit_batch"):
            m.edit_batch = 2
    with pytest.raises(RuntimeError, match="stacked edits"):
        model(seg)
    model.cfg.fused = False
    with pytest.raises(RuntimeError, match="stacked edits"):
        model(seg)
