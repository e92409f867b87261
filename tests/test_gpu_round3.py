"""Round 3, on the MI355X: the dense-layer conv on the fp16 matrix cores (csrc/conv_wide.hpp) -- fp16 operands and split
fp16 operands ("f16x3": fp32-level results) -- against fp64 torch; the compute-dtype policy of BASELINE.json configs[4]
over its whole edit-ratio sweep with the ONE stated criterion (sige_amd.tolerance); the full pass on those kernels;
multi-step caches (cache_id > 0); a model on a non-current device."""
import os

import pytest
import torch
from torch import nn
from torch.nn import functional as F

from oracle import oracle
from sige_amd import tolerance
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from sige_amd import hip as h

    h.lib()
    return h


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _ref64(x, x2, s, t, act, w, b, residual, oaff, up):
    h = x if x2 is None else torch.cat([x, x2], 1)
    if up:
        h = F.interpolate(h, scale_factor=2.0, mode="nearest")
    h = h.double()
    if s is not None:
        h = h * s.double() + t.double()
        if act == "swish":
            h = F.silu(h)
    out = F.conv2d(h, w.double(), None if b is None else b.double(), 1, w.shape[2] // 2)
    if residual is not None:
        out = out + residual.double()
    raw = out
    if oaff is not None:
        out = out * oaff[0].double().view(1, -1, 1, 1) + oaff[1].double().view(1, -1, 1, 1)
        if oaff[2] == "swish":
            out = F.silu(out)
    return out, raw


WIDE_CASES = [
    # k, C1, C2, Cout, H, W, B, affine, act, residual, out_affine, up
    (3, 256, 0, 256, 32, 32, 1, True, "swish", True, False, False),     # 32x32 dense block conv (K split 2)
    (3, 512, 256, 256, 32, 32, 1, True, "swish", False, True, False),   # fused cat, epilogue affine + SiLU
    (3, 1024, 0, 512, 16, 16, 1, True, "swish", True, False, False),    # deep K, 4 pixel blocks (K split 16)
    (3, 512, 512, 512, 8, 8, 1, False, "identity", True, False, False),  # one pixel block, raw staging, cat
    (1, 768, 0, 256, 32, 32, 1, False, "identity", False, False, False),  # shortcut 1x1
    (1, 512, 0, 1536, 16, 16, 1, True, "identity", False, False, False),  # attention qkv: affine without activation
    (1, 256, 256, 512, 8, 8, 2, True, "swish", True, True, False),      # batch 2, per-batch affine
    (3, 128, 0, 128, 40, 24, 1, True, "swish", True, False, False),     # ragged: 24 = 3 patches, 40 = 5; no K split
    (3, 64, 0, 64, 12, 20, 2, False, "identity", False, False, False),  # H, W not multiples of 8 (masked stores)
    (3, 128, 0, 64, 32, 32, 1, True, "swish", False, False, True),      # nearest x2 upsampling fused into the addressing
    (3, 128, 0, 128, 128, 128, 1, True, "swish", False, False, False),  # a full-pass layer: 512 workgroups, no split
]


@pytest.mark.parametrize("compute", ["f16x3", "f16", "f32"])
@pytest.mark.parametrize("k,c1,c2,cout,H,W,B,aff,act,res,oaff,up", WIDE_CASES)
def test_wide_conv_vs_fp64(hip, compute, k, c1, c2, cout, H, W, B, aff, act, res, oaff, up):
    """One launch of the dense-layer kernel against the same expression in fp64 torch.
    f16x3: the result is fp32-level -- max |d| <= 2e-5 * (1 + max |ref|), three orders inside the 1e-3 of the fp32 path;
    f16: exactly the fp64 conv of the fp16-ROUNDED operands (products of two fp16 values are exact in fp32) to fp32
         summation accuracy, and the exact result within the stated f16 criterion."""
    g = torch.Generator().manual_seed(1000 * k + c1 + cout + H)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    cin = c1 + c2
    hs, ws = (H // 2, W // 2) if up else (H, W)
    x, x2 = _cl(r(B, c1, hs, ws)), (_cl(r(B, c2, hs, ws)) if c2 else None)
    w, b = r(cout, cin, k, k) / (k * cin ** 0.5), r(cout)
    nb = B if (aff and B > 1) else 1
    s, t = (r(nb, cin, 1, 1), r(nb, cin, 1, 1)) if aff else (None, None)
    residual = _cl(r(B, cout, H, W)) if res else None
    oa = (r(cout), r(cout), "swish") if oaff else None
    packed = hip.wide_conv_pack_weights(w, compute)
    assert packed is not None and packed.compute == compute + "w"
    got = hip.wide_conv_cl(x, x2, s, t, act, packed, b, cout, (k, k), residual=residual, out_affine=oa, upsample2x=up)
    assert got is not None and hip.is_cl(got) and tuple(got.shape) == (B, cout, H, W)
    want, _ = _ref64(x, x2, s, t, act, w, b, residual, oa, up)
    if compute in ("f16x3", "f32"):  # (f32 = exact products on v_mfma_f32_32x32x2_f32; SiLU in the staging path is v_exp / v_rcp)
        err = float((got.double() - want).abs().max())
        assert err <= 2e-5 * (1.0 + float(want.abs().max())), err
    else:
        assert tolerance.f16_check(got, want.float())["ok"]
        # exact-product check: round the operands as the kernel does and convolve in fp64
        h = x if x2 is None else torch.cat([x, x2], 1)
        if up:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
        if s is not None:
            h = h * s + t  # (two separately rounded fp32 ops, as in the kernel)
            if act == "swish":
                h = F.silu(h)
        out = F.conv2d(h.half().double(), w.half().double(), b.double(), 1, k // 2)
        if residual is not None:
            out = out + residual.double()
        if oa is not None:
            out = F.silu(out * oa[0].double().view(1, -1, 1, 1) + oa[1].double().view(1, -1, 1, 1))
        # (swish_fast in the staging path is <= 1e-6 relative; a value next to an fp16 rounding boundary may round the other way)
        bad = (got.double() - out).abs() > 3e-4 * (1.0 + out.abs())
        assert float(bad.double().mean()) < 1e-3


def test_wide_conv_split_k_is_deterministic_and_equals_unsplit(hip, tuning):
    """The in-launch K-split finish: same bits on every run (fixed summation order), fp32-close to the unsplit launch, and no
    stale partial sum when two inputs alternate over the same workspace memory -- eagerly and in graph replays."""
    g = torch.Generator().manual_seed(7)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    xa, xb = _cl(r(1, 1024, 16, 16)), _cl(r(1, 1024, 16, 16))
    w, b = r(512, 1024, 3, 3) / 96, r(512)
    packed = hip.wide_conv_pack_weights(w, "f16x3")
    run = lambda x: hip.wide_conv_cl(x, None, None, None, "identity", packed, b, 512, (3, 3))  # noqa: E731
    a1, b1, a2, b2 = run(xa).clone(), run(xb).clone(), run(xa).clone(), run(xb).clone()
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and not torch.equal(a1, b1)
    hip.wide_conv_force_ksplit(1)
    try:
        a0 = run(xa).clone()
    finally:
        hip.wide_conv_force_ksplit(0)
    torch.testing.assert_close(a1, a0, rtol=0, atol=2e-5)
    for ks in (2, 5, 16):
        hip.wide_conv_force_ksplit(ks)
        try:
            torch.testing.assert_close(run(xa), a0, rtol=0, atol=2e-5)
        finally:
            hip.wide_conv_force_ksplit(0)
    xin = xa.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(xin)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            out = run(xin)
    torch.cuda.current_stream().wait_stream(s)
    for i in range(6):
        xin.copy_(xa if i % 2 == 0 else xb)
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, a1 if i % 2 == 0 else b1)


def test_wide_conv_small_operands_keep_their_precision(hip):
    """Split operands at small magnitudes: weights of 1e-4 (their lo parts would be fp16 subnormals without the power-of-two
    pre-scaling at pack time) and activations around 1e-2 still give fp32-level results."""
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    x = _cl(r(1, 256, 16, 16) * 1e-2)
    w = r(64, 256, 3, 3) * 1e-4
    packed = hip.wide_conv_pack_weights(w, "f16x3")
    assert packed.wshift > 20
    got = hip.wide_conv_cl(x, None, None, None, "identity", packed, None, 64, (3, 3))
    want = F.conv2d(x.double(), w.double(), None, 1, 1)
    rel = float((got.double() - want).abs().max() / want.abs().max())
    assert rel < 2e-5, rel


def test_wide_conv_twins_and_dense_module_routing(hip):
    """fused_conv2d on a conv whose compute dtype is "f16x3" runs the dense-layer kernel (one launch), with twins."""
    from sige_amd.nn.dense import fused_conv2d

    torch.manual_seed(5)
    conv = nn.Conv2d(768, 256, 3, 1, 1).to(DEV)
    conv.compute_dtype = "f16x3"
    x, x2 = _cl(torch.randn(1, 512, 64, 64, device=DEV)), _cl(torch.randn(1, 256, 64, 64, device=DEV))  # (14.5 GFLOP: above WIDE_MIN_FLOP)
    s, t = torch.randn(1, 768, 1, 1, device=DEV), torch.randn(1, 768, 1, 1, device=DEV)
    res = _cl(torch.randn(1, 256, 64, 64, device=DEV))
    tw = {"a": (torch.randn(256, device=DEV), torch.randn(256, device=DEV))}
    with torch.no_grad():
        n0 = hip.launch_count()
        got = fused_conv2d(conv, x, s, t, "swish", x2=x2, residual=res, twins=tw)
        assert hip.launch_count() - n0 == 2  # (weight packing + ONE conv launch)
        want = conv(F.silu(torch.cat([x, x2], 1) * s + t)) + res
        torch.testing.assert_close(got, want, rtol=0, atol=2e-4)
        twin = got._sige_twins["a"]
        torch.testing.assert_close(twin, F.silu(want * tw["a"][0].view(1, -1, 1, 1) + tw["a"][1].view(1, -1, 1, 1)), rtol=0, atol=3e-4)


# ---- the whole DDPM-256 U-Net under the three compute dtypes -------------------------------------------------------------------
@pytest.fixture(scope="module")
def ddpm_reference():
    """The fp32 reference: the DDPM-256 U-Net on the CPU oracle backend (sige/cpu restated + torch CPU convs), full pass
    on the original, sparse outputs at the five edit ratios of BASELINE.json configs[4]."""
    import bench
    from sige_amd import runtime
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval()
    x0, noise = bench.make_inputs()
    t = torch.zeros(1)
    n = min(32, os.cpu_count() or 1)
    torch.set_num_threads(n)
    oracle.set_num_threads(n)
    runtime.register_backend("cpu", oracle)
    outs = {}
    try:
        with torch.no_grad():
            model.set_mode("full")
            full = model(x0, t)
            for r in (0.01, 0.02, 0.05, 0.1, 0.2):
                mask = bench.edit_mask(r)
                model.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
                model.set_mode("sparse")
                outs[r] = model(x0 + noise * mask, t).clone()
    finally:
        runtime.unregister_backend("cpu")
    model.clear_cache()
    return {"state": model.state_dict(), "x0": x0, "noise": noise, "full": full, "sparse": outs}


@pytest.fixture(scope="module")
def ddpm_gpu(ddpm_reference):
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    model = DDPMSparseUNet(DDPMConfig()).eval()
    model.load_state_dict(ddpm_reference["state"])
    model = model.to(DEV).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    return model


def _gpu_sparse(model, ref, ratio, full_dtype="f32"):
    import bench
    from sige_amd.utils import dilate_mask, downsample_mask

    cl = lambda a: a.to(DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    t = torch.zeros(1, device=DEV)
    mask = bench.edit_mask(ratio)
    with torch.no_grad():
        model.set_masks(downsample_mask(dilate_mask(mask.to(DEV), 5), 8))
        model.set_mode("sparse")
        x1 = cl(ref["x0"] + ref["noise"] * mask)
        model(x1, t)  # (consumers register their activated twins on the first forward)
        return model(x1, t).clone()


@pytest.mark.parametrize("ratio", [0.01, 0.02, 0.05, 0.1, 0.2])
def test_f16_policy_ddpm_forward_whole_sweep(hip, ddpm_reference, ddpm_gpu, ratio):
    """BASELINE.json configs[4] at its own size over its WHOLE sweep, ONE criterion (sige_amd.tolerance.F16_CRITERION):
    set_compute_dtype("f16") = fp16 operands everywhere except the model's F16_KEEP convs, which run split fp16 operands."""
    ref = ddpm_reference
    model = ddpm_gpu
    cl = lambda a: a.to(DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    with torch.no_grad():
        model.set_compute_dtype("f32")
        model.set_mode("full")
        model(cl(ref["x0"]), torch.zeros(1, device=DEV))  # the original's cache: the fp32 full pass
        model.set_compute_dtype("f16", edit_ratio=ratio)
        assert model.compute_policy["keep"] == (model.F16_KEEP if ratio > model.F16_KEEP_ABOVE else ())
        out = _gpu_sparse(model, ref, ratio)
        model.set_compute_dtype("f32")
    chk = tolerance.f16_check(out, ref["sparse"][ratio])
    assert chk["ok"], chk


@pytest.mark.parametrize("ratio", [0.01, 0.02, 0.05, 0.1, 0.2])
def test_f16_cache_ddpm_forward_whole_sweep(hip, ddpm_reference, ddpm_gpu, ratio):
    """Round 4: the same sweep with the cache STORED as fp16 -- SIGEModel.set_cache_dtype("f16"):
    every Scatter / ScatterGather / ScatterWithBlockResidual cache and the activated copies are fp16 tensors, read by the "_c16"
    conv kernels and the fp16 refresh; no fp32 copy of a cache exists -- under the same ONE criterion against the fp32 CPU
    reference, and the resident cache is half the bytes."""
    from sige_amd import parallel
    from sige_amd.nn import Scatter, ScatterGather, ScatterWithBlockResidual

    ref = ddpm_reference
    model = ddpm_gpu
    cl = lambda a: a.to(DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    try:
        with torch.no_grad():
            model.set_compute_dtype("f32")
            model.set_cache_dtype("f16")
            model.set_mode("full")
            model(cl(ref["x0"]), torch.zeros(1, device=DEV))
            caches = [c for mod in model.modules() if isinstance(mod, (Scatter, ScatterGather, ScatterWithBlockResidual))
                      for d in (mod.original_outputs, getattr(mod, "original_residuals", {}), getattr(mod, "activated_outputs", {}))
                      for c in d.values()]
            assert len(caches) > 40 and all(c.dtype == torch.float16 for c in caches)
            cache_bytes = sum(parallel._get(sl).numel() * parallel._get(sl).element_size() for sl in parallel.cache_slots(model))
            assert 3.3e8 < cache_bytes < 3.5e8, cache_bytes  # (672.9 MB in fp32: SURVEY.md 8e)
            model.set_compute_dtype("f16", edit_ratio=ratio)
            out = _gpu_sparse(model, ref, ratio)
            import bench

            x1 = cl(ref["x0"] + ref["noise"] * bench.edit_mask(ratio))
            n0 = hip.launch_count()
            again = model(x1, torch.zeros(1, device=DEV))
            # (a steady-state forward: no conversion pass in front of or behind any launch.  102 launches; since round 6 a conv1 that the
            #  router sends to the fp16 tile conv v3 launches its held 1x1 shortcut on its own: at most the 9 paired blocks of the up path)
            assert hip.launch_count() - n0 <= 102 + 9
            assert torch.equal(again, out)
    finally:
        model.set_compute_dtype("f32")
        model.set_cache_dtype("f32")
        model.clear_cache()
    chk = tolerance.f16_check(out, ref["sparse"][ratio])
    assert chk["ok"], chk


@pytest.mark.parametrize("ratio", [0.01, 0.2])
def test_f16x3_ddpm_forward_meets_the_fp32_tolerance(hip, ddpm_reference, ddpm_gpu, ratio):
    """Split fp16 operands ("f16x3": dense remainder AND the cache-producing full pass on the fp16 matrix cores, tile convs
    exact fp32) against the fp32 CPU reference with the FP32 path's tolerance, 1e-3 abs."""
    ref = ddpm_reference
    model = ddpm_gpu
    cl = lambda a: a.to(DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    with torch.no_grad():
        model.set_compute_dtype("f16x3")
        model.set_mode("full")
        n0 = hip.launch_count()
        full = model(cl(ref["x0"]), torch.zeros(1, device=DEV)).clone()
        assert hip.launch_count() - n0 >= 60  # the full pass ran on the library's kernels, not on MIOpen
        out = _gpu_sparse(model, ref, ratio)
        model.set_compute_dtype("f32")
    torch.testing.assert_close(full.cpu(), ref["full"], rtol=0, atol=util.CONV_ATOL)
    torch.testing.assert_close(out.cpu(), ref["sparse"][ratio], rtol=0, atol=util.CONV_ATOL)


def test_multi_step_caches_cache_id(hip, ddpm_reference, ddpm_gpu):
    """One cache per denoising step (sige/nn/scatter.py:59-60, diffusion_demo/runner.py:134-164): two different
    originals under cache_id 0 / 1, the same edit against each; every (cache_id, edit) pair equals the CPU reference."""
    import bench
    from sige_amd import runtime
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    ref = ddpm_reference
    x0 = [ref["x0"], ref["x0"].flip(3) * 0.9]
    mask = bench.edit_mask(0.05)
    t = torch.zeros(1)
    cpu = DDPMSparseUNet(DDPMConfig()).eval()
    cpu.load_state_dict(ref["state"])
    runtime.register_backend("cpu", oracle)
    want = {}
    try:
        with torch.no_grad():
            for cid in (0, 1):
                cpu.set_cache_id(cid)
                cpu.set_mode("full")
                cpu(x0[cid], t)
            cpu.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
            cpu.set_mode("sparse")
            for cid in (1, 0, 1):
                cpu.set_cache_id(cid)
                want[cid] = cpu(x0[cid] + ref["noise"] * mask, t).clone()
    finally:
        runtime.unregister_backend("cpu")
    model = ddpm_gpu
    cl = lambda a: a.to(DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    td = torch.zeros(1, device=DEV)
    with torch.no_grad():
        model.set_compute_dtype("f32")
        for cid in (0, 1):
            model.set_cache_id(cid)
            model.set_mode("full")
            model(cl(x0[cid]), td)
        model.set_masks(downsample_mask(dilate_mask(mask.to(DEV), 5), 8))
        model.set_mode("sparse")
        for cid in (1, 0, 1, 0):
            model.set_cache_id(cid)
            got = model(cl(x0[cid] + ref["noise"] * mask), td)
            torch.testing.assert_close(got.cpu(), want[cid], rtol=0, atol=util.CONV_ATOL)
        model.set_cache_id(0)


def test_model_on_a_non_current_device(hip):
    """a paired / unpaired conv of a model that lives on a device other than HIP's current one launches on the
    model's device (skipped on a one-GPU box)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    dev1 = torch.device("cuda", 1)
    x = _cl(torch.randn(1, 64, 32, 32, device=dev1))
    w = torch.randn(64, 64, 3, 3, device=dev1) / 24
    idx = hip.all_tiles(32, 32, (4, 4), (1, 1), (1, 1), dev1)
    p = hip.conv_pack_weights(w, 6, 6, (1, 1))
    torch.cuda.set_device(0)
    with hip.conv_pair(x):
        out = hip.gather_conv_cl(x, None, (6, 6), idx, None, None, "identity", p, None, 64, (3, 3), (1, 1),
                                 full=dict(offset=(1, 1), out_res=(32, 32), residual=None))
    torch.cuda.synchronize(dev1)
    torch.testing.assert_close(out, F.conv2d(x, w, None, 1, 1), rtol=0, atol=2e-4)


def test_sd_unet_at_its_own_size_vs_cpu_oracle(hip):
    """BASELINE.json configs[3] AT ITS OWN SIZE (model_channels 320: the 860 M-parameter SD v1 U-Net, latent [2,4,64,64], text
    context [2,77,768], 15 % edit): channels-last + in-place buffers on the GPU against the same network on the CPU oracle
    backend, 1e-3.  (The committed fixture pins the structure at model_channels 128 to the reference's own model class;
    this test carries that parity to the size the benchmark runs.)"""
    from sige_amd import runtime
    from sige_amd.utils import downsample_mask
    from sige_amd.workloads.sd_unet import SDConfig, SDUNet
    from tests.golden.model_init import init_by_name

    model = SDUNet(SDConfig()).eval()
    init_by_name(model)
    g = torch.Generator().manual_seed(4)
    x0, noise = torch.randn(2, 4, 64, 64, generator=g), torch.randn(2, 4, 64, 64, generator=g)
    ctx, ts = torch.randn(2, 77, 768, generator=g), torch.full((2,), 500.0)
    mask512 = torch.zeros(512, 512, dtype=torch.bool)
    mask512[150:348, 120:318] = True
    masks = downsample_mask(mask512, min_res=8, dilation=1)
    x1 = x0 + noise * masks[(64, 64)]
    n = min(32, os.cpu_count() or 1)
    torch.set_num_threads(n)
    oracle.set_num_threads(n)
    runtime.register_backend("cpu", oracle)
    try:
        with torch.no_grad():
            model.set_mode("full")
            full_c = model(x0, ts, context=ctx)
            model.set_masks(masks)
            model.set_mode("sparse")
            sparse_c = model(x1, ts, context=ctx)
    finally:
        runtime.unregister_backend("cpu")
    model.clear_cache()
    model = model.to(DEV).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    cl = lambda a: a.to(DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    with torch.no_grad():
        model.set_mode("full")
        full_g = model(cl(x0), ts.to(DEV), context=ctx.to(DEV))
        model.set_masks(downsample_mask(mask512.to(DEV), min_res=8, dilation=1))
        model.set_mode("sparse")
        sparse_g = model(cl(x1), ts.to(DEV), context=ctx.to(DEV))
    assert float(sparse_c.abs().max()) > 1e-2 and float((sparse_c - full_c).abs().max()) > 1e-3
    torch.testing.assert_close(full_g.cpu(), full_c, rtol=0, atol=util.CONV_ATOL)
    torch.testing.assert_close(sparse_g.cpu(), sparse_c, rtol=0, atol=util.CONV_ATOL)


# ---- split fp16 operands in the TILE kernels (ConvGeoX) -------------------------------------------------------------------------
@pytest.mark.parametrize("k,cin,cout,T,mt", [(3, 128, 128, 124, 0), (3, 256, 128, 40, 16), (3, 64, 256, 7, 32), (3, 192, 64, 33, 16),
                                              (1, 256, 128, 56, 0), (1, 128, 256, 9, 32), (1, 384, 128, 30, 16), (3, 36, 64, 5, 0)])
@pytest.mark.parametrize("wmag", [1.0, 1e-4])
def test_f16x3_tile_conv_is_fp32_level(hip, k, cin, cout, T, mt, wmag, tuning):
    """The stacked-block conv on split fp16 operands against an fp64 conv of the EXACT operands: max |d| <= 2e-5 * (1 + max |ref|)
    -- also with weights of 1e-4 (the device-side power-of-two pre-scaling keeps their lo parts normal fp16 numbers)."""
    torch.manual_seed(T + cin)
    R = 6 if k == 3 else 4
    x = _cl(torch.randn(T, cin, R, R, device=DEV))
    w = torch.randn(cout, cin, k, k, device=DEV) / (k * cin ** 0.5) * wmag
    b = torch.randn(cout, device=DEV) * wmag
    packed = hip.conv_pack_weights(w, R, R, (1, 1), "f16x3")
    assert packed.compute == "f16x3"
    hip.conv_force_tile(mt, 1 if mt else 0)
    try:
        got = hip.block_conv_cl(x, packed, b, cout, (k, k), (1, 1))
    finally:
        hip.conv_force_tile(0, 0)
    want = F.conv2d(x.double(), w.double(), b.double())
    err = float((got.double() - want).abs().max())
    assert err <= 2e-5 * (wmag + float(want.abs().max())), (err, float(want.abs().max()))


@pytest.mark.parametrize("act", ["swish", "identity"])
def test_f16x3_fused_gather_scatter_gather_and_pair(hip, act):
    """The fused forms (gather -> conv to tiles / into a full tensor with residual and epilogue affine, scatter_gather -> conv ->
    scatter with block residual, shortcut + conv1 in one launch) on split fp16 operands equal the exact-fp32 kernels to 5e-5."""
    from sige_amd.utils import reduce_mask

    torch.manual_seed(3)
    C, Co, H = 128, 128, 64
    mask = torch.zeros(H, H, dtype=torch.bool, device=DEV)
    mask[10:40, 20:50] = True
    mask[0, 0] = True
    idx = reduce_mask(mask, (6, 6), (4, 4), (1, 1))
    idx4 = reduce_mask(mask, (4, 4), (4, 4), (0, 0))
    x, y = _cl(torch.randn(1, C, H, H, device=DEV)), _cl(torch.randn(1, C, H, H, device=DEV))
    w, b = torch.randn(Co, C, 3, 3, device=DEV) / 34, torch.randn(Co, device=DEV)
    w1, b1 = torch.randn(Co, C, 1, 1, device=DEV) / 11, torch.randn(Co, device=DEV)
    sc_, sh_ = (torch.randn(1, C, 1, 1, device=DEV), torch.randn(1, C, 1, 1, device=DEV)) if act == "swish" else (None, None)
    oa = (torch.randn(Co, device=DEV), torch.randn(Co, device=DEV), "swish")
    smap = hip.get_scatter_map(H, H, 6, 6, 3, 3, 1, 1, 1, 1, idx)
    t4 = _cl(torch.randn(idx.shape[0], C, 4, 4, device=DEV))
    res = _cl(torch.randn(1, Co, H, H, device=DEV))
    full = dict(offset=(1, 1), out_res=(H, H), residual=res)

    def run(compute):
        p3, p1 = hip.conv_pack_weights(w, 6, 6, (1, 1), compute), hip.conv_pack_weights(w1, 4, 4, (1, 1), compute)
        out = {}
        out["tiles"] = hip.gather_conv_cl(x, None, (6, 6), idx, sc_, sh_, act, p3, b, Co, (3, 3), (1, 1), out_affine=oa)
        out["full"] = hip.gather_conv_cl(x, None, (6, 6), hip.all_tiles(H, H, (4, 4), (1, 1), (1, 1), DEV), sc_, sh_, act, p3, b, Co,
                                         (3, 3), (1, 1), full=full)
        out["sg"] = hip.scatter_gather_conv_cl(t4, y, (6, 6), idx, smap, sc_, sh_, act, p3, b, Co, (3, 3), (1, 1))
        o = y.clone(memory_format=torch.preserve_format)
        hip.scatter_gather_conv_scatter_cl(t4, y, (6, 6), idx, smap, sc_, sh_, act, p3, b, Co, (3, 3), (1, 1), o, residual=res)
        out["sgs"] = o
        n0 = hip.conv_pairs_fused()
        with hip.conv_pair(x):
            out["short"] = hip.gather_conv_cl(x, None, (4, 4), idx4, None, None, "identity", p1, b1, Co, (1, 1), (1, 1))
            out["conv1"] = hip.gather_conv_cl(x, None, (6, 6), idx, sc_ if sc_ is not None else torch.ones(1, C, 1, 1, device=DEV),
                                              sh_ if sh_ is not None else torch.zeros(1, C, 1, 1, device=DEV), "swish", p3, b, Co,
                                              (3, 3), (1, 1), out_affine=oa)
        out["paired"] = hip.conv_pairs_fused() - n0
        return out

    a, bx = run("f32"), run("f16x3")
    assert bx["paired"] == a["paired"] == 1
    for key in ("tiles", "full", "sg", "sgs", "short", "conv1"):
        d = float((a[key] - bx[key]).abs().max())
        assert d <= 5e-5 * (1.0 + float(a[key].abs().max())), (key, d)


@pytest.mark.parametrize("B,C,H,W,act", [(1, 128, 64, 64, "swish"), (2, 256, 16, 24, "swish"), (1, 64, 8, 8, "identity")])
def test_affine_act_cl_equals_torch(hip, B, C, H, W, act):
    """The one-pass activated copy of a ScatterGather cache (full pass) against the torch expression it replaces."""
    torch.manual_seed(C)
    x = _cl(torch.randn(B, C, H, W, device=DEV))
    sc, sh = torch.randn(B, C, 1, 1, device=DEV), torch.randn(1, C, 1, 1, device=DEV).expand(B, C, 1, 1).contiguous()
    got = hip.affine_act_cl(x, sc, sh, act)
    want = x * sc + sh
    want = F.silu(want) if act == "swish" else want
    assert got is not None and hip.is_cl(got)
    torch.testing.assert_close(got, want, rtol=2e-6, atol=1e-6)
    out = torch.empty_like(x)
    assert hip.affine_act_cl(x, sc, sh, act, out=out) is out and torch.equal(out, got)


@pytest.mark.parametrize("shape,groups", [((1, 128, 64, 64), 32), ((1, 512, 16, 16), 32), ((2, 256, 8, 24), 32)])
def test_group_norm_affine_with_channel_bias(hip, shape, groups):
    """GroupNorm statistics of x + bias[c] without materialising the sum (a residual block's norm of h + temb in the full pass):
    x * scale + shift == GroupNorm(x + bias)."""
    torch.manual_seed(shape[1])
    x = _cl(torch.randn(*shape, device=DEV) * 2 + 0.5)
    gamma, beta, cb = (torch.randn(shape[1], device=DEV) for _ in range(3))
    sc, sh = hip.group_norm_affine_cl(x, groups, 1e-6, gamma, beta, cb)
    want = F.group_norm((x + cb.view(1, -1, 1, 1)).double(), groups, gamma.double(), beta.double(), 1e-6).float()
    torch.testing.assert_close(x * sc + sh, want, rtol=0, atol=2e-5)


# ---- NCHW scatter_gather, row form (one lane per (channel, window row)) ------------------------------------------------
@pytest.mark.parametrize("bsize,k,B,C,res,act,first,affine", [
    (6, 3, 1, 200, 96, "swish", False, "channel"), (6, 3, 2, 64, 64, "identity", False, "none"),
    (6, 3, 1, 72, 64, "swish", True, "spatial"), (4, 1, 1, 136, 64, "swish", False, "channel"),
    (5, 3, 1, 40, 64, "identity", False, "channel"), (6, 3, 1, 8, 32, "swish", False, "batch"),
    # enough tiles for the grouped form (8 consecutive tiles x 32 channels per workgroup, halo pixels through LDS)
    (6, 3, 1, 200, 128, "swish", False, "channel"), (6, 3, 2, 96, 128, "identity", False, "none"),
    (6, 3, 1, 192, 128, "swish", True, "spatial"), (6, 3, 2, 128, 128, "swish", False, "batch")])
def test_scatter_gather_row_form_bit_exact(hip, bsize, k, B, C, res, act, first, affine, tuning):
    """The row form of the reference-layout scatter_gather (sige/cuda/scatter_gather_kernel.cu:8-67) is bit-identical to the
    element form and equals the oracle: windows over the image border, holes in the tile grid (rows that mix conv-1 tiles and
    the cached tensor), tiles narrower than the vector part (5x5 windows over 3x3 tiles), ragged channel chunks, every affine
    broadcast shape and both activation orders."""
    g = torch.Generator().manual_seed(bsize * 1000 + C)
    off = 1 if k == 3 else 0
    n_side = res // 4
    keep = torch.rand(n_side, n_side, generator=g) < 0.7
    idx = (keep.nonzero() * 4 - off).int().contiguous()            # origins -1, 3, ... for 3x3: the first row / column leaves the image
    N = idx.shape[0]
    R = bsize - k + 1                                               # conv-1's output tile
    x = torch.randn(B * N, C, R, R, generator=g)
    y = torch.randn(B, C, res, res, generator=g)
    shape = {"none": None, "channel": (1, C, 1, 1), "batch": (B, C, 1, 1), "spatial": (1, C, res, res)}[affine]
    scale = None if shape is None else torch.randn(*shape, generator=g)
    shift = None if shape is None else torch.randn(*shape, generator=g)
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    smap = hip.get_scatter_map(res, res, bsize, bsize, k, k, off, off, 1, 1, d(idx))
    want_map = oracle.get_scatter_map(res, res, bsize, bsize, k, k, off, off, 1, 1, idx)
    assert torch.equal(smap.cpu(), want_map)
    try:
        hip.scatter_gather_force_elements(True)
        elems = hip.scatter_gather(d(x), d(y), bsize, bsize, d(idx), smap, d(scale), d(shift), act, first)
        hip.scatter_gather_force_elements(False)
        rows = hip.scatter_gather(d(x), d(y), bsize, bsize, d(idx), smap, d(scale), d(shift), act, first)
    finally:
        hip.scatter_gather_force_elements(False)
    torch.cuda.synchronize()
    assert torch.equal(rows, elems)
    want = oracle.scatter_gather(x, y, bsize, bsize, idx, want_map, scale, shift, act, first)
    torch.testing.assert_close(rows.cpu(), want, rtol=1e-6, atol=1e-6)


# ---- per-channel statistics out of the dense-layer conv: the next GroupNorm without a pass over the tensor -----------------
@pytest.mark.parametrize("compute", ["f16x3", "f32"])
@pytest.mark.parametrize("k,c1,c2,cout,H,W,B,res,up", [
    (3, 128, 0, 128, 64, 64, 1, True, False),     # a full-pass layer, no K split
    (3, 1024, 0, 512, 16, 16, 1, True, False),    # deep K: the statistics come from the workgroup that finishes the K split
    (3, 128, 128, 256, 40, 24, 2, False, False),  # cat input, batch 2, H not a multiple of 8 (partly empty pixel blocks)
    (1, 256, 256, 128, 32, 32, 1, False, False),  # 1x1
    (3, 128, 0, 64, 32, 32, 1, False, True),      # nearest x2 upsampling fused into the conv
])
def test_wide_conv_stats_give_the_group_norm_affine(hip, compute, k, c1, c2, cout, H, W, B, res, up):
    """wide_conv_cl(stats=True) leaves (sum, sum of squares) per 8x8 pixel block and output channel; the GroupNorm affine taken
    from them (one launch over the partial sums) equals GroupNorm of the tensor the conv wrote -- with a per-channel bias
    (the timestep embedding), and over two tensors as the norm of their torch.cat."""
    g = torch.Generator().manual_seed(31 * k + cout + H)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    hs, ws = (H // 2, W // 2) if up else (H, W)
    x, x2 = _cl(r(B, c1, hs, ws)), (_cl(r(B, c2, hs, ws)) if c2 else None)
    w, b = r(cout, c1 + c2, k, k) / (k * (c1 + c2) ** 0.5), r(cout) * 0.5
    residual = _cl(r(B, cout, H, W)) if res else None
    packed = hip.wide_conv_pack_weights(w, compute)
    out = hip.wide_conv_cl(x, x2, None, None, "identity", packed, b, cout, (k, k), residual=residual, upsample2x=up, stats=True)
    st = hip.channel_stats(out)
    assert st is not None and st.channels == cout and st.count == H * W and st.tiles == ((H + 7) // 8) * ((W + 7) // 8)
    # the raw sums
    tot = st.data.double().view(B, st.tiles, cout, 2).sum(1)
    torch.testing.assert_close(tot[..., 0], out.double().sum((2, 3)), rtol=1e-5, atol=1e-2)
    torch.testing.assert_close(tot[..., 1], (out.double() ** 2).sum((2, 3)), rtol=1e-5, atol=1e-2)
    gamma, beta, cb = r(cout), r(cout), r(cout)
    for bias in (None, cb):
        sc, sh = hip.group_norm_affine_from_stats([st], 32, 1e-6, gamma, beta, bias)
        xin = out if bias is None else out + bias.view(1, -1, 1, 1)
        want = F.group_norm(xin.double(), 32, gamma.double(), beta.double(), 1e-6).float()
        torch.testing.assert_close(out * sc + sh, want, rtol=0, atol=3e-5)
    # two parts = the norm of the cat; the second tensor at another resolution's statistics is the caller's business: same here
    out2 = hip.wide_conv_cl(x, x2, None, None, "identity", packed, b * 2.0, cout, (k, k), upsample2x=up, stats=True)
    g2, b2 = r(2 * cout), r(2 * cout)
    sc, sh = hip.group_norm_affine_from_stats([st, hip.channel_stats(out2)], 32, 1e-6, g2, b2)
    both = torch.cat([out, out2], 1)
    want = F.group_norm(both.double(), 32, g2.double(), b2.double(), 1e-6).float()
    torch.testing.assert_close(both * sc + sh, want, rtol=0, atol=3e-5)
    # an in-place change of the tensor, or any new tensor, drops the statistics
    assert hip.channel_stats(out.clone()) is None
    out.add_(1.0)
    assert hip.channel_stats(out) is None


@pytest.mark.parametrize("shape,c2", [((1, 128, 64, 64), 0), ((1, 512, 32, 32), 256), ((2, 256, 24, 40), 128), ((1, 36, 10, 6), 0)])
def test_channel_stats_of_a_tensor_and_the_norm_of_a_cat(hip, shape, c2):
    """hip.channel_stats_cl (one pass over a tensor whose producer left no statistics) feeds the same finisher; 768 = 512 + 256
    channels in 32 groups of 24 (a group size that does not divide the workgroup) and 384 = 256 + 128 in groups of 12 are the
    U-Net's up-path norms over a torch.cat that the full pass never builds."""
    torch.manual_seed(shape[1] + c2)
    B, C1, H, W = shape
    x = _cl(torch.randn(*shape, device=DEV) * 1.5 + 0.3)
    st = hip.channel_stats_cl(x)
    assert st is not None and hip.channel_stats(x) is st and st.count == H * W
    tot = st.data.double().view(B, st.tiles, C1, 2).sum(1)
    torch.testing.assert_close(tot[..., 0], x.double().sum((2, 3)), rtol=1e-5, atol=1e-2)
    groups = 32 if (C1 + c2) % 32 == 0 else 4
    parts, full = [st], x
    if c2:
        x2 = _cl(torch.randn(B, c2, H, W, device=DEV) - 0.2)
        parts.append(hip.channel_stats_cl(x2))
        full = torch.cat([x, x2], 1)
    C = C1 + c2
    gamma, beta, cb = (torch.randn(C, device=DEV) for _ in range(3))
    for bias in (None, cb):
        sc, sh = hip.group_norm_affine_from_stats(parts, groups, 1e-6, gamma, beta, bias)
        xin = full if bias is None else full + bias.view(1, -1, 1, 1)
        want = F.group_norm(xin.double(), groups, gamma.double(), beta.double(), 1e-6).float()
        torch.testing.assert_close(full * sc + sh, want, rtol=0, atol=3e-5)


def test_graph_pool_recapture_per_mask(ddpm_gpu, ddpm_reference):
    """sige_amd.graphs.GraphPool: a new mask -> capture straight away (the capture is the first forward under that mask) ->
    replay gives exactly what an eager forward gives, for alternating masks, and the reserved memory stops growing once the pool
    is warm (a destroyed graph's blocks are reused by its successor's capture)."""
    import bench
    from sige_amd.graphs import GraphPool
    from sige_amd.utils import dilate_mask, downsample_mask

    model = ddpm_gpu
    cl = lambda a: a.to(DEV).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x0, noise, t = cl(ddpm_reference["x0"]), cl(ddpm_reference["noise"]), torch.zeros(1, device=DEV)
    pool = GraphPool(x0.device)
    g = None
    reserved = []
    with torch.no_grad():
        model.set_compute_dtype("f32")
        model.set_mode("full")
        model(x0, t)
        for i in range(6):
            m = bench.edit_mask((0.012, 0.03, 0.02)[i % 3]).to(x0.device)
            x1 = x0 + noise * m
            model.set_masks(downsample_mask(dilate_mask(m, 5), 8))
            model.set_mode("sparse")
            if i == 0:
                model(x1, t)  # (after a full pass the first sparse forward registers the activated twins: DESIGN 3.1)
            del g
            g, out = pool.capture(lambda: model(x1, t))
            g.replay()
            torch.cuda.synchronize()
            got = out.clone()
            assert torch.equal(got, model(x1, t))
            reserved.append(torch.cuda.memory_reserved(x0.device))
    assert reserved[-1] == reserved[-2] == reserved[-3], reserved
