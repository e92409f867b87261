"""BASELINE.json configs[2] / [3] as parity cases on committed fixtures (tests/golden/models.npz, produced from the REAL
reference's model classes by tests/golden/make_model_golden.py): sige_amd's workload models with the same name-keyed weights.
CPU (oracle backend) here; the `gpu`-marked tests run the HIP kernels -- module chain and fused SPADE modulation."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from tests.golden.model_init import gaugan_labels, init_by_name, sd_transformer_inputs, sd_unet_inputs, summarize  # noqa: E402

pytestmark = pytest.mark.oracle_parity  # (every test here is pinned to tests/golden/models.npz = the real reference's outputs)
GOLDEN = np.load(os.path.join(REPO, "tests", "golden", "models.npz"))
ATOL = 1e-3  # north_star: activations within 1e-3 fp32 on conv-containing paths


def _check(prefix, t, atol=ATOL):
    cs = GOLDEN[prefix + "/cstep"]
    s = summarize(t, cstep=int(cs[0]), step=int(cs[1]) if len(cs) > 1 else 4)
    assert list(GOLDEN[prefix + "/shape"]) == s["shape"]
    np.testing.assert_allclose(s["sub"], GOLDEN[prefix + "/sub"], rtol=0, atol=atol)
    n = float(np.prod(s["shape"]))
    assert abs(s["sum"] - GOLDEN[prefix + "/sums"][0]) <= atol * n * 0.05       # (errors are signed: the sum moves far less)
    assert abs(s["abs_sum"] - GOLDEN[prefix + "/sums"][1]) <= atol * n * 0.05


def _gaugan(device, channels_last, fused):
    from sige_amd.utils import compute_difference_mask, dilate_mask, downsample_mask
    from sige_amd.workloads.gaugan_spade import SPADEConfig, SpadeGenerator

    model = SpadeGenerator(SPADEConfig(fused=fused)).eval()
    init_by_name(model)
    x0, x1 = gaugan_labels()
    model, x0, x1 = model.to(device), x0.to(device), x1.to(device)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
        x0, x1 = x0.contiguous(memory_format=torch.channels_last), x1.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        model.set_mode("full")
        full = model(x0)
        diff = compute_difference_mask(x0, x1)
        model.set_masks(downsample_mask(dilate_mask(diff, 1), (model.sh, model.sw), dilation=2))
        model.set_mode("sparse")
        sparse = model(x1)
    assert abs(float(diff.float().mean()) - float(GOLDEN["gaugan/edit_ratio"][0])) < 1e-9
    return full, sparse


def test_gaugan_workload_on_the_oracle_backend_matches_the_reference_fixture():
    from oracle import oracle
    from sige_amd import runtime

    torch.set_num_threads(8)
    runtime.register_backend("cpu", oracle)
    try:
        full, sparse = _gaugan("cpu", False, False)
    finally:
        runtime.unregister_backend("cpu")
    _check("gaugan/full", full, 1e-5)
    _check("gaugan/sparse", sparse, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("channels_last,fused", [(False, False), (True, False), (True, True)])
def test_gaugan_generator_on_the_gpu_matches_the_reference_fixture(channels_last, fused):
    """The full 256x512 SPADE generator (93 M parameters, 5 % edit) through the HIP kernels: the reference's NCHW layout,
    channels-last with the module chain, channels-last with the one-pass SPADE modulation kernel."""
    full, sparse = _gaugan("cuda", channels_last, fused)
    _check("gaugan/full", full)
    _check("gaugan/sparse", sparse)


@pytest.mark.gpu
def test_spade_modulate_kernel_equals_the_module_chain():
    """sige_hip_spade_modulate_nhwc_f32 against Gather / ScatterGather + split + modulation + leaky ReLU as separate ops:
    the same fp32 operations in the same order -> bit-identical."""
    from sige_amd import hip
    from sige_amd.utils import reduce_mask

    torch.manual_seed(1)
    dev = "cuda"
    C, H, W, B = 64, 48, 40, 2
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)  # noqa: E731
    mask = torch.zeros(H, W, dtype=torch.bool, device=dev)
    mask[10:25, 8:30] = True
    mask[0, 0] = mask[H - 1, W - 1] = True
    idx6, idx4 = reduce_mask(mask, 6, 4, 1), reduce_mask(mask, 4, 4, 0)
    smap = hip.get_scatter_map(H, W, 6, 6, 3, 3, 1, 1, 1, 1, idx6)
    x, y = cl(torch.randn(B, C, H, W, device=dev)), cl(torch.randn(B, C, H, W, device=dev))
    gb_full = cl(torch.randn(B, 2 * C, H, W, device=dev))
    gb_t = cl(torch.randn(B * idx6.shape[0], 2 * C, 4, 4, device=dev))
    xt = cl(torch.randn(B * idx6.shape[0], C, 4, 4, device=dev))
    sc, sh = torch.randn(1, C, 1, 1, device=dev), torch.randn(1, C, 1, 1, device=dev)

    def chain(n_tiles, gb_tiles, slope):
        g, b = torch.split(gb_tiles, C, dim=1)
        o = n_tiles * (1 + g) + b
        return o if slope is None else torch.nn.functional.leaky_relu(o, slope)

    # main, first layer: Gather(x) modulated by ScatterGather(gamma|beta)
    want = chain(hip.gather_cl(x, 6, 6, idx6, sc, sh), hip.scatter_gather_cl(gb_t, gb_full, 6, 6, idx6, smap), 0.2)
    got = hip.spade_modulate_cl(x, None, None, sc, sh, gb_t, gb_full, smap, idx6, (6, 6), 0.2)
    assert torch.equal(got, want)
    # main, second layer: ScatterGather(conv tiles, cache) modulated
    want = chain(hip.scatter_gather_cl(xt, y, 6, 6, idx6, smap, sc, sh), hip.scatter_gather_cl(gb_t, gb_full, 6, 6, idx6, smap), 0.2)
    got = hip.spade_modulate_cl(y, xt, smap, sc, sh, gb_t, gb_full, smap, idx6, (6, 6), 0.2)
    assert torch.equal(got, want)
    # shortcut: 4x4 tiles of the 1x1 conv; gamma|beta = Gather(Scatter(.)) == the same lookup through the map; no activation
    table = hip.tile_table(idx6, (1, 1), (1, 1), (4, 4), (H, W))
    scattered = hip.scatter_cl(gb_t, gb_full, (1, 1), (1, 1), idx6, table)
    want = chain(hip.gather_cl(x, 4, 4, idx4, sc, sh), hip.gather_cl(scattered, 4, 4, idx4), None)
    got = hip.spade_modulate_cl(x, None, None, sc, sh, gb_t, gb_full, smap, idx4, (4, 4), None)
    assert torch.equal(got, want)


# ---- Stable Diffusion: the sparse-query spatial transformer (BASELINE.json configs[3]) -------------
def _sd_transformer(device, channels_last, inplace, sparse_kv=True):
    from sige_amd.nn import SIGEModel
    from sige_amd.utils import downsample_mask
    from sige_amd.workloads.sd_transformer import SpatialTransformer

    class Wrap(SIGEModel):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x, **kw):
            return self.m(x, **kw)

    model = Wrap(SpatialTransformer(320, 8, 40, depth=1, context_dim=768, block_size=4, sparse_kv=sparse_kv)).eval()
    init_by_name(model)
    x0, noise, ctx, mask512 = sd_transformer_inputs()
    model, x0, noise, ctx, mask512 = model.to(device), x0.to(device), noise.to(device), ctx.to(device), mask512.to(device)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
        x0, noise = x0.contiguous(memory_format=torch.channels_last), noise.contiguous(memory_format=torch.channels_last)
    model.set_scatter_inplace(inplace)
    masks = downsample_mask(mask512, min_res=8, dilation=1)  # stable-diffusion/runners/inpainting_runner.py:50-54
    x1 = x0 + noise * masks[(64, 64)]
    with torch.no_grad():
        model.set_mode("full")
        full = model(x0, context=ctx)
        model.set_masks(masks)
        model.set_mode("sparse")
        sparse = model(x1, context=ctx)
        again = model(x1, context=ctx)  # persistent buffers / cached K, V must survive a second step
    assert torch.equal(sparse, again)
    assert abs(float(masks[(64, 64)].float().mean()) - float(GOLDEN["sdt/active_ratio"][0])) < 1e-9
    return full, sparse


@pytest.mark.gpu
@pytest.mark.parametrize("channels_last,inplace,sparse_kv", [(False, False, False), (True, False, True), (True, True, True)])
def test_sd_spatial_transformer_on_the_gpu_matches_the_reference_fixture(channels_last, inplace, sparse_kv):
    """SD v1 level-1 transformer (320 ch, 8 heads, text context 768) at the 64 x 64 latent, CFG batch 2, 15 % edit: the
    reference's NCHW form with permute copies and full K / V re-projection; channels-last (tile <-> token glue as views) with
    K / V refreshed by two Scatters; the same with in-place persistent scatter outputs."""
    full, sparse = _sd_transformer("cuda", channels_last, inplace, sparse_kv)
    _check("sdt/full", full)
    _check("sdt/sparse", sparse)


# ---- Stable Diffusion: the whole U-Net (structure of SD v1 at model_channels 128) -----------------------------------------
def _sd_unet(device, channels_last, inplace):
    from sige_amd.utils import downsample_mask
    from sige_amd.workloads.sd_unet import SDConfig, SDUNet

    model = SDUNet(SDConfig(model_channels=128)).eval()
    init_by_name(model)
    x0, noise, ctx, ts, mask512 = (t.to(device) for t in sd_unet_inputs())
    model = model.to(device)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
        x0, noise = x0.contiguous(memory_format=torch.channels_last), noise.contiguous(memory_format=torch.channels_last)
    model.set_scatter_inplace(inplace)
    masks = downsample_mask(mask512, min_res=8, dilation=1)
    x1 = x0 + noise * masks[(64, 64)]
    with torch.no_grad():
        model.set_mode("full")
        full = model(x0, ts, context=ctx)
        model.set_masks(masks)
        model.set_mode("sparse")
        sparse = model(x1, ts, context=ctx)
    return full, sparse


@pytest.mark.gpu
@pytest.mark.parametrize("channels_last,inplace", [(False, False), (True, True)])
def test_sd_unet_on_the_gpu_matches_the_reference_fixture(channels_last, inplace):
    """BASELINE.json configs[3] at model level: the SD v1 U-Net structure (137 M parameters at model_channels 128), CFG batch
    2, 15 % edit, through the HIP kernels in the reference's layout and channels-last with in-place persistent outputs."""
    full, sparse = _sd_unet("cuda", channels_last, inplace)
    _check("sdunet/full", full)
    _check("sdunet/sparse", sparse)


def test_sd_unet_workload_on_the_oracle_backend_matches_the_reference_fixture():
    from oracle import oracle
    from sige_amd import runtime

    torch.set_num_threads(8)
    runtime.register_backend("cpu", oracle)
    try:
        full, sparse = _sd_unet("cpu", False, False)
    finally:
        runtime.unregister_backend("cpu")
    _check("sdunet/full", full, 2e-5)
    _check("sdunet/sparse", sparse, 1e-4)
