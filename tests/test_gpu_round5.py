"""Round 5, on the MI355X: the stacked-edit mode pinned to the CPU oracle directly, the code-object preload
(sige_hip_preload), launch plans and pointers the plan does not own, the weight-sharing tile conv."""
import os

import pytest
import torch

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


def _build_masks(mask):
    from sige_amd.utils import dilate_mask, downsample_mask

    return downsample_mask(dilate_mask(mask, 5), 8)


def _cpu_reference_backend():
    """The reference's own sige/cpu (oracle/_ref, built in the container from /root/reference and shipped as a .so) when it is
    there, else the C restatement -- the same choice as bench.py's parity leg."""
    from oracle import oracle

    try:
        from oracle import build_ref

        return oracle.as_backend(build_ref.load()), "oracle/_ref"
    except Exception:
        return oracle, "oracle (C restatement)"


# ---- stacked edits against the CPU oracle, per edit ---------------------------------------------
def test_stacked_edits_vs_cpu_oracle(hip):
    """E = 4 edits of one original, each with its OWN mask, through ONE stacked forward on the GPU (sige_amd/stacked.py,
    sige_hip_set_edit_batch); every edit's slice of the output against THAT edit's sparse forward of the same network on the CPU
    with the oracle as native backend (the reference's semantics: one forward per mask, sige/nn/base.py:115-129).  Seam cases:
    edit 1 touches rows 0.. of its image and edit 2 rows ..255 of its image -- adjacent images of the tall tensor, so the halos
    of active tiles on BOTH sides of the seam between images 1|2 and 2|3 would read the neighbour without the seam rule; edit 3
    covers columns 0.. as well.  north_star tolerance 1e-3."""
    import bench
    from oracle import oracle
    from sige_amd import runtime, stacked
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval()
    x0, noise = bench.make_inputs()
    t = torch.zeros(1)
    places = [(0.012, 100, 90), (0.03, 256 - 44, 150), (0.02, 0, 40), (0.05, 0, 0)]
    masks = [bench.square_mask(*p[:1], top=p[1], left=p[2]) for p in places]
    assert masks[1][255].any() and masks[2][0].any() and masks[3][0, 0]
    E = len(masks)
    backend, _ = _cpu_reference_backend()
    n_thr = min(32, os.cpu_count() or 1)
    torch.set_num_threads(n_thr)
    oracle.set_num_threads(n_thr)
    runtime.register_backend("cpu", backend)
    want = []
    try:
        with torch.no_grad():
            model.set_mode("full")
            full_c = model(x0, t)
            for m in masks:
                model.set_masks(_build_masks(m))
                model.set_mode("sparse")
                want.append(model(x0 + noise * m, t)[0].clone())
    finally:
        runtime.unregister_backend("cpu")
    model.clear_cache()
    model = model.to(DEV).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0g, ng, tg = _cl(x0.to(DEV)), _cl(noise.to(DEV)), t.to(DEV)
    gm = [m.to(DEV) for m in masks]
    with torch.no_grad():
        model.set_mode("full")
        model(x0g, tg)
        xe = _cl(torch.cat([x0g + ng * m for m in gm], 0))
        stacked.stack_caches(model, E)
        try:
            stacked.set_masks(model, [_build_masks(m) for m in gm])
            model.set_mode("sparse")
            with stacked.edit_batch(model, E):
                model(xe, tg)  # (registers the activated twins; from the next forward on they are read)
                out = model(xe, tg).clone()
        finally:
            stacked.unstack_caches(model)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (E, 3, 256, 256)
    for e in range(E):
        err = float((out[e].cpu() - want[e]).abs().max())
        assert err <= util.CONV_ATOL, (e, places[e], err)
        # the edit is a real one: its sparse output differs from the original's
        assert float((want[e] - full_c[0]).abs().max()) > 1e-2


# ---- code-object preload (the ~40 ms one-off of a new tile count) -----------------------------------------
def test_preload_touches_every_translation_unit(hip):
    from sige_amd import build

    # the first launch on the device already preloaded (hip._stream); asking again is a no-op ...
    torch.zeros(4, device=DEV)
    x = torch.randn(1, 8, 16, 16, device=DEV)
    hip.gather(x, 6, 6, torch.tensor([[0, 0]], dtype=torch.int32, device=DEV), None, None, "identity", False)
    assert torch.cuda.current_device() in hip.lib().preloaded
    assert hip.lib().sige_hip_preload() == 0
    # ... and the first one touched one anchor kernel per translation unit of the library
    if not os.environ.get("SIGE_HIP_NO_PRELOAD"):
        assert hip.lib().preloaded_units[torch.cuda.current_device()] == len(build.SOURCES)
