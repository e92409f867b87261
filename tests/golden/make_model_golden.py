#!/usr/bin/env python3
"""Generate tests/golden/models.npz from the REAL reference's model classes (build container only: /root/reference).

    python tests/golden/make_model_golden.py

BASELINE.json configs[2] (GauGAN SPADE generator, 256x512, ~5 % edit) and configs[3] (Stable-Diffusion U-Net block stack,
CFG batch 2, 15 % edit) run on the reference's own sige.nn + its compiled sige/cpu backend (oracle/_ref), with weights drawn
by tests/golden/model_init.py::init_by_name (name-keyed: the GPU tests re-create exactly these weights in sige_amd's workload
models).  Stored: every 4th pixel of the full and sparse outputs + sums over all values.  /root/reference does not exist on
the GPU box, hence committed fixtures.
"""
import argparse
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SIGE_REFERENCE", "/root/reference")
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.append(REPO)
from oracle import build_ref  # noqa: E402
from tests.golden.model_init import gaugan_labels, init_by_name, sd_transformer_inputs, sd_unet_inputs, summarize  # noqa: E402

build_ref.build(REF, verbose=False)
ref_cpu = build_ref.load()
import sige  # noqa: E402

assert os.path.abspath(sige.__file__).startswith(REF), sige.__file__
sys.modules["sige.cpu"] = ref_cpu
sige.cpu = ref_cpu
from sige.utils import compute_difference_mask, dilate_mask, downsample_mask  # noqa: E402

torch.set_num_threads(8)
out = {}


def put(prefix, t, cstep=1, step=4):
    s = summarize(t, step=step, cstep=cstep)
    out[prefix + "/cstep"] = np.array([cstep, step], dtype=np.int64)
    out[prefix + "/sub"] = s["sub"]
    out[prefix + "/sums"] = np.array([s["sum"], s["abs_sum"]], dtype=np.float64)
    out[prefix + "/shape"] = np.array(s["shape"], dtype=np.int64)


def gaugan():
    sys.path.insert(1, os.path.join(REF, "gaugan"))
    from models.spade_generators.sige_fused_spade_generator import SIGEFusedSPADEGenerator

    opt = argparse.Namespace(ngf=64, semantic_nc=36, norm_G="spadesyncbatch3x3", num_upsampling_layers="more",
                             main_block_size=6, shortcut_block_size=4, num_sparse_layers=5, crop_size=512, aspect_ratio=2,
                             separable_conv_norm="instance")
    model = SIGEFusedSPADEGenerator(opt).eval()
    init_by_name(model)
    x0, x1 = gaugan_labels()
    with torch.no_grad():
        model.set_mode("full")
        full = model(x0)
        diff = compute_difference_mask(x0, x1)
        masks = downsample_mask(dilate_mask(diff, 1), (model.sh, model.sw), dilation=2)  # gaugan/test.py recipe
        model.set_masks(masks)
        model.set_mode("sparse")
        sparse = model(x1)
        model.set_mode("full")
        dense_edit = model(x1)
    put("gaugan/full", full)
    put("gaugan/sparse", sparse)
    out["gaugan/edit_ratio"] = np.array([float(diff.float().mean())])
    print("gaugan: edit ratio %.3f, |sparse - full| max %.3f, |sparse - dense(edited)| max %.3f"
          % (float(diff.float().mean()), float((sparse - full).abs().max()), float((sparse - dense_edit).abs().max())))
    sys.path.pop(1)


def sd_transformer():
    """stable-diffusion/ldm/modules/sige_attention.py::SIGESpatialTransformer at the SD v1 level-1 shape (320 channels, 8 heads,
    context 768), 64 x 64 latent, CFG batch 2 with per-sample cached affine, 15 % edit (inpainting_runner.py:50-54 masks)."""
    sys.path.insert(1, os.path.join(REF, "stable-diffusion"))
    for name in ("omegaconf", "omegaconf.listconfig"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from ldm.modules.sige_attention import SIGESpatialTransformer
    from sige.nn import SIGEModel

    class Wrap(SIGEModel):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x, **kw):
            return self.m(x, **kw)

    model = Wrap(SIGESpatialTransformer(320, 8, 40, depth=1, context_dim=768, use_checkpoint=False, block_size=4)).eval()
    init_by_name(model)
    x0, noise, ctx, mask512 = sd_transformer_inputs()
    masks = downsample_mask(mask512, min_res=8, dilation=1)
    m64 = masks[(64, 64)]
    x1 = x0 + noise * m64
    with torch.no_grad():
        model.set_mode("full")
        full = model(x0, context=ctx)
        model.set_masks(masks)
        model.set_mode("sparse")
        sparse = model(x1, context=ctx)
    put("sdt/full", full, cstep=5)
    put("sdt/sparse", sparse, cstep=5)
    out["sdt/active_ratio"] = np.array([float(m64.float().mean())])
    print("sd transformer: active ratio %.3f at 64x64, |sparse - full| max %.3f" % (float(m64.float().mean()), float((sparse - full).abs().max())))
    sys.path.pop(1)


def sd_unet():
    """stable-diffusion/ldm/modules/diffusionmodules/sige_openaimodel.py::SIGEUNetModel with SD v1's structure (2 res blocks per
    level, channel mult 1-2-4-4, transformers at ds 1 / 2 / 4, 8 heads, context 768) at model_channels 128 (the full 320 is
    860 M parameters: too slow for a CPU fixture), 64 x 64 latent, CFG batch 2, 15 % edit (inpainting_runner.py:50-54)."""
    sys.path.insert(1, os.path.join(REF, "stable-diffusion"))
    oc, lc = types.ModuleType("omegaconf"), types.ModuleType("omegaconf.listconfig")

    class ListConfig(list):
        pass

    lc.ListConfig = ListConfig
    oc.listconfig = lc
    sys.modules["omegaconf"], sys.modules["omegaconf.listconfig"] = oc, lc
    from ldm.modules.diffusionmodules.sige_openaimodel import SIGEUNetModel

    model = SIGEUNetModel(image_size=32, in_channels=4, model_channels=128, out_channels=4, num_res_blocks=2,
                          attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                          transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False).eval()
    init_by_name(model)
    x0, noise, ctx, ts, mask512 = sd_unet_inputs()
    masks = downsample_mask(mask512, min_res=8, dilation=1)
    x1 = x0 + noise * masks[(64, 64)]
    with torch.no_grad():
        model.set_mode("full")
        full = model(x0, ts, context=ctx)
        model.set_masks(masks)
        model.set_mode("sparse")
        sparse = model(x1, ts, context=ctx)
    put("sdunet/full", full, step=2)
    put("sdunet/sparse", sparse, step=2)
    print("sd unet (mc 128, %.1fM params): |sparse - full| max %.3f, |out| max %.2f"
          % (sum(p.numel() for p in model.parameters()) / 1e6, float((sparse - full).abs().max()), float(sparse.abs().max())))
    sys.path.pop(1)


if __name__ == "__main__":
    import warnings

    warnings.simplefilter("ignore")
    gaugan()
    sd_transformer()
    sd_unet()
    path = os.path.join(HERE, "models.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
