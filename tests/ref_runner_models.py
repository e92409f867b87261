#!/usr/bin/env python3
"""Run the reference's UNCHANGED GauGAN / Stable-Diffusion SIGE models on CPU through one of two
software stacks and dump the outputs (helper of tests/test_reference_models.py; build container only):

  --stack reference   the reference's sige.nn + its compiled sige/cpu backend
  --stack ours        the same model file, but `sige` is sige_amd (compat.install) with the CPU oracle as backend
                      (--deferred: Gather / ScatterGather return DeferredTiles as they do on the GPU)
  --stack ours-workload   sige_amd's own workload model (sige_amd/workloads/gaugan_spade.py, sd_unet.py) loading the
                      reference model's state dict

  --model gaugan      gaugan/models/spade_generators/sige_fused_spade_generator.py   (BASELINE configs[2])
  --model sd          stable-diffusion/ldm/modules/diffusionmodules/sige_openaimodel.py (BASELINE configs[3])
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("SIGE_REFERENCE", "/root/reference")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stack", required=True, choices=["reference", "ours", "ours-workload"])
    ap.add_argument("--model", required=True, choices=["gaugan", "sd", "sdt"])
    ap.add_argument("--state", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--deferred", action="store_true")
    a = ap.parse_args()

    sys.path = [p for p in sys.path if os.path.abspath(p or ".") not in (REPO, HERE)]
    if a.stack == "reference":
        sys.path.insert(0, REF)
    sys.path.insert(1, os.path.join(REF, "gaugan" if a.model == "gaugan" else "stable-diffusion"))
    if a.model == "sdt" and a.stack != "ours-workload":
        import types as _t

        for name in ("omegaconf", "omegaconf.listconfig"):  # (imported by the package, unused by this module)
            sys.modules.setdefault(name, _t.ModuleType(name))
    sys.path.append(REPO)

    import torch

    torch.manual_seed(0)
    torch.set_num_threads(8)
    if a.stack == "reference":
        from oracle import build_ref

        build_ref.build(REF, verbose=False)
        ref_cpu = build_ref.load()
        import sige

        assert os.path.abspath(sige.__file__).startswith(REF)
        sys.modules["sige.cpu"] = ref_cpu
        sige.cpu = ref_cpu
    else:
        from oracle import oracle
        from sige_amd import compat, runtime

        compat.install()
        runtime.register_backend("cpu", oracle)
        if a.deferred:
            from sige_amd.nn import deferred

            deferred.FORCE_ON_CPU = True
    from sige.utils import compute_difference_mask, dilate_mask, downsample_mask

    import warnings

    warnings.simplefilter("ignore")
    rs = np.random.RandomState(3)
    if a.model == "gaugan":
        if a.stack == "ours-workload":
            from sige_amd.workloads.gaugan_spade import SPADEConfig, SpadeGenerator

            model = SpadeGenerator(SPADEConfig(ngf=16, crop_size=256)).eval()
        else:
            from models.spade_generators.sige_fused_spade_generator import SIGEFusedSPADEGenerator

            opt = argparse.Namespace(ngf=16, semantic_nc=36, norm_G="spadesyncbatch3x3", num_upsampling_layers="more",
                                     main_block_size=6, shortcut_block_size=4, num_sparse_layers=5, crop_size=256,
                                     aspect_ratio=2, separable_conv_norm="instance")
            model = SIGEFusedSPADEGenerator(opt).eval()
        H, W = 128, 256
        lab0 = rs.randint(0, 36, size=(H, W))
        lab1 = lab0.copy()
        lab1[40:70, 100:150] = (lab0[40:70, 100:150] + 5) % 36  # a 5 % region relabelled
        onehot = lambda l: torch.nn.functional.one_hot(torch.from_numpy(l), 36).permute(2, 0, 1)[None].float()  # noqa: E731
        x0, x1 = onehot(lab0), onehot(lab1)
        diff = compute_difference_mask(x0, x1)
        masks = downsample_mask(dilate_mask(diff, 1), (model.sh, model.sw), dilation=2)
        run = lambda x: model(x)  # noqa: E731
    elif a.model == "sdt":
        # one sparse-query spatial transformer (SD v1 level-1 shape, scaled down), CFG batch 2
        if a.stack == "ours-workload":
            from sige_amd.workloads.sd_transformer import SpatialTransformer

            model = SpatialTransformer(64, 4, 16, depth=1, context_dim=96, block_size=4).eval()
        else:
            from ldm.modules.sige_attention import SIGESpatialTransformer

            model = SIGESpatialTransformer(64, 4, 16, depth=1, context_dim=96, use_checkpoint=False, block_size=4).eval()
        from sige.nn import SIGEModel

        class Wrap(SIGEModel):  # (a SIGEModel drives set_masks / set_mode; the same wrapper on both stacks)
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, x, **kw):
                return self.m(x, **kw)

        model = Wrap(model).eval()
        g = torch.Generator().manual_seed(1)
        for p_ in model.parameters():  # (proj_out is zero-initialised in the reference)
            if p_.abs().max() == 0:
                p_.data.copy_(torch.randn(p_.shape, generator=g) * 0.05)
        x0 = torch.from_numpy(rs.standard_normal((2, 64, 32, 32)).astype(np.float32))
        m = torch.zeros(32, 32, dtype=torch.bool)
        m[9:21, 6:19] = True
        x1 = x0 + torch.from_numpy(rs.standard_normal((2, 64, 32, 32)).astype(np.float32)) * m
        masks = {(32, 32): m}
        ctx = torch.from_numpy(rs.standard_normal((2, 77, 96)).astype(np.float32))
        run = lambda x: model(x, context=ctx)  # noqa: E731
    else:
        oc = types.ModuleType("omegaconf")
        lc = types.ModuleType("omegaconf.listconfig")

        class ListConfig(list):
            pass

        lc.ListConfig = ListConfig
        oc.listconfig = lc
        sys.modules["omegaconf"], sys.modules["omegaconf.listconfig"] = oc, lc
        if a.stack == "ours-workload":
            from sige_amd.workloads.sd_unet import SDConfig, SDUNet

            model = SDUNet(SDConfig(model_channels=64, num_res_blocks=1, num_heads=4, context_dim=96)).eval()
        else:
            from ldm.modules.diffusionmodules.sige_openaimodel import SIGEUNetModel

            model = SIGEUNetModel(image_size=32, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1,
                                  attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_heads=4,
                                  use_spatial_transformer=True, transformer_depth=1, context_dim=96, use_checkpoint=False,
                                  legacy=False).eval()
        g = torch.Generator().manual_seed(1)
        for p in model.parameters():  # zero_module()-initialised convs would make every output trivially equal
            if p.abs().max() == 0:
                p.data.copy_(torch.randn(p.shape, generator=g) * 0.05)
        x0 = torch.from_numpy(rs.standard_normal((2, 4, 64, 64)).astype(np.float32))
        mask512 = torch.zeros(512, 512, dtype=torch.bool)
        mask512[150:350, 120:320] = True  # ~15 % of the image
        mlat = torch.nn.functional.interpolate(mask512[None, None].float(), size=(64, 64))[0, 0] > 0.5
        x1 = x0 + torch.from_numpy(rs.standard_normal((2, 4, 64, 64)).astype(np.float32)) * mlat
        masks = downsample_mask(mask512, min_res=8, dilation=1)
        ts = torch.tensor([500.0, 500.0])
        ctx = torch.from_numpy(rs.standard_normal((2, 77, 96)).astype(np.float32))
        run = lambda x: model(x, ts, context=ctx)  # noqa: E731

    if a.stack == "reference":
        # BatchNorm running statistics away from (0, 1): the cached affine then really matters
        g = torch.Generator().manual_seed(7)
        for name, buf in model.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.3)
            elif name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
        torch.save(model.state_dict(), a.state)
    else:
        res = model.load_state_dict(torch.load(a.state), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    with torch.no_grad():
        model.set_mode("full")
        full = run(x0)
        model.set_masks(masks)
        model.set_mode("sparse")
        sparse = run(x1)
    np.savez(a.out, full=full.numpy(), sparse=sparse.numpy())
    print("ok", a.stack, a.model, float(full.abs().mean()), float(sparse.abs().mean()), float((sparse - full).abs().max()))


if __name__ == "__main__":
    main()
