#!/usr/bin/env python3
"""Run the DDPM SIGE U-Net on CPU through one of three software stacks and dump
the outputs (helper of tests/test_reference_models.py; build container only).

  --stack reference      the reference's sige.nn + its compiled sige/cpu backend +
                         its own model file (diffusion/models/ddpm_arch/sige_fused_unet.py)
  --stack ours-refmodel  the SAME unchanged reference model file, but `sige` is
                         sige_amd (compat.install) with the CPU oracle as backend
  --stack ours-workload  sige_amd.workloads.ddpm_unet.DDPMSparseUNet (+ oracle backend),
                         loading the reference model's state dict
"""
import argparse
import os
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("SIGE_REFERENCE", "/root/reference")


class AttrDict(dict):
    def __getattr__(self, k):
        v = self[k]
        return AttrDict(v) if isinstance(v, dict) else v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stack", required=True, choices=["reference", "ours-refmodel", "ours-workload"])
    ap.add_argument("--ch", type=int, default=32)
    ap.add_argument("--ratio", type=float, default=0.05)
    ap.add_argument("--state", required=True, help="state-dict file (written by `reference`, read by the others)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--deferred", action="store_true", help="force DeferredTiles on CPU (fusion mechanism)")
    a = ap.parse_args()

    sys.path = [p for p in sys.path if os.path.abspath(p or ".") not in (REPO, HERE)]
    if a.stack == "reference":
        sys.path.insert(0, REF)
    sys.path.insert(1, os.path.join(REF, "diffusion"))
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
    from sige.utils import dilate_mask, downsample_mask

    cfg = yaml.safe_load(open(os.path.join(REF, "diffusion", "configs", "church_ddpm256-sige.yml")))
    cfg["model"]["ch"] = a.ch
    if a.stack == "ours-workload":
        from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

        m = cfg["model"]
        model = DDPMSparseUNet(DDPMConfig(ch=m["ch"], ch_mult=tuple(m["ch_mult"]), num_res_blocks=m["num_res_blocks"],
                                          attn_resolutions=tuple(m["attn_resolutions"]),
                                          main_block=m["sige_block_size"]["normal"],
                                          shortcut_block=m["sige_block_size"]["instance"],
                                          sparse_threshold=m["sparse_resolution_threshold"],
                                          reference_attn_quirk=True))
    else:
        from models.ddpm_arch.sige_fused_unet import SIGEFusedUNet

        import warnings
        warnings.simplefilter("ignore")
        model = SIGEFusedUNet(None, AttrDict(cfg))
    model.eval()
    if a.stack == "reference":
        torch.save(model.state_dict(), a.state)
    else:
        missing = model.load_state_dict(torch.load(a.state), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys

    rs = np.random.RandomState(7)
    x0 = torch.from_numpy(rs.standard_normal((1, 3, 256, 256)).astype(np.float32))
    side = int(round((a.ratio ** 0.5) * 256))
    mask = torch.zeros(256, 256, dtype=torch.bool)
    mask[100:100 + side, 90:90 + side] = True
    x1 = x0 + torch.from_numpy(rs.standard_normal((1, 3, 256, 256)).astype(np.float32)) * mask
    t = torch.zeros(1)
    with torch.no_grad():
        model.set_mode("full")
        full = model(x0, t)
        masks = downsample_mask(dilate_mask(mask, 5), 256 // 2 ** (len(cfg["model"]["ch_mult"]) - 1))  # runner.py:157-165
        model.set_masks(masks)
        model.set_mode("sparse")
        sparse = model(x1, t)
    np.savez(a.out, full=full.numpy(), sparse=sparse.numpy())
    print("ok", a.stack, float(full.abs().mean()), float(sparse.abs().mean()))


if __name__ == "__main__":
    main()
