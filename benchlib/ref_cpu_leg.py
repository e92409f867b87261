#!/usr/bin/env python3
"""bench.py's `cpu_baseline` leg exactly as BASELINE.md 3 specifies it, for when the reference is mounted (the build container:
SIGE_REFERENCE or /root/reference; never on the GPU box): the reference's UNMODIFIED `sige.nn` package and
`diffusion/models/ddpm_arch/sige_fused_unet.py` on its own compiled sige/cpu backend (oracle/_ref), with the weights, the original
image, the noise and the mask of the GPU run (handed over in --job).  Runs in a process of its own: the reference's `sige` package
and this repo's drop-in alias cannot share an interpreter.  Prints one JSON line: per-forward times and where the sparse output
was saved."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


class AttrDict(dict):
    def __getattr__(self, k):
        v = self[k]
        return AttrDict(v) if isinstance(v, dict) else v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True)
    ap.add_argument("--job", required=True, help="torch file: {state, x0, noise, masks: {ratio: mask}, headline}")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", required=True, help="torch file for {ratio: sparse output}")
    a = ap.parse_args()
    ref = os.path.abspath(a.reference)
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") not in (REPO, HERE)]
    sys.path.insert(0, ref)
    sys.path.insert(1, os.path.join(ref, "diffusion"))
    sys.path.append(REPO)
    os.environ["OMP_NUM_THREADS"] = str(a.threads)

    import torch
    import yaml

    torch.set_num_threads(a.threads)
    from oracle import build_ref

    ref_cpu = build_ref.load()
    import sige

    assert os.path.abspath(sige.__file__).startswith(ref), sige.__file__
    sys.modules["sige.cpu"] = ref_cpu
    sige.cpu = ref_cpu
    import warnings

    from models.ddpm_arch.sige_fused_unet import SIGEFusedUNet
    from sige.utils import dilate_mask, downsample_mask

    warnings.simplefilter("ignore")
    cfg = yaml.safe_load(open(os.path.join(ref, "diffusion", "configs", "church_ddpm256-sige.yml")))
    job = torch.load(a.job)
    model = SIGEFusedUNet(None, AttrDict(cfg)).eval()
    res = model.load_state_dict(job["state"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    x0, noise, t = job["x0"], job["noise"], torch.zeros(1)
    min_res = 256 // 2 ** (len(cfg["model"]["ch_mult"]) - 1)  # diffusion/runner.py:157-165
    outs, times = {}, []
    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        for r, mask in job["masks"].items():
            model.set_masks(downsample_mask(dilate_mask(mask, 5), min_res))
            model.set_mode("sparse")
            outs[r] = model(x0 + noise * mask, t).clone()
        if a.seconds > 0:
            mask = job["masks"][job["headline"]]
            x1 = x0 + noise * mask
            model.set_masks(downsample_mask(dilate_mask(mask, 5), min_res))
            t_begin = time.perf_counter()
            for i in range(25):
                t0 = time.perf_counter()
                model(x1, t)
                if i >= 5:
                    times.append(time.perf_counter() - t0)
                if time.perf_counter() - t_begin > a.seconds and len(times) >= 3:
                    break
    torch.save(outs, a.out)
    print(json.dumps({"times": times, "threads": a.threads, "model": "SIGEFusedUNet", "sige": os.path.dirname(sige.__file__)}))


if __name__ == "__main__":
    main()
