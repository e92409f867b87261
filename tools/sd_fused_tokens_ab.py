"""The SD U-Net's sparse forward (bench.py --workload sd: same model, latent, context, mask) with what sits between the token GEMMs
of a transformer block as library launches (sd_transformer.FUSED_TOKENS: residual add + bias + LayerNorm, GEGLU, last add) or as the
torch kernels the reference runs, hipGraph replay each, alternating, in one process.

    python tools/sd_fused_tokens_ab.py --out gpurun_out/sd_fused_tokens.json
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from benchlib.common import _replay_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--settings", default="0,1,0,1")
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from sige_amd import hip
    from sige_amd.utils import downsample_mask
    from sige_amd.workloads import sd_unet
    from sige_amd.workloads.sd_unet import SDConfig, SDUNet

    dev = torch.device("cuda:0")
    hip.lib()
    torch.manual_seed(0)
    model = SDUNet(SDConfig()).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    gen = torch.Generator().manual_seed(1)
    cl = lambda t_: t_.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x0, noise = cl(torch.randn(2, 4, 64, 64, generator=gen)), cl(torch.randn(2, 4, 64, 64, generator=gen))
    ctx = torch.randn(2, 77, 768, generator=gen).to(dev)
    ts = torch.full((2,), 500.0, device=dev)
    mask512 = torch.zeros(512, 512, dtype=torch.bool, device=dev)
    mask512[150:348, 120:318] = True
    masks = downsample_mask(mask512, min_res=8, dilation=1)
    x1 = cl(x0 + noise * masks[(64, 64)])
    run = lambda x: model(x, ts, context=ctx)  # noqa: E731
    rows, ref = [], None
    with torch.no_grad():
        model.set_mode("full")
        run(x0)
        model.set_masks(masks)
        model.set_mode("sparse")
        from sige_amd.workloads import sd_transformer
        for st in args.settings.split(","):
            sd_transformer.FUSED_TOKENS = bool(int(st))
            run(x1)
            n0 = hip.launch_count()
            run(x1)
            launches = hip.launch_count() - n0
            ms, out, g = _replay_ms(lambda: run(x1), k=args.replays, warm=3)
            o = out.float().clone()
            del g
            if ref is None:
                ref = o
            rows.append({"fused_tokens": bool(int(st)), "forward_ms": round(ms, 3), "library_launches": launches,
                         "max_abs_vs_first_setting": round(float((o - ref).abs().max()), 8)})
            print(json.dumps(rows[-1]), flush=True)
        sd_transformer.FUSED_TOKENS = False
    res = {"workload": "bench.py --workload sd (SD v1 U-Net, latent [2,4,64,64], 15 % edit), hipGraph replay", "rows": rows}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
