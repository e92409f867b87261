"""VERDICT r5 next #1a: the |difference| distribution of the comparison that turned GPUTEST_r05 red
(tests/test_gpu_round2.py test_ddpm_forward_twins_vs_no_twins), per edit ratio, under BOTH full-pass modes (torch / MIOpen vs
the library's exact-fp32 kernels), over --reps repetitions with a fresh full pass each: twins vs no twins, and each of them
against the CPU oracle's sparse forward.  One JSON line per (mode, repetition, ratio)."""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default="gpurun_out/r6_twins_margin.jsonl")
    a = ap.parse_args()
    import bench
    from sige_amd import hip
    from sige_amd.nn import dense
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet, ResBlock
    from tests import util

    hip.lib()
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)  # noqa: E731
    ratios = (0.012, 0.15, 0.05)
    _, wants = util.ddpm_cpu_oracle([bench.edit_mask(r) for r in ratios])
    x0, noise = bench.make_inputs()
    t = torch.zeros(1, device="cuda")
    rows = []
    for mode in ("torch", "native"):
        dense.FULL_PASS_F32_NATIVE = mode == "native"
        for rep in range(a.reps):
            torch.manual_seed(0)
            model = DDPMSparseUNet(DDPMConfig()).eval().to("cuda").to(memory_format=torch.channels_last)
            model.set_scatter_inplace(True)
            blocks = [m for m in model.modules() if isinstance(m, ResBlock)]
            with torch.no_grad():
                model.set_mode("full")
                full = model(cl(x0.to("cuda")), t).clone()
                for ratio, want in zip(ratios, wants):
                    mask = bench.edit_mask(ratio)
                    x1 = cl((x0 + noise * mask).to("cuda"))
                    outs = {}
                    for twins in (False, True):
                        for b in blocks:
                            b.use_twins = twins
                            b._drop_twin_links()
                        model.set_masks(downsample_mask(dilate_mask(mask.to("cuda"), 5), 8))
                        model.set_mode("sparse")
                        for _ in range(4):
                            out = model(x1, t)
                        outs[twins] = out.clone()
                    row = {"full_pass": mode, "rep": rep, "ratio": ratio,
                           "twins_vs_no_twins": float((outs[True] - outs[False]).abs().max()),
                           "no_twins_vs_oracle": float((outs[False].cpu() - want).abs().max()),
                           "twins_vs_oracle": float((outs[True].cpu() - want).abs().max()),
                           "full_checksum": float(full.double().sum())}
                    rows.append(row)
                    print(json.dumps(row), flush=True)
            del model
    dense.FULL_PASS_F32_NATIVE = False
    with open(a.out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    for mode in ("torch", "native"):
        for ratio in ratios:
            v = [r["twins_vs_no_twins"] for r in rows if r["full_pass"] == mode and r["ratio"] == ratio]
            o = [max(r["twins_vs_oracle"], r["no_twins_vs_oracle"]) for r in rows if r["full_pass"] == mode and r["ratio"] == ratio]
            print("%-6s ratio %-5g twins-vs-no-twins min %.3e max %.3e distinct %d | worst vs oracle %.3e" % (
                mode, ratio, min(v), max(v), len(set(v)), max(o)))


if __name__ == "__main__":
    main()
