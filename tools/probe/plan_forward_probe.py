"""The sparse forward under pinned workgroup shapes of the tile convs (hip.conv_force_waves / conv_force_tile): does a global
override beat the per-launch plan?    python tools/probe/plan_forward_probe.py"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    import bench
    import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
    from sige_amd import hip
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).to(dev).eval().to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = (t.to(dev).contiguous(memory_format=torch.channels_last) for t in bench.make_inputs())
    t = torch.zeros(1, device=dev)
    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        m = bench.edit_mask(0.012).to(dev)
        x1 = x0 + noise * m
        model.set_masks(downsample_mask(dilate_mask(m, 5), 8))
        model.set_mode("sparse")
        combos = [(w, mt, nb, 0) for w in (0, 4, 8) for mt, nb in ((0, 0), (16, 1), (16, 2), (32, 1), (32, 2))]
        if os.environ.get("PLAN_PROBE") == "ksplit":
            combos = [(w, 0, 0, ks) for w in (4, 8) for ks in (0, 2, 3, 4)]
        for waves, mt, nb, ks in combos:
            if True:
                hip.conv_force_waves(waves)
                hip.conv_force_tile(mt, nb)
                hip.conv_force_ksplit(ks)
                try:
                    model(x1, t)
                    model(x1, t)
                    g, out = bench.capture(model, x1, t)
                    ms = bench.timed_replays(g, 150, 20, 1) * 1e3 / 150
                    row = {"waves": waves or "auto", "tile": "%dx%d" % (mt, mt * nb) if mt else "auto", "ksplit": ks or "plan", "forward_ms": round(ms, 4)}
                except Exception as e:
                    row = {"waves": waves or "auto", "tile": "%dx%d" % (mt, mt * nb) if mt else "auto", "error": repr(e)[:100]}
                finally:
                    hip.conv_force_waves(0)
                    hip.conv_force_tile(0, 0)
                    hip.conv_force_ksplit(0)
                print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
