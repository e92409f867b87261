"""The sparse forward under a pinned cross-workgroup K split of the dense layers (hip.conv_force_ksplit): is the automatic
choice the best one?    python tools/probe/ksplit_forward_probe.py"""
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
        for ks in (0, 1, 2, 3, 4, 6, 8):
            hip.conv_force_ksplit(ks)
            try:
                model(x1, t)
                model(x1, t)
                g, out = bench.capture(model, x1, t)
                ms = bench.timed_replays(g, 200, 20, 1) * 1e3 / 200
            finally:
                hip.conv_force_ksplit(0)
            print(json.dumps({"ksplit": ks or "auto", "forward_ms": round(ms, 4)}), flush=True)
            del g, out


if __name__ == "__main__":
    main()
