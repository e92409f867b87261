"""Which torch (ATen) ops still run inside the cache-producing full pass, and from which line of the package?

    python tools/full_pass_ops.py [--dtype f16x3] [--out gpurun_out/full_pass_ops.json]

Counts every ATen call made during ONE full-mode forward of the DDPM-256 U-Net (TorchDispatchMode), grouped by op and by the
innermost sige_amd/ source line on the Python stack -- the to-do list for moving the full pass onto the library's kernels."""
import argparse
import collections
import json
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


class Counter(TorchDispatchMode):
    def __init__(self, depth=1):
        super().__init__()
        self.depth = depth
        self.rows = collections.Counter()
        self.elems = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in ("view", "reshape", "expand", "unbind", "slice", "select", "transpose", "permute", "detach",
                                   "alias", "unsqueeze", "squeeze", "as_strided", "t.default", "sym_", "_unsafe_view")):
            return out
        frames = ["%s:%d" % (os.path.relpath(fr.filename, REPO), fr.lineno) for fr in reversed(traceback.extract_stack())
                  if "/sige_amd/" in fr.filename]
        site = " < ".join(frames[:self.depth]) if frames else "?"
        n = out.numel() if isinstance(out, torch.Tensor) else 0
        self.rows[(name, site)] += 1
        self.elems[(name, site)] += n
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16x3")
    ap.add_argument("--out", default="")
    ap.add_argument("--depth", type=int, default=1, help="package frames per call site")
    a = ap.parse_args()
    import bench
    from sige_amd import hip
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).to(dev).eval().to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    model.set_compute_dtype(a.dtype)
    x0, _ = (t.to(dev).contiguous(memory_format=torch.channels_last) for t in bench.make_inputs())
    t = torch.zeros(1, device=dev)
    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        model(x0, t)
        n0 = hip.launch_count()
        with Counter(a.depth) as c:
            model(x0, t)
        launches = hip.launch_count() - n0
    rows = [{"op": k[0], "site": k[1], "calls": v, "out_MB": round(c.elems[k] * 4 / 1e6, 2)} for k, v in c.rows.most_common()]
    print(json.dumps({"library_launches": launches, "aten_calls": sum(r["calls"] for r in rows)}))
    for r in rows:
        print("%4d  %9.2f MB  %-42s %s" % (r["calls"], r["out_MB"], r["op"], r["site"]))
    if a.out:
        os.makedirs(os.path.dirname(os.path.join(REPO, a.out)) or ".", exist_ok=True)
        with open(os.path.join(REPO, a.out), "w") as f:
            json.dump({"library_launches": launches, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
