"""bench.py's data-movement section; the NCHW scatter_gather also in its element form and its one-tile row form.

    python tools/sg_bench.py [--out gpurun_out/sg_bench.jsonl]
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import bench
    import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
    from sige_amd import hip

    hip.lib()
    dev = torch.device("cuda", 0)
    rows = []
    for form, knob in (("elements", 1), ("rows", 2), ("automatic", 0)):
        hip.scatter_gather_force_elements(knob)
        try:
            res = bench.data_movement_rooflines(hip, dev)
        finally:
            hip.scatter_gather_force_elements(False)
        for r in res["data_movement"]:
            if form == "automatic" or (r["layout"] == "nchw" and r["op"].startswith("scatter_gather")):
                r = dict(r, scatter_gather_form=form)
                print(json.dumps(r), flush=True)
                rows.append(r)
    if args.out:
        with open(os.path.join(REPO, args.out), "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
