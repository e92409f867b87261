import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
from sige_amd import hip
from sige_amd.utils import reduce_mask
hip.lib()
dev = torch.device("cuda", 0)
B, C = 2, 256
mask = bench.square_mask(0.15).to(dev)
idx6 = reduce_mask(mask, 6, 4, 1); n6 = idx6.shape[0]
smap = hip.get_scatter_map(256, 256, 6, 6, 3, 3, 1, 1, 1, 1, idx6)
nsets = 8
ys = [torch.randn(B, C, 256, 256, device=dev) for _ in range(nsets)]
t4 = [torch.randn(B * n6, C, 4, 4, device=dev) for _ in range(nsets)]
sc, sh = torch.randn(1, C, 1, 1, device=dev), torch.randn(1, C, 1, 1, device=dev)
nbytes = 2 * 4 * B * n6 * C * 36
def run(name, knob, f):
    hip.scatter_gather_force_elements(knob)
    it = [0]
    def rot():
        f(it[0] % nsets); it[0] += 1
    us = bench.time_graph_of(rot, reps=nsets * 2)
    hip.scatter_gather_force_elements(0)
    print(json.dumps({"case": name, "us": round(us, 2), "frac": round(nbytes / us / 1e3 / 8000, 4)}), flush=True)
sw = lambda i: hip.scatter_gather(t4[i], ys[i], 6, 6, idx6, smap, sc, sh, "swish", False)
idn = lambda i: hip.scatter_gather(t4[i], ys[i], 6, 6, idx6, smap, sc, sh, "identity", False)
raw = lambda i: hip.scatter_gather(t4[i], ys[i], 6, 6, idx6, smap)
run("elements swish", 1, sw)
run("rows swish", 2, sw)
run("grouped swish", 0, sw)
run("grouped raw", 0, raw)
