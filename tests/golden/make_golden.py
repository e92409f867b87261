#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (lmxyy/sige v0.3.0).

Run HERE (build container, /root/reference mounted, no GPU):

    python tests/golden/make_golden.py

It imports the reference's own Python (`sige.nn`, `sige.utils`) from
/root/reference and its own CPU backend compiled by oracle/build_ref.py
(oracle/_ref/sige_ref_cpu.so, injected as `sige.cpu`), runs every case of
tests/golden_cases.py through the reference's five native functions +
`F.conv2d` (the stacked-block conv, sige/nn/base.py:89) and stores the OUTPUTS.
Inputs are re-created from seeds by the tests.  /root/reference does not exist
on the GPU box, hence committed fixtures.

Files written:
  ops.npz    per-case reference outputs (index tensors, gather, conv, scatter x3,
             scatter_map, scatter_gather, scatter_with_block_residual)
  masks.npz  reduce_mask / dilate_mask / downsample_mask outputs on the reference's
             own fixture masks (assets/mask.npy, gaugan label diff, SD inpainting mask)
             and on synthetic square edits
  example.npz  example.py's config (Gather -> 3x3 conv -> Scatter through the
             reference's sige.nn modules): active indices and the sparse output
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SIGE_REFERENCE", "/root/reference")

# the reference's `sige` package must win over anything in this repo
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

sys.path.append(REPO)
from oracle import build_ref  # noqa: E402
from tests.golden_cases import CASES, make_inputs  # noqa: E402

build_ref.build(REF, verbose=False)
ref_cpu = build_ref.load()
import sige  # noqa: E402  (the reference package)

assert os.path.abspath(sige.__file__).startswith(REF), sige.__file__
sys.modules["sige.cpu"] = ref_cpu
sige.cpu = ref_cpu
from sige.nn import Gather, Scatter, SIGEConv2d, SIGEModel, SIGEModule  # noqa: E402
from sige.utils import dilate_mask, downsample_mask, reduce_mask  # noqa: E402

T = torch.from_numpy


def opt(a):
    return None if a is None else T(a)


def run_case(case):
    g = case["geom"]
    d = make_inputs(case)
    mask = T(d["mask"])
    idx = reduce_mask(mask, g.block, g.block_stride, g.offset)
    x = T(d["x"])
    out = {"idx": idx.numpy()}
    gathered = ref_cpu.gather(x, g.block[0], g.block[1], idx, opt(d["scale"]), opt(d["shift"]),
                              case["act"], case["act_first"])
    out["gather"] = gathered.numpy()
    conv = F.conv2d(gathered, T(d["weight"]), T(d["bias"]), g.stride, (0, 0))
    out["conv"] = conv.numpy()
    y = T(d["y"])
    args = (g.offset[0], g.offset[1], g.stride[0], g.stride[1], idx)
    out["scatter"] = ref_cpu.scatter(conv, y, *args, None).numpy()
    out["scatter_res"] = ref_cpu.scatter(conv, y, *args, T(d["residual"])).numpy()
    out["scatter_resc"] = ref_cpu.scatter(conv, y, *args, T(d["residual_c"])).numpy()
    Ho, Wo = d["out_res"]
    smap = ref_cpu.get_scatter_map(Ho, Wo, g.block[0], g.block[1], g.kernel[0], g.kernel[1],
                                   g.offset[0], g.offset[1], g.stride[0], g.stride[1], idx)
    out["map"] = smap.numpy()
    out["sg"] = ref_cpu.scatter_gather(conv, y, g.block[0], g.block[1], idx, smap, opt(d["scale2"]),
                                       opt(d["shift2"]), case["act"], case["act_first"]).numpy()
    # shortcut branch (4x4 tiles on their own grid, offset 0, stride 1)
    m1 = d["mask"][:: g.stride[0], :: g.stride[1]][:Ho, :Wo]
    m1 = np.ascontiguousarray(np.pad(m1, ((0, Ho - m1.shape[0]), (0, Wo - m1.shape[1]))))
    idx1 = reduce_mask(T(m1), (4, 4), (4, 4), (0, 0))
    out["idx1"] = idx1.numpy()
    rs = np.random.RandomState(int(d["x1_seed"]))
    x1 = rs.standard_normal((case["B"] * idx1.shape[0], case["cout"], 4, 4)).astype(np.float32)
    out["swbr"] = ref_cpu.scatter_with_block_residual(conv, y, T(x1), T(d["y1"]), *args[:4], idx, idx1).numpy()
    return out


def pack(m):
    return np.packbits(np.asarray(m, dtype=bool), axis=None)


def square_mask(ratio, H=256, W=256, top=100, left=90):
    side = int(round((ratio ** 0.5) * H))
    m = np.zeros((H, W), dtype=bool)
    m[top:top + side, left:left + side] = True
    return m


def main():
    ops = {}
    for case in CASES:
        for k, v in run_case(case).items():
            ops["%s/%s" % (case["name"], k)] = v
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops)

    # ---- mask helpers on the reference's own fixtures -------------------
    masks = {}
    fixtures = {"assets_mask": np.load(os.path.join(REF, "assets", "mask.npy"))}
    gt = np.load(os.path.join(REF, "gaugan", "assets", "gt_label.npy"))
    syn = np.load(os.path.join(REF, "gaugan", "assets", "synthetic_label.npy"))
    fixtures["gaugan_label_diff"] = np.asarray(gt != syn).reshape(gt.shape[-2:])
    sd = np.load(os.path.join(REF, "stable-diffusion", "assets", "inpainting", "masks", "0.npy"))
    fixtures["sd_inpaint"] = np.asarray(sd).reshape(sd.shape[-2:]).astype(bool)
    for r in (0.012, 0.05, 0.15):
        fixtures["square_%g" % r] = square_mask(r)
    geoms = {"b6s4p1": ((6, 6), (4, 4), (1, 1)), "b4s4p0": ((4, 4), (4, 4), (0, 0)),
             "b5s4p0": ((5, 5), (4, 4), (0, 0)), "b5s4p1": ((5, 5), (4, 4), (1, 1))}
    for name, m in fixtures.items():
        m = np.ascontiguousarray(m.astype(bool))
        masks[name + "/mask"] = pack(m)
        masks[name + "/shape"] = np.array(m.shape)
        for gname, (b, s, p) in geoms.items():
            masks["%s/reduce/%s" % (name, gname)] = reduce_mask(T(m), b, s, p).numpy()
        for dil in (1, 2, 5):
            masks["%s/dilate/%d" % (name, dil)] = pack(dilate_mask(T(m), dil).numpy())
        for min_res, dil in ((8, 1), (8, 2), (4, 1)):
            pyr = downsample_mask(T(m), min_res=min_res, dilation=dil)
            for (h, w), pm in pyr.items():
                masks["%s/pyramid/%d_%d/%dx%d" % (name, min_res, dil, h, w)] = pack(pm.numpy())
        # the DDPM runner's recipe (diffusion/runner.py:157-165): dilate 5, pyramid to 8
        dm = dilate_mask(T(m), 5)
        for (h, w), pm in downsample_mask(dm, min_res=8).items():
            masks["%s/ddpm/%dx%d" % (name, h, w)] = pack(pm.numpy())
            if min(h, w) >= 16:
                masks["%s/ddpm_reduce_b6/%dx%d" % (name, h, w)] = reduce_mask(pm, 6, 4, 1).numpy()
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **masks)

    # ---- example.py through the reference's own sige.nn ----
    class ExampleModule(SIGEModule):
        def __init__(self, cin, cout):
            super().__init__()
            self.conv = SIGEConv2d(cin, cout, 3, 1, 1, bias=True)
            self.gather = Gather(self.conv, block_size=6)
            self.scatter = Scatter(self.gather)

        def forward(self, x):
            return self.scatter(self.conv(self.gather(x)))

    class ExampleModel(SIGEModel):
        def __init__(self, cin, cout):
            super().__init__()
            self.m = ExampleModule(cin, cout)

        def forward(self, x):
            return self.m(x)

    ex = {}
    mask = T(fixtures["assets_mask"])
    for cin, cout in ((16, 32),):
        rs = np.random.RandomState(4242)
        orig = rs.standard_normal((1, cin, 256, 256)).astype(np.float32)
        noise = rs.standard_normal((1, cin, 256, 256)).astype(np.float32)
        w = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
        b = rs.standard_normal((cout,)).astype(np.float32)
        edited = orig + noise * fixtures["assets_mask"][None, None]
        model = ExampleModel(cin, cout).eval()
        with torch.no_grad():
            model.m.conv.weight.copy_(T(w))
            model.m.conv.bias.copy_(T(b))
            model.set_mode("full")
            std = model(T(edited))
            model(T(orig))
            model.set_mode("sparse")
            model.set_masks({(256, 256): mask})
            sp = model(T(edited))
        assert torch.isclose(std, sp, atol=1e-4).all()  # example.py:95
        ex["c%d_%d/idx" % (cin, cout)] = model.m.gather.active_indices.numpy()
        # keep the fixture small: the sparse output on a strided sub-grid + full checksum
        ex["c%d_%d/sparse_sub" % (cin, cout)] = sp.numpy()[:, ::4, ::3, ::3].copy()
        ex["c%d_%d/sparse_sum" % (cin, cout)] = np.array(sp.double().sum().item())
    np.savez_compressed(os.path.join(HERE, "example.npz"), **ex)
    for f in ("ops.npz", "masks.npz", "example.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
