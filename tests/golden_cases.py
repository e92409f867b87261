"""Shared case table for the op-level golden vectors.

Used by tests/golden/make_golden.py (runs the REAL reference, here in the build
container) and by the tests (which re-create the same seeded inputs and compare
the oracle / the HIP path with the stored reference outputs).  Inputs come from
numpy's legacy RandomState (bit-stable across numpy versions), so only the
reference OUTPUTS are stored in tests/golden/ops.npz.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np


class Geometry:
    """Tile geometry derived the way Gather.__init__ does (sige/nn/gather.py:26-43)."""

    def __init__(self, kernel: Tuple[int, int], stride: Tuple[int, int], padding: Tuple[int, int],
                 block: Tuple[int, int], offset: Optional[Tuple[int, int]] = None):
        self.kernel, self.stride, self.padding = kernel, stride, padding
        n = [max(block[i] - kernel[i], 0) // stride[i] for i in (0, 1)]
        self.block = tuple(n[i] * stride[i] + kernel[i] for i in (0, 1))
        self.block_stride = tuple((n[i] + 1) * stride[i] for i in (0, 1))
        self.offset = padding if offset is None else offset
        self.out_tile = tuple(n[i] + 1 for i in (0, 1))


def make_mask(kind: str, H: int, W: int, rs: np.random.RandomState) -> np.ndarray:
    m = np.zeros((H, W), dtype=bool)
    if kind == "empty":
        pass
    elif kind == "full":
        m[:] = True
    elif kind == "pixel":
        m[H // 2, W // 3] = True
    elif kind == "corners":  # tiles touching all four borders (idx -1 and H-b+...)
        m[0, 0] = m[0, W - 1] = m[H - 1, 0] = m[H - 1, W - 1] = True
        m[H // 2, W // 2] = True
    elif kind == "blobs":
        for _ in range(3):
            h0, w0 = rs.randint(0, H), rs.randint(0, W)
            hh, ww = rs.randint(1, max(2, H // 3)), rs.randint(1, max(2, W // 3))
            m[h0:h0 + hh, w0:w0 + ww] = True
    elif kind == "random":
        m = rs.rand(H, W) < 0.05
    else:
        raise ValueError(kind)
    return m


# (name, B, C, H, W, geometry, mask kind, scale kind, shift kind, act, act_first, shortcut?)
# scale/shift kinds: None | "c" [1,C,1,1] | "bc" [B,C,1,1] | "full" [1,C,H,W] | "bfull" [B,C,H,W]
def _g(k, s, p, b, off=None):
    return Geometry((k, k) if isinstance(k, int) else k, (s, s) if isinstance(s, int) else s,
                    (p, p) if isinstance(p, int) else p, (b, b) if isinstance(b, int) else b, off)


CASES: List[Dict] = []


def _add(name, B, C, H, W, geom, mask, scale=None, shift=None, act="identity", act_first=False, cout=None):
    CASES.append(dict(name=name, B=B, C=C, H=H, W=W, geom=geom, mask=mask, scale=scale, shift=shift,
                      act=act, act_first=act_first, cout=cout or C, seed=1000 + len(CASES)))


_add("main3x3_blobs", 1, 4, 24, 28, _g(3, 1, 1, 6), "blobs")
_add("main3x3_corners_affine_swish", 1, 5, 24, 24, _g(3, 1, 1, 6), "corners", "c", "c", "swish")
_add("main3x3_b2_bc_swish", 2, 3, 20, 28, _g(3, 1, 1, 6), "blobs", "bc", "bc", "swish")
_add("main3x3_full_affine", 1, 3, 16, 20, _g(3, 1, 1, 6), "full", "full", "full", "identity")
_add("main3x3_scale_only", 1, 3, 22, 26, _g(3, 1, 1, 6), "random", "c", None, "identity")
_add("main3x3_shift_only_actfirst", 1, 3, 22, 26, _g(3, 1, 1, 6), "random", None, "c", "swish", True)
_add("main3x3_bfull_actfirst", 2, 2, 18, 18, _g(3, 1, 1, 6), "blobs", "bfull", "bfull", "swish", True)
_add("main3x3_ragged_hw", 1, 4, 30, 37, _g(3, 1, 1, 6), "corners", "c", "c", "swish")
_add("main3x3_empty", 1, 3, 16, 16, _g(3, 1, 1, 6), "empty", "c", "c", "swish")
_add("main3x3_pixel", 1, 3, 16, 16, _g(3, 1, 1, 6), "pixel")
_add("short1x1_blobs", 1, 6, 24, 28, _g(1, 1, 0, 4), "blobs", cout=4)
_add("short1x1_corners_b2", 2, 3, 21, 23, _g(1, 1, 0, 4), "corners", "c", "c", "identity")
_add("down3x3s2p0", 1, 4, 24, 28, _g(3, 2, 0, 6), "blobs")  # block 6 -> 5 (gather.py:26-31)
_add("down3x3s2p0_corners", 1, 3, 26, 22, _g(3, 2, 0, 6), "corners", "c", "c", "swish")
_add("down3x3s2p1", 2, 3, 24, 24, _g(3, 2, 1, 6), "blobs")
_add("down3x3s2p1_ragged", 1, 3, 27, 25, _g(3, 2, 1, 6), "corners")
_add("nonsquare_block_6x4", 1, 3, 24, 24, _g(3, 1, 1, (6, 4)), "blobs", "c", "c", "swish")
_add("big_block_10", 1, 2, 32, 32, _g(3, 1, 1, 10), "blobs")
_add("k5_block_8", 1, 2, 24, 24, _g(5, 1, 2, 8), "corners", "c", None, "swish")
_add("main3x3_wide_c", 1, 70, 16, 16, _g(3, 1, 1, 6), "blobs", "c", "c", "swish", cout=8)


def bcast_shape(kind, B, C, H, W):
    return {None: None, "c": (1, C, 1, 1), "bc": (B, C, 1, 1), "full": (1, C, H, W), "bfull": (B, C, H, W)}[kind]


def make_inputs(case: Dict) -> Dict[str, np.ndarray]:
    """Seeded inputs of one case (everything except index tensors, which come from
    reduce_mask on `mask`)."""
    rs = np.random.RandomState(case["seed"])
    B, C, H, W, g = case["B"], case["C"], case["H"], case["W"], case["geom"]
    f = lambda *s: rs.standard_normal(s).astype(np.float32)  # noqa: E731
    d = {"mask": make_mask(case["mask"], H, W, rs), "x": f(B, C, H, W)}
    for key in ("scale", "shift"):
        shp = bcast_shape(case[key], B, C, H, W)
        d[key] = None if shp is None else f(*shp)
    Cout = case["cout"]
    d["weight"] = (f(Cout, C, *g.kernel) / np.sqrt(C * g.kernel[0] * g.kernel[1])).astype(np.float32)
    d["bias"] = f(Cout)
    Ho = (H + 2 * g.padding[0] - g.kernel[0]) // g.stride[0] + 1
    Wo = (W + 2 * g.padding[1] - g.kernel[1]) // g.stride[1] + 1
    d["out_res"] = (Ho, Wo)
    d["y"] = f(B, Cout, Ho, Wo)          # cached original conv output
    d["residual"] = f(B, Cout, Ho, Wo)   # full residual
    d["residual_c"] = f(1, Cout, 1, 1)   # broadcast residual
    # second affine (applied by scatter_gather on the conv-output channels)
    for key in ("scale", "shift"):
        shp = bcast_shape(case[key], B, Cout, Ho, Wo)
        d[key + "2"] = None if shp is None else f(*shp)
    # shortcut branch for scatter_with_block_residual (1x1 conv tiles on its own 4x4 grid)
    d["y1"] = f(B, Cout, Ho, Wo)
    d["x1_seed"] = np.int64(case["seed"] + 7)
    return d
