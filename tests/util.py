"""Helpers shared by the tests: golden loading, seeded inputs as torch tensors."""
import os

import numpy as np
import torch

from tests.golden_cases import CASES, make_inputs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
    return _cache[name]


def case_ids():
    return [c["name"] for c in CASES]


def case_by_name(name):
    return next(c for c in CASES if c["name"] == name)


def tensors(case, device="cpu"):
    """Seeded inputs of a case as torch tensors on `device` (None stays None)."""
    d = make_inputs(case)
    out = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray) and v.dtype != np.int64:
            out[k] = torch.from_numpy(v).to(device)
        else:
            out[k] = v
    return out


def ref(case, key, device="cpu"):
    return torch.from_numpy(golden("ops")["%s/%s" % (case["name"], key)]).to(device)


def x1_tiles(case, d, n1, device="cpu"):
    rs = np.random.RandomState(int(d["x1_seed"]))
    a = rs.standard_normal((case["B"] * n1, case["cout"], 4, 4)).astype(np.float32)
    return torch.from_numpy(a).to(device)


def shortcut_mask(case, d):
    g = case["geom"]
    Ho, Wo = d["out_res"]
    m = d["mask"].cpu().numpy()[:: g.stride[0], :: g.stride[1]][:Ho, :Wo]
    m = np.ascontiguousarray(np.pad(m, ((0, Ho - m.shape[0]), (0, Wo - m.shape[1]))))
    return torch.from_numpy(m)


def unpack(bits, shape):
    n = int(shape[0]) * int(shape[1])
    return torch.from_numpy(np.unpackbits(bits)[:n].reshape(int(shape[0]), int(shape[1])).astype(bool))


SWISH_RTOL = 1e-6  # swish <= 1e-6 rel (reference computes it in mixed float/double)
SWISH_ATOL = 1e-7
CONV_ATOL = 1e-3   # north_star: activations within 1e-3 fp32 on conv-containing paths
