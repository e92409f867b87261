"""The stated tolerances of the path, in ONE place.

    fp32 conv-containing paths : |got - ref| <= 1e-3                         (north_star)
    f16 compute / f16 storage  : |got - ref| <= 2e-2 + 1e-2 * |ref|          (per element, against the fp32 reference)

`bench.py` and every `-m gpu` test of the f16 path call `f16_check`, so there is exactly one criterion
(round 2 had an absolute-only check in bench.py and an abs + rel check in the tests).
"""
import torch

CONV_ATOL = 1e-3
F16_ATOL, F16_RTOL = 2e-2, 1e-2
F16_CRITERION = "|got - ref| <= %g + %g * |ref| per element, ref = the fp32 reference (sige/cpu + fp32 convs)" % (F16_ATOL, F16_RTOL)


def f16_check(got: torch.Tensor, ref: torch.Tensor) -> dict:
    """Numbers of the f16 criterion for one output: `ok`, the worst element's |delta| / allowed (<= 1 passes), max |delta|,
    max |ref| and max |delta| / max |ref|."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    delta = (got - ref).abs()
    allowed = F16_ATOL + F16_RTOL * ref.abs()
    worst = float((delta / allowed).max())
    mref = float(ref.abs().max())
    return {"ok": bool(worst <= 1.0), "worst_over_allowed": round(worst, 4), "max_abs": round(float(delta.max()), 6),
            "max_ref": round(mref, 4), "max_abs_over_max_ref": round(float(delta.max()) / max(mref, 1e-30), 7)}
