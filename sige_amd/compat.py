"""Make `import sige` resolve to this package.

The reference's model files (diffusion/models/ddpm_arch/sige_fused_unet.py,
gaugan/models/**/sige_*.py, stable-diffusion/ldm/modules/**/sige_*.py) import
`sige.nn` / `sige.utils`.  After `sige_amd.compat.install()` those imports get
the MI355X implementation, and `sige.cuda` is the HIP backend module with the
reference's five native functions -- so even the reference's own `sige/nn`
Python would find its "cuda" runtime here.

Alternative without code changes: put `sige_amd/dropin` on PYTHONPATH (it holds
a three-line `sige` package that calls install()).
"""
import sys
import types


def install(force: bool = False) -> None:
    """Register this package as `sige`.  Refuses (RuntimeError) when a different `sige` package is already imported,
    unless `force=True` (the foreign package's modules are then replaced for the rest of the process)."""
    import sige_amd
    from sige_amd import nn, utils
    from sige_amd.nn import base, gather, scatter, scatter_gather
    from sige_amd.nn import utils as nn_utils

    if "sige" in sys.modules and not force and not getattr(sys.modules["sige"], "__sige_amd__", False):
        raise RuntimeError("another `sige` package is already imported")

    pkg = types.ModuleType("sige")
    pkg.__dict__.update(__version__=sige_amd.__version__, __path__=[], __sige_amd__=True, nn=nn, utils=utils)
    mods = {
        "sige": pkg,
        "sige.nn": nn,
        "sige.nn.base": base,
        "sige.nn.gather": gather,
        "sige.nn.scatter": scatter,
        "sige.nn.scatter_gather": scatter_gather,
        "sige.nn.utils": nn_utils,
        "sige.utils": utils,
    }

    class _LazyHip(types.ModuleType):
        """`sige.cuda`: resolves to sige_amd.hip on first attribute access, so
        installing the alias does not require the built library."""

        def __getattr__(self, name):
            from sige_amd import hip

            return getattr(hip, name)

    cuda = _LazyHip("sige.cuda")
    pkg.cuda = cuda
    mods["sige.cuda"] = cuda
    sys.modules.update(mods)


def uninstall() -> None:
    """Remove the alias again (only if the installed `sige` is ours: a foreign package is left alone)."""
    if not getattr(sys.modules.get("sige"), "__sige_amd__", False):
        return
    for name in [n for n in sys.modules if n == "sige" or n.startswith("sige.")]:
        sys.modules.pop(name, None)
