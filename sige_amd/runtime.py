"""Device-type -> native backend registry.

The reference resolves its native functions with
`importlib.import_module("sige.<device>")` for device in cpu/cuda/mps
(sige/nn/base.py:35-50).  Here the only product backend is the HIP library,
registered under "cuda" (the device type of torch-ROCm tensors).  Other device
types have NO backend: calling a sparse op on them raises.  Tests may register
the CPU oracle under "cpu" to exercise the host logic without a GPU.
"""
from types import ModuleType
from typing import Callable, Dict, Optional, Union

_backends: Dict[str, Union[ModuleType, Callable[[], ModuleType]]] = {}


def _load_hip():
    from . import hip

    return hip


_backends["cuda"] = _load_hip


def register_backend(device_type: str, module) -> None:
    """Register an object exposing gather/scatter/... for a torch device type."""
    _backends[device_type] = module


def unregister_backend(device_type: str) -> None:
    _backends.pop(device_type, None)


def get_backend(device_type: str):
    b = _backends.get(device_type)
    if b is None:
        return None
    if callable(b) and not isinstance(b, ModuleType) and not hasattr(b, "gather"):
        b = b()
        _backends[device_type] = b
    return b


def resolve(device_type: str, function_name: str) -> Optional[Callable]:
    b = get_backend(device_type)
    return None if b is None else getattr(b, function_name, None)
