import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")


def pytest_collection_modifyitems(config, items):
    import torch

    has_gpu = torch.cuda.is_available()
    has_ref = os.path.isdir(os.environ.get("SIGE_REFERENCE", "/root/reference"))
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU in this container"))
        if "reference" in item.keywords and not has_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not mounted"))


@pytest.fixture
def tuning():
    """Run the test on the measurement build (lib/libsige_hip_tuning.so), the only library with dispatch knobs
    (include/sige_hip.h: sige_hip_tuning_set); knobs are reset and the product library restored afterwards."""
    from sige_amd import build, hip

    build.build_tuning(verbose=False)
    with hip.tuning_build():
        yield hip
