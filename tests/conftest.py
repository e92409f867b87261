import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")
    config.addinivalue_line("markers", "oracle_parity: compares with the reference's outputs through a helper the collection-order "
                                       "rule cannot see in the test's own source (tier 0, like the tests that name the oracle)")
    config.addinivalue_line("markers", "selfcheck: a HIP path against another HIP path / a stress run -- collected LAST, after "
                                       "every oracle-parity test of the SURVEY.md 8 rows (tests/rows.py)")


# what marks a test as pinned to the CPU oracle / the reference's golden vectors: its own source or one of its fixtures
_ORACLE_WORDS = ("oracle.", "util.golden(", "util.ref(", "ddpm_cpu_oracle(", "cpu_backend(", "_cpu_reference_backend(",
                 "register_backend(", "golden(")
_ORACLE_FIXTURES = ("ddpm_reference",)


def tier(item):
    """Collection order of the GPU run (`pytest -x -q -m gpu` stops at the first failure, so what must not be blanked goes first):
    0 = compares with the CPU oracle or the committed golden vectors (the row-defining tests of SURVEY.md 8),
    1 = everything else (torch / fp64 references of single launches, host logic on the GPU),
    2 = `selfcheck`: HIP-vs-HIP comparisons of whole forwards and stress runs."""
    import inspect

    if item.get_closest_marker("selfcheck") is not None:
        return 2
    if item.get_closest_marker("oracle_parity") is not None:
        return 0
    fn = getattr(item, "function", None)
    try:
        src = inspect.getsource(fn) if fn is not None else ""
    except (OSError, TypeError):
        src = ""
    if any(w in src for w in _ORACLE_WORDS) or any(f in getattr(item, "fixturenames", ()) for f in _ORACLE_FIXTURES):
        return 0
    return 1


def pytest_collection_modifyitems(config, items):
    import torch

    items.sort(key=tier)  # (stable: file / definition order inside a tier)
    dump = os.environ.get("SIGE_DUMP_ORDER")
    if dump:  # (tests/test_collection_order.py reads the order back)
        import json

        with open(dump, "w") as f:
            for item in items:
                f.write(json.dumps({"id": item.nodeid, "tier": tier(item), "gpu": "gpu" in item.keywords}) + "\n")

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
