"""CPU test of the GPU run's collection order (VERDICT r5 next #1d): oracle-parity tests of every SURVEY.md 8 row first, HIP-vs-HIP
self-checks and stress runs last."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _order(tmp_path):
    dump = str(tmp_path / "order.jsonl")
    env = dict(os.environ, SIGE_DUMP_ORDER=dump)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"],
                       cwd=REPO, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [json.loads(line) for line in open(dump)]
    return [r_ for r_ in rows if r_["gpu"]]


def test_gpu_collection_order_puts_row_defining_tests_first(tmp_path):
    from tests.rows import ROW_TESTS

    items = _order(tmp_path)
    assert len(items) >= 462
    tiers = [it["tier"] for it in items]
    assert tiers == sorted(tiers), "the GPU run is not collected in tier order"
    assert tiers.count(0) >= 100 and 3 <= tiers.count(2) <= 40
    first_other = tiers.index(1)
    by_fn = {}
    for pos, it in enumerate(items):
        name = it["id"].split("tests/")[-1].split("[")[0]
        by_fn.setdefault(name, []).append((pos, it["tier"]))
    for row, names in ROW_TESTS.items():
        for n in names:
            assert n in by_fn, "%s: %s is not a collected GPU test" % (row, n)
            assert all(t == 0 and pos < first_other for pos, t in by_fn[n]), "%s: %s is not collected before the non-oracle tests" % (row, n)
    # every self-check sits behind every other test
    last_non_self = max(pos for pos, t in enumerate(tiers) if t != 2)
    assert all(pos > last_non_self for pos, t in enumerate(tiers) if t == 2)
    # the test that turned GPUTEST_r05 red is one of them
    assert by_fn["test_gpu_round2.py::test_ddpm_forward_twins_vs_no_twins"][0][1] == 2
