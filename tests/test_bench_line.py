"""The driver-facing bench line (benchlib/line.py): built from a canned full result -- the committed 26 KB round-4 line, which the
driver could not parse -- the final line stays under the limit and carries the contract keys; the detail goes to a file and to an
EARLIER stdout line."""
import io
import json
import os

import pytest

from benchlib import line as bl

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def canned(name):
    path = os.path.join(REPO, "profiles", name)
    if not os.path.isfile(path):
        pytest.skip("no profiles/%s" % name)
    return json.loads(open(path).read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["r4_bench.json", "r4_bench_f16.json", "r4_bench_2ranks_gloo.json", "r4_bench_sd.json"])
def test_compact_line_is_small_and_complete(name):
    full = canned(name)
    text = json.dumps(bl.compact(full))
    assert len(text) < bl.MAX_BYTES < 8192
    got = json.loads(text)
    for k in bl.CONTRACT_KEYS:
        if k in full:
            assert k in got, k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "dtype", "config"):
        assert got[k] == full[k] or k == "config"
    assert "workload" in got["config"]
    if "roofline" in full:
        r = got["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac"):
            assert k in r
        assert "traffic" in r or full["roofline"].get("traffic") is None
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 5e-3
    if "cpu_baseline" in full:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in got["cpu_baseline"]


def test_hbm_rooflines_lead_with_counter_bytes():
    full = canned("r4_bench.json")
    got = bl.compact(full)
    h = got["roofline_hbm"]
    assert h["bytes"] == "counter" and h["frac"] == full["roofline_hbm"]["frac_on_counter_bytes"]
    assert h["frac_alg"] == full["roofline_hbm"]["frac"] and h["frac"] <= h["frac_alg"]


def test_oversized_sections_are_dropped_not_truncated():
    full = canned("r4_bench.json")
    full["sweep"] = [dict(full["sweep"][0], edit_ratio=i / 100.0) for i in range(5)]
    full["config"]["workload"] = "x" * 5000
    full["batched_edits"]["rows"] = [dict(full["batched_edits"]["rows"][0], edits=e) for e in range(1, 40)]
    text = json.dumps(bl.compact(full))
    assert len(text) < bl.MAX_BYTES
    got = json.loads(text)
    for k in ("metric", "value", "roofline", "cpu_baseline"):
        assert k in got


def test_emit_prints_the_compact_line_last(tmp_path):
    full = canned("r4_bench.json")
    out = io.StringIO()
    bl.emit(full, stream=out, detail_dirs=[str(tmp_path)])
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 2
    detail = json.loads(lines[0])["bench_detail"]
    assert detail["batched_edits"] == full["batched_edits"] and detail["data_movement"] == full["data_movement"]
    last = json.loads(lines[-1])
    assert len(lines[-1]) < bl.MAX_BYTES and last["metric"] == full["metric"] and last["detail"] == "bench_detail.json"
    on_disk = json.load(open(tmp_path / "bench_detail.json"))
    assert on_disk["gaugan"] == full["gaugan"]
    # the tail the driver keeps (8 KB) contains the whole last line
    tail = out.getvalue()[-8192:]
    assert json.loads(tail.strip().splitlines()[-1]) == last


def test_result_lines_are_the_only_stdout_whatever_libraries_print(tmp_path):
    """bench.py reserves the process's stdout for its result lines (benchlib.line.reserve_stdout): what a library prints to stdout
    afterwards -- RCCL's banner sits in a C stdio buffer until the process exits, i.e. AFTER the contract line -- lands on stderr, so
    the last stdout line is the contract line for every N."""
    import subprocess
    import sys

    script = r'''
import ctypes, json, sys
sys.path.insert(0, %r)
from benchlib import line
canned = json.loads(open(%r).read().strip().splitlines()[-1])
line.reserve_stdout()
libc = ctypes.CDLL(None)
libc.printf(b"Librccl path : /somewhere/librccl.so\n")   # buffered C stdio: flushed at exit
print("python noise")
line.emit(canned, detail_dirs=[%r], name="detail.json")
libc.printf(b"more noise at the end\n")
''' % (REPO, os.path.join(REPO, "profiles", "r4_bench.json"), str(tmp_path))
    p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout.strip().splitlines()
    assert len(out) == 2 and "bench_detail" in json.loads(out[0])
    last = json.loads(out[-1])
    assert last["metric"] and len(out[-1]) < 4096
    assert "Librccl path" in p.stderr and "python noise" in p.stderr and "more noise" in p.stderr


def test_contract_line_is_last_even_when_the_caller_merges_the_streams():
    """... and for a caller that reads stdout and stderr as ONE stream, emit() first flushes what sits in C stdio buffers (RCCL's
    banner is written when the communicator is created, long before the result): the contract line still ends the stream."""
    import subprocess
    import sys

    script = r'''
import ctypes, json, sys
sys.path.insert(0, %r)
from benchlib import line
canned = json.loads(open(%r).read().strip().splitlines()[-1])
line.reserve_stdout()
ctypes.CDLL(None).printf(b"Librccl path : /somewhere/librccl.so\n")
print("python noise")
line.emit(canned, detail_dirs=[])
''' % (REPO, os.path.join(REPO, "profiles", "r4_bench.json"))
    p = subprocess.run([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-2000:]
    out = p.stdout.strip().splitlines()
    assert any("Librccl path" in ln for ln in out[:-2])
    assert json.loads(out[-1])["metric"] and "bench_detail" in json.loads(out[-2])
