"""RCCL on the one GPU a box has: a one-rank process group running the collectives of sige_amd/parallel.py (VERDICT r4 missing #4:
"RCCL has never executed").  In a subprocess: a process group is per process, and a hang stays bounded."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_single_rank_runs_the_collectives_of_parallel():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=REPO)
    p = subprocess.run([sys.executable, os.path.join(REPO, "tests", "rccl_single_rank_worker.py"), str(port)], capture_output=True, text=True,
                       timeout=240, env=env, cwd=REPO)
    assert p.returncode == 0, p.stderr[-3000:]
    # (RCCL prints a banner to stdout through C stdio -- it comes out when the worker exits, after the worker's own line)
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert "RCCL version" in p.stdout
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["cache_unchanged"] is True
    assert res["reductions"] == [1.25, 1]
    assert res["choice"]["method_chosen"] in ("broadcast", "recompute")
    assert res["choice"]["methods_ms"]["raises"] is None and "boom" in res["choice"]["errors"]["raises"]
