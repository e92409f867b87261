"""The committed evidence is reproducible from what is committed: the traffic tables bench.py reads can be re-derived from the raw
counter rows kept next to them, and the bench lines under profiles/ carry the fields of the contract.  (CPU only.)"""
import gzip
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(REPO, "profiles")


@pytest.mark.parametrize("suffix", ["", "_f16"])
def test_traffic_table_rederives_from_the_raw_counter_rows(tmp_path, suffix):
    """tools/pmc_traffic.py over the kept FETCH_SIZE / WRITE_SIZE rows of the conv kernels (profiles/r4_pmc_*_rows*.csv.gz) and
    the manifest gives the per-family figures of profiles/pmc_traffic*.json (what `roofline.traffic` prints)."""
    table = os.path.join(PROF, "pmc_traffic%s.json" % suffix)
    tag = next((t for t in ("r6", "r5") if os.path.exists(os.path.join(PROF, "%s_pmc_fetch_rows%s.csv.gz" % (t, suffix)))), "r4")
    rows = {c: os.path.join(PROF, "%s_pmc_%s_rows%s.csv.gz" % (tag, c, suffix)) for c in ("fetch", "write")}
    manifest = os.path.join(PROF, "%s_manifest%s.json" % (tag, suffix))
    for f in [table, manifest, *rows.values()]:
        if not os.path.exists(f):
            pytest.skip("%s not committed" % os.path.basename(f))
    want = json.load(open(table))
    if want["source_hash"] != json.load(open(manifest))["source_hash"]:
        pytest.skip("the traffic table was re-measured after these rows were kept")
    plain = {}
    for c, path in rows.items():
        plain[c] = str(tmp_path / (c + ".csv"))
        with gzip.open(path, "rt") as src, open(plain[c], "w") as dst:
            dst.write(src.read())
    out = str(tmp_path / "traffic.json")
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_traffic.py"), plain["fetch"], plain["write"], manifest, out],
                   check=True, capture_output=True)
    got = json.load(open(out))
    for fam in ("block_conv_mfma", "dense_conv_mfma"):
        assert got["families"][fam] == want["families"][fam]
    assert got["source_hash"] == want["source_hash"]


@pytest.mark.parametrize("name", ["r4_bench.json", "r4_bench_f16.json", "r4_bench_2ranks_gloo.json", "r4_bench_sd.json"])
def test_bench_lines_carry_the_contract(name):
    path = os.path.join(PROF, name)
    if not os.path.exists(path):
        pytest.skip("%s not committed" % name)
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    if name in ("r4_bench.json", "r4_bench_f16.json"):
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] < 1.0
        assert r["traffic"] is None or r["traffic"] > 0
    if name == "r4_bench.json":
        c = d["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
        assert d["parity_ok"] is True


def test_design_tables_are_generated_from_the_final_evidence():
    """DESIGN.md's result tables ARE tools/design_tables.py over profiles/<TAG>_bench*.json (the document
    quotes the final evidence files and nothing else)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    try:
        import design_tables
    finally:
        sys.path.pop(0)
    want = design_tables.block()
    if want is None:
        pytest.skip("profiles/%s_bench.json / _bench_detail.json not committed yet" % design_tables.TAG)
    text = open(os.path.join(REPO, "DESIGN.md")).read()
    i, j = text.index(design_tables.BEGIN), text.index(design_tables.END)
    assert text[i + len(design_tables.BEGIN):j].strip() == want.strip()


@pytest.mark.parametrize("name", ["r5_bench.json", "r5_bench_f16.json", "r5_bench_2ranks_gloo.json", "r5_bench_sd.json",
                                  "r6_bench.json", "r6_bench_f16.json", "r6_bench_2ranks_gloo.json", "r6_bench_sd.json"])
def test_round5_bench_lines_are_small_and_carry_the_contract(name):
    path = os.path.join(PROF, name)
    if not os.path.exists(path):
        pytest.skip("%s not committed" % name)
    text = open(path).read().strip().splitlines()[-1]
    assert len(text) < 4096
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"] and d["value"] > 0
    if name[3:] in ("bench.json", "bench_f16.json"):
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] < 1.0
    if name[3:] == "bench.json":
        assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["cores"] >= 1 and d["parity_ok"] is True
        assert os.path.exists(os.path.join(PROF, name[:3] + "bench_detail.json"))
        if name.startswith("r6"):
            assert d["cpu_baseline"]["model"]  # (which model ran on the CPU leg: VERDICT r5 next #8)


def test_tuned_gemm_table_is_a_gfx950_tunableop_file():
    """sige_amd/workloads/tunableop_sd_gfx950.csv: validator lines first (PyTorch checks them against the running stack), then
    fp32 GEMM rows only."""
    path = os.path.join(REPO, "sige_amd", "workloads", "tunableop_sd_gfx950.csv")
    rows = [l.strip().split(",") for l in open(path) if l.strip()]
    val = {r[1]: r[2] for r in rows if r[0] == "Validator"}
    assert val["GCN_ARCH_NAME"].startswith("gfx950") and "PT_VERSION" in val and "ROCBLAS_VERSION" in val and "HIPBLASLT_VERSION" in val
    ops = [r for r in rows if r[0] != "Validator"]
    assert len(ops) >= 40 and all(r[0].endswith(("_float_TN", "_float_NN", "_float_NT", "_float_TT")) for r in ops)
