"""Tuned solutions for the token GEMMs of the Stable Diffusion workload (configs[3]).

The spatial transformers' projections / feed-forward layers are `torch.nn.Linear` exactly as in the reference
(stable-diffusion/ldm/modules/sige_attention.py:86-185, attention.py: `to_q / to_k / to_v / to_out`, GEGLU `proj`, `net[2]`): on
ROCm they run as hipBLASLt / rocBLAS fp32 GEMMs at M = 8192 rows (full pass), 2016 / 160 / 48 (active tokens of a 15 % edit) and
154 (text context) -- shapes the libraries' default heuristics serve at 10 - 33 % of the fp32 MFMA peak.  PyTorch's TunableOp can
time every fp32 solution of both libraries per shape and remember the fastest; `tunableop_sd_gfx950.csv` is that table for the
shapes of the benchmarked SD forward, measured on an MI355X with this image's libraries (tools/sessions/r6_call47.sh: SD forward
10.07 -> 9.65 ms, parity against the CPU path unchanged at 7e-6 -- every candidate is an fp32 GEMM).  The table's validator lines
(torch / HIP / hipBLASLt / rocBLAS versions, gfx950) must match the running stack or PyTorch rejects it; shapes it does not hold
(another mask's token counts) run the default solution -- `tune=True` times them on first sight (tens of ms per new shape).

This changes a process-wide PyTorch setting, so nothing in the package calls it: the caller opts in (bench.py --workload sd does).
"""
import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_sd_gfx950.csv")


def enable_tuned_gemms(table: str = TABLE, tune: bool = False, results_file: str = "") -> bool:
    """Turn TunableOp on with `table` loaded.  Returns False (and leaves TunableOp off) when there is no GPU or PyTorch rejects the
    table (validators).  `tune`: also time shapes the table does not hold; they are written to `results_file` at exit if given
    (timing runs the candidates: do one eager forward under every new mask BEFORE capturing a hipGraph of it)."""
    if not torch.cuda.is_available():
        return False
    from torch.cuda import tunable

    tunable.enable(True)
    tunable.tuning_enable(bool(tune))
    if results_file:
        tunable.set_filename(results_file, insert_device_ordinal=False)
    ok = bool(os.path.isfile(table) and tunable.read_file(table))
    if not ok and not tune:
        tunable.enable(False)
        return False
    return ok or bool(tune)


def disable_tuned_gemms() -> None:
    if torch.cuda.is_available():
        from torch.cuda import tunable

        tunable.tuning_enable(False)
        tunable.enable(False)
