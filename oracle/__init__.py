"""Test oracle for the SIGE tiling-sparse-conv hot path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (sige_amd/) never imports it.
"""
