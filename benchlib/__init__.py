"""Pieces of bench.py: the contract line (line.py) and, per section, the measurements it prints."""
