"""Pieces of bench.py (the driver's one command stays `python bench.py ...`; this package only keeps the script readable)."""
