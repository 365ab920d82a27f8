"""Legs of bench.py (repo root): workloads, CPU baseline, per-layer roofline tables, oracle checks, side legs, the compact line.  bench.py keeps the timed path."""
