"""Modes of bench.py (repo root): common, cpu, single, dist, native, launch."""
