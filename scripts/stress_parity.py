#!/usr/bin/env python3
"""Randomised GPU-vs-oracle stress (tests/stress_trials.py) as a long run.
Usage: python scripts/stress_parity.py [trials] [seed]   (on an MI355X; exits non-zero on a mismatch)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from stress_trials import run  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = run(trials, seed, lambda *a, **k: print(*a, flush=True))
print("trials", trials, "mismatches", bad)
sys.exit(1 if bad else 0)
