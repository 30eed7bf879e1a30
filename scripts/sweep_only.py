#!/usr/bin/env python3
"""Runs only the fused core sweep (for rocprofv3 counter passes): python scripts/sweep_only.py [n] [lam_hr] [N] [L]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pansim_amd as pa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hr = float(sys.argv[2]) if len(sys.argv) > 2 else 3000.0
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 1200000
idx = np.random.default_rng(0).integers(0, N, N).astype(np.uint32)
core = pa.Population(N, L, 4, True, 0.0, 0, 2000, global_cols=1200000)
core.set_rates([60000.0], [hr])
for g in range(n):
    core.step(g, idx, True)
core.sync()
print("done", n)
