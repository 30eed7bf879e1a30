#!/usr/bin/env python3
"""Accessory chain only (for rocprofv3): python scripts/acc_only.py [n] [hgt_mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pansim_amd as pa  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, G = 1000, 4000
idx = np.random.default_rng(0).integers(0, N, N).astype(np.uint32)
acc = pa.Population(N, G, 2, False, 0.25, 0, 2000)
acc.set_tuning("hgt_mode", mode)
acc.set_rates([3600.0, 400000.0], [27000.0, 2999.9999999999995], [0, 3600], [3600, 4000])
for g in range(n):
    acc.step(g, idx, True)
acc.sync()
print("done")
