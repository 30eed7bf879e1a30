#!/usr/bin/env python3
"""HGT (accessory recombination) alone at cfg3 rates, per kernel variant: python scripts/hgt_bench.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pansim_amd as pa  # noqa: E402

N, G = 1000, 4000
idx = np.random.default_rng(0).integers(0, N, N).astype(np.uint32)
for rates in ([27000.0, 2999.9999999999995], [2700.0, 299.99999999999994]):
    for mode in (1, 2):
        acc = pa.Population(N, G, 2, False, 0.25, 0, 2000)
        acc.set_tuning("hgt_mode", mode)
        acc.set_rates([3600.0, 400000.0], rates, [0, 3600], [3600, 4000])
        for g in range(3):
            acc.step(g, idx, True)       # reach the steady-state density
        acc.sync()
        t0 = time.perf_counter()
        n = 10
        for g in range(n):
            acc.recombine(100 + g)
        acc.sync()
        dt = (time.perf_counter() - t0) / n
        dens = float(acc.read_matrix().mean())
        print(json.dumps({"hgt_mode": mode, "lam_hgt": rates[0], "recombine_ms": round(dt * 1e3, 3), "density": round(dens, 3)}), flush=True)
        acc.close()
