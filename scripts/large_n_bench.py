#!/usr/bin/env python3
"""Large-population path (block-per-row sweep): N in {8192, 65536} at N*L ~ 1.2e9 cells."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pansim_amd as pa  # noqa: E402

for N, L in ((2048, 600000), (8192, 150000), (65536, 18000)):
    rng = np.random.default_rng(0)
    idx = rng.integers(0, N, N).astype(np.uint32)
    core = pa.Population(N, L, 4, True, 0.0, 0, 2000, global_cols=1200000)
    core.set_rates([60000.0], [3000.0])
    for g in range(2):
        core.step(g, idx, True)
    core.sync()
    t0 = time.perf_counter()
    n = 5
    for g in range(n):
        core.step(10 + g, idx, True)
    core.sync()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"op": "core.step block sweep", "N": N, "L": L, "ms": dt * 1e3, "GBps": 2.0 * N * L / dt / 1e9}), flush=True)
    core.close()
# whole loop at N=8192 (cfg5 population) with a short genome
sim = pa.Simulation(pa.make_params(pop_size=8192, core_size=150000, seed=0, n_gen=40, max_distances=100000))
sim.run(10)            # (the first generations allocate the HGT scratch)
sim.sync()
t0 = time.perf_counter()
sim.run(30)
sim.sync()
dt = (time.perf_counter() - t0) / 30
print(json.dumps({"op": "generation loop", "N": 8192, "L": 150000, "ms_per_gen": dt * 1e3}), flush=True)
t0 = time.perf_counter()
c, a = sim.final_distances()
print(json.dumps({"op": "final_distances", "N": 8192, "P": 100000, "ms": (time.perf_counter() - t0) * 1e3}), flush=True)
sim.close()
