#!/usr/bin/env python3
"""cfg3 generation loop under (lds_limit, hgt_slices) of the binned HGT: period and sweep ms per combination (same process,
alternating): python scripts/cfg3_hgt_tune.py"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pansim_amd as pa  # noqa: E402

sim = pa.Simulation(pa.make_params(seed=0, n_gen=10, max_distances=1000, pop_size=1000, core_size=1200000, pan_genes=6000,
                                   HR_rate=0.5, HGT_rate=0.5))
sim.enable_timing(True)
for _ in range(6):
    sim.run(60)
    sim.sync()
combos = [(160 * 1024, 0), (160 * 1024, 16), (160 * 1024, 32), (68 * 1024, 0), (68 * 1024, 16), (36 * 1024, 16), (36 * 1024, 32), (20 * 1024, 16)]
res = {c: [] for c in combos}
for r in range(5):
    for c in (combos if r % 2 == 0 else combos[::-1]):
        sim.pan_genome.set_tuning("lds_limit", c[0])
        sim.pan_genome.set_tuning("hgt_slices", c[1])
        sim.run(8)
        sim.sync()
        sim.sweep_timing(reset=True)
        t0 = time.perf_counter()
        sim.run(60)
        sim.sync()
        dt = (time.perf_counter() - t0) / 60 * 1e3
        n, ms, _b = sim.sweep_timing(reset=True)
        res[c].append((dt, ms / n))
for c in combos:
    print(json.dumps({"lds_limit": c[0], "hgt_slices": c[1], "period_ms": round(statistics.median(x[0] for x in res[c]), 4),
                      "sweep_ms": round(statistics.median(x[1] for x in res[c]), 4)}), flush=True)
sim.close()
