#!/usr/bin/env python3
"""BASELINE configs[3] on ONE GPU: --pop_size 65536 --core_size 1200000 --pan_genes 6000 (78.6 GB of core state)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pansim_amd as pa  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1200000
t0 = time.perf_counter()
sim = pa.Simulation(pa.make_params(pop_size=65536, core_size=L, pan_genes=6000, seed=0, n_gen=10, max_distances=100000))
print(json.dumps({"op": "create", "s": time.perf_counter() - t0}), flush=True)
sim.run(1); sim.sync()
sim.enable_timing(True)
t0 = time.perf_counter()
n = 4
sim.run(n); sim.sync()
dt = (time.perf_counter() - t0) / n
nl, ms, b = sim.sweep_timing()
print(json.dumps({"op": "generation loop", "N": 65536, "L": L, "ms_per_gen": dt * 1e3, "sweep_ms": ms / nl,
                  "sweep_GBps": b / (ms / nl) / 1e6}), flush=True)
t0 = time.perf_counter()
c, a = sim.final_distances()
print(json.dumps({"op": "final_distances", "P": len(c), "s": time.perf_counter() - t0, "core_mean": float(c.mean()),
                  "acc_mean": float(a.mean())}), flush=True)
sim.close()
