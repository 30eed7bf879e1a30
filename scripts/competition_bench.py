#!/usr/bin/env python3
"""Generation loop with --competition_strength > 0 (D-avg every generation)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pansim_amd as pa  # noqa: E402
for comp in (0.0, 100.0):
    sim = pa.Simulation(pa.make_params(seed=0, n_gen=200, max_distances=1000, competition_strength=comp))
    sim.run(5); sim.sync()
    t0 = time.perf_counter(); sim.run(40); sim.sync()
    print(json.dumps({"op": "generation loop", "competition_strength": comp, "ms_per_gen": (time.perf_counter() - t0) / 40 * 1e3}), flush=True)
    sim.close()
