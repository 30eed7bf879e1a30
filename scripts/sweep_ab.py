#!/usr/bin/env python3
"""Same-process A/B of sweep variants inside the generation loop: alternates the values of one tuning key of the
core handle over batches of generations and reports the in-loop HIP-event time per sweep launch for each value.
  python scripts/sweep_ab.py KEY v0,v1,...  [rounds] [batch] [pop_size] [core_size] [HR_rate] [HGT_rate]"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pansim_amd as pa  # noqa: E402

key = sys.argv[1]
vals = [int(x) for x in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 60
N = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
L = int(sys.argv[6]) if len(sys.argv) > 6 else 1200000
hr = float(sys.argv[7]) if len(sys.argv) > 7 else 0.05
hgt = float(sys.argv[8]) if len(sys.argv) > 8 else 0.05
sim = pa.Simulation(pa.make_params(seed=0, n_gen=10, max_distances=1000, pop_size=N, core_size=L, pan_genes=6000,
                                   HR_rate=hr, HGT_rate=hgt))
sim.enable_timing(True)
for _ in range(6):                  # clock ramp / first touch
    sim.run(batch)
    sim.sync()
sim.sweep_timing(reset=True)
res = {v: [] for v in vals}
import time
wall = {v: [] for v in vals}
for r in range(rounds):
    for v in (vals if r % 2 == 0 else vals[::-1]):
        (sim.pan_genome if key.startswith("hgt_") else sim.core_genome).set_tuning(key, v)
        sim.run(8)
        sim.sync()
        sim.sweep_timing(reset=True)
        t0 = time.perf_counter()
        sim.run(batch)
        sim.sync()
        wall[v].append((time.perf_counter() - t0) / batch * 1e3)
        n, ms, _b = sim.sweep_timing(reset=True)
        res[v].append(ms / n)
out = {"key": key, "N": N, "L": L, "rounds": rounds, "batch": batch}
for v in vals:
    out["%s=%d" % (key, v)] = {"sweep_ms_median": round(statistics.median(res[v]), 4), "sweep_ms_min": round(min(res[v]), 4),
                              "sweep_ms_max": round(max(res[v]), 4), "period_ms_median": round(statistics.median(wall[v]), 4)}
print(json.dumps(out), flush=True)
sim.close()
