#!/usr/bin/env python3
"""Accessory chain at the cfg4 population (N = 65536, G = 4000, default rates): per-operator times."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pansim_amd as pa  # noqa: E402

N, G = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 4000
idx = np.random.default_rng(0).integers(0, N, N).astype(np.uint32)
for mode in (0, 1, 2):
    acc = pa.Population(N, G, 2, False, 0.25, 0, 2000)
    acc.set_tuning("hgt_mode", mode)
    acc.set_rates([3600.0, 400000.0], [2700.0, 299.99999999999994], [0, 3600], [3600, 4000])
    res = {"N": N, "hgt_mode": mode}
    for name, fn in (("step(gather+mut)", lambda g: acc.step(g, idx, False)), ("recombine(HGT)", lambda g: acc.recombine(g)),
                     ("fitness_terms", lambda g: acc.fitness_terms(np.zeros(G))),
                     ("average_distance", lambda g: acc.average_distance())):
        for g in range(2):
            fn(g)
        acc.sync()
        t0 = time.perf_counter()
        n = 5
        for g in range(n):
            fn(10 + g)
        acc.sync()
        res[name + "_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
    print(json.dumps(res), flush=True)
    acc.close()
