#!/usr/bin/env python3
"""All-pairs core distance phase at the cfg5 population (N = 8192, L = 1.2 M by default): kernel time of the
matrix-core form against the xor + popcount tiles: python scripts/allpairs_bench.py [N] [L] [P]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pansim_amd as pa  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1200000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
modes = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [6, 5]
rng = np.random.default_rng(0)
r1 = rng.integers(0, N, P).astype(np.uint32)
r2 = rng.integers(0, N, P).astype(np.uint32)
pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
pop.set_rates([0.05 * L], [0.0025 * L])
idx = rng.integers(0, N, N).astype(np.uint32)
for g in range(3):
    pop.step(g, idx, True)          # a diverse one-hot state
res = {}
ref = None
for mode in modes:
    pop.set_tuning("pair_mode", mode)
    (c,) = pop.pairwise_counts(r1, r2)          # builds the scratch buffers
    t0 = time.perf_counter()
    (c,) = pop.pairwise_counts(r1, r2)
    dt = time.perf_counter() - t0
    res["mode%d_ms" % mode] = round(dt * 1e3, 2)
    res["mode%d_form" % mode] = pop.last_pair_form()
    if ref is None:
        ref = c
    res["mode%d_equal_first" % mode] = bool(np.array_equal(ref, c))
res.update({"N": N, "L": L, "P": P, "pair_sites": N * (N - 1) / 2 * L})
print(json.dumps(res), flush=True)
pop.close()
