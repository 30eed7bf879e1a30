#!/usr/bin/env python3
"""Phase breakdown of the block sweep (pitch > 1024): gather only, mutate+HR only, fused."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pansim_amd as pa  # noqa: E402

cases = [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])] or [(2048, 600000), (8192, 150000), (65536, 18000)]
for N, L in cases:
    rng = np.random.default_rng(0)
    idx = rng.integers(0, N, N).astype(np.uint32)
    core = pa.Population(N, L, 4, True, 0.0, 0, 2000, global_cols=1200000)
    core.set_rates([60000.0], [3000.0])

    def timed(fn, n=5):
        for g in range(2):
            fn(g)
        core.sync()
        t0 = time.perf_counter()
        for g in range(n):
            fn(10 + g)
        core.sync()
        return (time.perf_counter() - t0) / n

    res = {"N": N, "L": L}
    for name, fn in (("fused", lambda g: core.step(g, idx, True)),
                     ("gather", lambda g: core.next_generation(idx)),
                     ("mutate", lambda g: core.mutate_alleles(g)),
                     ("recombine", lambda g: core.recombine(g))):
        dt = timed(fn)
        res[name + "_ms"] = round(dt * 1e3, 3)
        res[name + "_GBps"] = round(2.0 * N * L / dt / 1e9, 1)
    print(json.dumps(res), flush=True)
    core.close()
