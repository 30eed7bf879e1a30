#!/usr/bin/env python3
"""Is the wave sweep bound by the LDS bank conflicts of its byte gather?  Times the fused sweep at cfg2 with
random parents (the real case: ~3.5-way conflicts per 32-lane group) and with a parent vector laid out so
that the 32 lanes of a group read 32 different banks in every one of the 16 gather instructions."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pansim_amd as pa  # noqa: E402

N, L = 1000, 1200000
rng = np.random.default_rng(0)
i = np.arange(N)
lane, k = i // 16, i % 16
free = (4 * (lane % 32) + (k & 3) + 128 * (k >> 2) + 512 * (lane // 32)) % N
cases = {"random parents": rng.integers(0, N, N), "conflict-free parents": free, "identity": i}
for lm, lh in ((60000.0, 3000.0), (0.0, 0.0)):
    for name, idx in cases.items():
        idx = idx.astype(np.uint32)
        core = pa.Population(N, L, 4, True, 0.0, 0, 2000)
        core.set_rates([lm], [lh])
        for g in range(3):
            core.step(g, idx, True)
        core.sync()
        t0 = time.perf_counter()
        for g in range(20):
            core.step(10 + g, idx, True)
        core.sync()
        dt = (time.perf_counter() - t0) / 20
        print(json.dumps({"case": name, "lam_mut": lm, "lam_hr": lh, "ms": round(dt * 1e3, 4), "GBps": round(2.0 * N * L / dt / 1e9, 1)}), flush=True)
        core.close()
