#!/usr/bin/env python3
"""D-avg (average_distance, population.rs:753-784) at wide populations: the LDS-tile popcount kernel against the
matrix-core form, whole population and one row shard of 8: python scripts/davg_bench.py [N] [G]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pansim_amd as pa  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rng = np.random.default_rng(1)
m = (rng.random((N, G)) < 0.25).astype(np.uint8)
pop = pa.Population(N, G, 2, False, 0.25, 0, 2000)
pop.load_matrix(m)
del m
out = {"N": N, "G": G}
ref = None
FORMS = (("popcount_tiles", 1, 0), ("matrix_cores_nb1", 2, 1), ("matrix_cores_nb2", 2, 2), ("two_phase_nb2", 3, 2), ("two_phase_nb1", 3, 1), ("two_phase_nb4", 3, 4),
         ("two_phase_nb2_ib32", 3, 2), ("two_phase_nb2_ib16", 3, 2))
if len(sys.argv) > 3:
    FORMS = tuple(f for f in FORMS if f[0] in sys.argv[3].split(","))
for name, form, nb in FORMS:
    pop.set_tuning("davg_form", form)
    pop.set_tuning("davg_nb", nb)
    pop.set_tuning("davg_ib", 32 if "ib32" in name else 16 if "ib16" in name else 0)
    v = pop.average_distance()
    if ref is None:
        ref = v
    t0 = time.perf_counter()
    for _ in range(3):
        v = pop.average_distance()
    out[name + "_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    out[name + "_equal"] = bool(np.array_equal(v, ref))
    if form >= 2:
        t0 = time.perf_counter()
        for _ in range(3):
            r = pop.average_distance_rows(0, N // 8)
        out[name + "_rows_1_of_8_ms"] = (time.perf_counter() - t0) / 3 * 1e3
        out[name + "_rows_equal"] = bool(np.array_equal(r, ref[:N // 8]))
print(json.dumps(out))
