#!/usr/bin/env python3
"""Freeze the build's own keyed specification as regression vectors.

These are NOT reference outputs (the reference's stochastic operators are not reproducible, see
DESIGN.md 3): they are outputs of oracle/pansim_oracle.c on tiny inputs, so that a later change of a
stream, a counter layout or a threshold rule cannot slip in unnoticed.  Re-run only when the spec is
changed on purpose:  python tests/golden/make_spec_regression.py
"""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

from oracle import oracle as o  # noqa: E402
from orc_sim import OracleSim  # noqa: E402


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8))


out = {"poisson_tables": [], "sims": []}
for lam in (0.003, 3.0, 299.99999999999994, 2700.0):
    kmin, thr = o.poisson_table(lam)
    out["poisson_tables"].append({"lambda": lam, "kmin": kmin, "len": len(thr), "first": int(thr[0]),
                                  "middle": int(thr[len(thr) // 2]), "last": int(thr[-1]), "crc": crc(thr)})
for kw, extra, gens in ((dict(pop_size=20, core_size=50, pan_genes=60, core_genes=30), dict(), 3),
                        (dict(pop_size=33, core_size=129, pan_genes=200, core_genes=70, HR_rate=0.4, HGT_rate=0.3),
                         dict(prop_positive=0.25, competition_strength=5.0), 4)):
    sim = OracleSim(seed=12345, **kw, **extra)
    per_gen = []
    for g in range(gens):
        sim.generation(g)
        per_gen.append({"parents_crc": crc(sim.last_idx), "core_crc": crc(sim.core), "acc_crc": crc(sim.acc)})
    out["sims"].append({"params": kw, "extra": extra, "seed": 12345, "generations": per_gen,
                        "first_parents": [int(x) for x in sim.last_idx[:8]],
                        "core_row0": [int(x) for x in sim.core[0, :16]], "acc_row0": [int(x) for x in sim.acc[0, :16]]})
json.dump(out, open(os.path.join(HERE, "spec_regression.json"), "w"), indent=1)
print("wrote", os.path.join(HERE, "spec_regression.json"))
