#!/usr/bin/env python3
"""Micro-benchmarks of the individual operators on one GPU (not part of the driver contract).
Prints one JSON line per measurement.  Usage: python scripts/kernel_bench.py [what ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pansim_amd as pa  # noqa: E402


def timeit(fn, sync, n=10, warm=2):
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n


def main():
    what = set(sys.argv[1:]) or {"sweep", "acc", "pairs", "loop"}
    N, L, G = 1000, 1200000, 4000
    rng = np.random.default_rng(0)
    idx = rng.integers(0, N, N).astype(np.uint32)
    if "sweep" in what:
        for hr in (3000.0, 30000.0):
            for bpc in (4, 5, 6, 7, 8):
                core = pa.Population(N, L, 4, True, 0.0, 0, 2000)
                core.set_tuning("sweep_blocks_per_cu", bpc)
                core.set_rates([60000.0], [hr])
                gen = [0]

                def f():
                    gen[0] += 1
                    core.step(gen[0], idx, True)
                dt = timeit(f, core.sync, n=10)
                print(json.dumps({"op": "core.step(gather+mut+HR)", "lam_hr": hr, "blocks_per_cu": bpc,
                                  "ms": dt * 1e3, "GBps": 2.0 * N * L / dt / 1e9}), flush=True)
                for name, fn in (("gather", lambda: core.next_generation(idx)),
                                 ("mutate", lambda: core.mutate_alleles(3)),
                                 ("recombine", lambda: core.recombine(3))):
                    if bpc == 6 and hr == 3000.0:
                        dt = timeit(fn, core.sync, n=5)
                        print(json.dumps({"op": "core." + name, "ms": dt * 1e3, "GBps": 2.0 * N * L / dt / 1e9}), flush=True)
                core.close()
    if "ablate" in what:
        # same kernel, different rates: 0 = gather only; tiny = L1 Philox + detection, ~no candidates
        for name, lm, lh in (("no events", 0.0, 0.0), ("tiny rates (bC=0)", 1e-4, 0.0), ("mut only 60000", 60000.0, 0.0),
                             ("mut 60000 + HR 3000", 60000.0, 3000.0), ("mut 6000 + HR 300", 6000.0, 300.0)):
            core = pa.Population(N, L, 4, True, 0.0, 0, 2000)
            core.set_rates([lm], [lh])
            gen = [0]

            def f():
                gen[0] += 1
                core.step(gen[0], idx, True)
            dt = timeit(f, core.sync, n=10)
            print(json.dumps({"op": "core.step ablation", "case": name, "ms": dt * 1e3, "GBps": 2.0 * N * L / dt / 1e9}), flush=True)
            core.close()
    if "acc" in what:
        acc = pa.Population(N, G, 2, False, 0.25, 0, 2000)
        for lr in ([2700.0, 299.99999999999994], [27000.0, 2999.9999999999995]):
            acc.set_rates([3600.0, 400000.0], lr, [0, 3600], [3600, 4000])
            for name, fn in (("step(gather+mut+HGT)", lambda: acc.step(1, idx, True)),
                             ("next_generation", lambda: acc.next_generation(idx)),
                             ("mutate", lambda: acc.mutate_alleles(1)),
                             ("recombine(HGT)", lambda: acc.recombine(1)),
                             ("fitness_terms", lambda: acc.fitness_terms(np.zeros(G)))):
                dt = timeit(fn, acc.sync, n=10)
                print(json.dumps({"op": "acc." + name, "lam_hgt": lr[0], "ms": dt * 1e3}), flush=True)
        acc.close()
    if "pairs" in what:
        core = pa.Population(N, L, 4, True, 0.0, 0, 2000)
        for mode in (1, 2):
            core.set_tuning("pair_mode", mode)
            for P in (100000, 1000000):
                r1, r2 = pa.sample_pairs(0, N, P)
                dt = timeit(lambda: core.pairwise_counts(r1, r2), core.sync, n=3, warm=1)
                print(json.dumps({"op": "core.pairwise_counts", "mode": mode, "P": P, "ms": dt * 1e3,
                                  "Mpairs_per_s": P / dt / 1e6}), flush=True)
        core.close()
    if "rows" in what:
        for hr in (3000.0, 30000.0):
            for rows in (2, 3, 4):
                for bpc in (5, 6, 7, 8):
                    core = pa.Population(N, L, 4, True, 0.0, 0, 2000)
                    core.set_tuning("sweep_blocks_per_cu", bpc)
                    core.set_tuning("sweep_rows", rows)
                    core.set_rates([60000.0], [hr])
                    gen = [0]

                    def f():
                        gen[0] += 1
                        core.step(gen[0], idx, True)
                    dt = timeit(f, core.sync, n=10)
                    print(json.dumps({"op": "core.step", "lam_hr": hr, "rows": rows, "blocks_per_cu": bpc,
                                      "ms": dt * 1e3, "GBps": 2.0 * N * L / dt / 1e9}), flush=True)
                    core.close()
    if "loop" in what:
        for bpc in ("5", "6", "7", "8"):
            os.environ["PANSIM_SWEEP_BLOCKS_PER_CU"] = bpc
            sim = pa.Simulation(pa.make_params(seed=0, n_gen=1000, max_distances=1000))
            sim.run(5)
            sim.sync()
            sim.enable_timing(True)
            t0 = time.perf_counter()
            sim.run(40)
            sim.sync()
            dt = (time.perf_counter() - t0) / 40
            n, ms, b = sim.sweep_timing()
            print(json.dumps({"op": "generation loop", "blocks_per_cu": int(bpc), "ms_per_gen": dt * 1e3,
                              "gen_per_s": 1 / dt, "sweep_ms": ms / n, "sweep_GBps": b / (ms / n) / 1e6}), flush=True)
            sim.close()


if __name__ == "__main__":
    main()
