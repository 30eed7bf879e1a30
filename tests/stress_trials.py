"""Randomised GPU-vs-oracle stress of the operators whose kernels have host-chosen geometry
(distances: thread table, 2-bit / nibble / all-pairs / transposed forms; HGT: atomic / binned forms,
partitions; core sweep: wave / block / inline forms and their launch parameters; whole generation loops).
Used by tests/test_gpu_stress.py (a time-boxed subset in the -m gpu suite) and by
scripts/stress_parity.py (long runs)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(trials, seed=1, log=print):
    """run `trials` randomised trials; returns the number of mismatches"""
    import pansim_amd as pa
    from oracle import oracle as o
    rng = np.random.default_rng(seed)
    bad = 0
    for t in range(trials):
        # ---- distances
        N = int(rng.choice([2, 3, 17, 64, 100, 257, 1000, 1100, 1500, 3000]))
        L = int(rng.choice([1, 7, 16, 129, 512, 777, 2049]))
        P = int(rng.choice([1, 2, 33, 500, 4000]))
        m = (1 << rng.integers(0, 4, (N, L))).astype(np.uint8)
        if rng.random() < 0.15:
            m[rng.integers(0, N), rng.integers(0, L)] = int(rng.choice([0, 3, 5, 15, 16, 200]))     # not one-hot / not a nibble
        r1 = rng.integers(0, N, P).astype(np.uint32)
        r2 = rng.integers(0, N, P).astype(np.uint32)
        if rng.random() < 0.3:
            r1[:] = r1[0]                                  # one long run of equal first individuals
        want = o.pairwise_hamming_counts(m, 0, L, r1, r2)
        pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
        pop.set_tuning("pair_mode", int(rng.integers(0, 8)))
        pop.load_matrix(m)
        (got,) = pop.pairwise_counts(r1, r2)
        if not np.array_equal(got, want):
            bad += 1
            log("DISTANCE MISMATCH", t, N, L, P)
        pop.close()
        # ---- HGT
        N = int(rng.choice([2, 5, 50, 130, 700, 2500]))
        G = int(rng.choice([1, 63, 64, 65, 300, 1000, 4000]))
        g1 = int(rng.integers(0, G + 1))
        comps = [(0, g1), (g1, G)] if 0 < g1 < G else [(0, G)]
        cb, ce = [c[0] for c in comps], [c[1] for c in comps]
        lr = [float(rng.choice([0.0, 0.5, 3.0, 40.0, 700.0])) for _ in comps]
        a = (rng.random((N, G)) < rng.choice([0.0, 0.05, 0.3, 0.9])).astype(np.uint8)
        seed, gen = int(rng.integers(0, 2**40)), int(rng.integers(0, 2**31))
        want = a.copy()
        o.recombine_acc(want, seed, gen, cb, ce, lr)
        acc = pa.Population(N, G, 2, False, 0.3, seed, 0)
        acc.set_tuning("hgt_mode", int(rng.integers(0, 3)))
        if rng.random() < 0.3:
            acc.set_tuning("hgt_list_in_global", 1)
        if rng.random() < 0.4:
            row_bytes = 8 * ((G + 63) // 64)
            acc.set_tuning("lds_limit", int(min(160 * 1024, max(1024 + int(rng.integers(2, 40)) * row_bytes, 2 * G + 8192))))
        if rng.random() < 0.3:
            acc.set_tuning("hgt_slices", int(rng.integers(1, 9)))
        if rng.random() < 0.25:
            acc.set_tuning("hgt_bin_cap", int(rng.choice([1, 5, 64])))        # full bins: the overflow image
        if rng.random() < 0.3:
            acc.set_tuning("hgt_apply_threads", int(rng.choice([256, 512])))  # smaller workgroups of the LDS-image pass
        acc.set_rates([0.0] * len(comps), lr, cb, ce)
        acc.load_matrix(a)
        acc.recombine(gen)
        if not np.array_equal(acc.read_matrix(), want):
            bad += 1
            log("HGT MISMATCH", t, N, G, comps, lr)
        acc.close()
        # ---- core sweep (fused gather + mutate + HR) under random kernel geometries
        N = int(rng.choice([2, 30, 100, 1000, 1024, 1025, 1500, 2048, 5000, 9000, 20000]))
        L = int(rng.choice([1, 3, 10, 40]))
        LG = int(rng.choice([L, 50 * L, 1200000]))
        off = int(rng.integers(0, LG - L + 1))
        # (per-site rates on every side of the plan's limits: no symbol-decided mutations (k = 0), the default plan, plans
        # that reach the symbols of n = 1 (cshift 1) and plans the queued sweeps do not take (cshift >= 2))
        lm = float(rng.choice([0.0, 0.2, 0.03 * LG, 0.05 * LG, 0.11 * LG, 0.2 * LG]))
        lh = float(rng.choice([0.0, 0.1, 0.02 * LG, 0.2 * LG])) if N > 1 else 0.0
        m = (1 << rng.integers(0, 4, (N, L))).astype(np.uint8)
        if rng.random() < 0.2:
            # a loaded matrix may hold any byte (ps_load_matrix): the sweeps must carry bytes above 15 through
            # gather, mutation and HR unchanged
            for _ in range(int(rng.integers(1, 8))):
                m[rng.integers(0, N), rng.integers(0, L)] = int(rng.choice([0, 3, 16, 17, 128, 200, 255]))
        sample = rng.integers(0, N, N).astype(np.uint32)
        if rng.random() < 0.4:
            sample = np.sort(sample)          # ascending parents: the window sweep above 1024 individuals
        seed, gen = int(rng.integers(0, 2**40)), int(rng.integers(0, 2**31))
        plan = o.core_plan(lm, lh, LG)
        want = o.next_generation(m, sample)
        o.mutate_core(want, off, seed, gen, plan)
        o.recombine_core(want, off, seed, gen, plan)
        core = pa.Population(N, L, 4, True, 0.0, seed, 0, col_offset=off, global_cols=LG)
        tune = {}
        if rng.random() < 0.4:
            tune["force_block_sweep"] = 1
        if rng.random() < 0.3:
            tune["block_waves"] = int(rng.choice([4, 8, 16]))
        if rng.random() < 0.3:
            tune["block_batch"] = 2
        if rng.random() < 0.2:
            tune["no_block_preload"] = 1
        if rng.random() < 0.2:
            tune["force_inline_sweep"] = 1
        if rng.random() < 0.5:
            tune["sweep_out_of_place"] = int(rng.integers(-1, 3))
        if rng.random() < 0.25:
            tune["sweep_queue_cap"] = int(rng.choice([1, 8, 32, 100]))      # full queues: the queue-free redo path
        if rng.random() < 0.3:
            tune["sweep_rows"] = int(rng.integers(2, 5))
            tune["sweep_blocks_per_cu"] = int(rng.integers(1, 9))
        for k, v in tune.items():
            core.set_tuning(k, v)
        core.set_rates([lm], [lh])
        core.load_matrix(m)
        try:
            core.step(gen, sample, True)
            if not np.array_equal(core.read_matrix(), want):
                bad += 1
                log("SWEEP MISMATCH", t, N, L, LG, off, lm, lh, tune)
        except pa.PansimError as e:
            if e.code != -1:       # (a geometry that does not fit is refused with PS_ERR_INVALID, never wrong)
                bad += 1
                log("SWEEP ERROR", t, N, L, lm, lh, tune, e)
        core.close()
        # ---- D-avg: LDS-tile popcount kernels against the matrix-core form (either fragment count), whole and in row shards
        if t % 2 == 0:
            N = int(rng.choice([2, 3, 31, 64, 65, 200, 1000, 2300]))
            G = int(rng.choice([1, 8, 63, 64, 65, 300, 1000, 4000, 4200]))
            cg = int(rng.choice([0, 1, 7, 2000]))
            a = (rng.random((N, G)) < rng.choice([0.0, 0.05, 0.3, 0.9])).astype(np.uint8)
            if N > 4 and rng.random() < 0.5:
                a[1] = a[0]
            want = o.average_distance(a, False, cg)
            acc = pa.Population(N, G, 2, False, 0.3, 0, cg)
            acc.load_matrix(a)
            acc.set_tuning("davg_form", int(rng.integers(0, 4)))
            acc.set_tuning("davg_nb", int(rng.choice([0, 1, 2, 4])))
            acc.set_tuning("davg_ib", int(rng.choice([0, 16, 32])))
            got = acc.average_distance()
            K = int(rng.integers(1, min(N, 5) + 1))
            rows = np.concatenate([acc.average_distance_rows(N * r // K, N * (r + 1) // K - N * r // K) for r in range(K)])
            if not (np.array_equal(got, want, equal_nan=True) and np.array_equal(rows, want, equal_nan=True)):
                bad += 1
                log("D-AVG MISMATCH", t, N, G, cg, K)
            acc.close()
        # ---- whole generation loop (every 5th trial): random parameters, 4 generations without host sync
        if t % 5 == 0:
            from orc_sim import OracleSim
            N = int(rng.choice([2, 9, 64, 200, 1030, 2100, 3300, 5000]))
            L = int(rng.choice([1, 17, 300, 1500]))
            cg = int(rng.choice([0, 5, 40]))
            pg = cg + int(rng.choice([0, 1, 64, 500]))
            if pg == 0:
                pg, cg = 10, 10
            kw = dict(pop_size=N, core_size=L, pan_genes=pg, core_genes=cg, HR_rate=float(rng.choice([0.0, 0.05, 0.6])),
                      HGT_rate=float(rng.choice([0.0, 0.05, 0.6])), avg_gene_freq=float(rng.choice([0.3, 0.5, 0.9])))
            extra = {}
            if rng.random() < 0.4:
                extra["prop_positive"] = float(rng.choice([0.0, 0.3, 1.0]))
            if rng.random() < 0.3:
                extra["competition_strength"] = float(rng.choice([0.5, 30.0]))
            if rng.random() < 0.3:
                extra["no_control_genome_size"] = True
            seed = int(rng.integers(0, 2**40))
            for k in ("PANSIM_HEAVY_HGT", "PANSIM_HGT_MODE", "PANSIM_WINDOW_SWEEP", "PANSIM_SWEEP_OOP"):
                os.environ.pop(k, None)
            if rng.random() < 0.5:
                os.environ["PANSIM_WINDOW_SWEEP"] = str(int(rng.integers(0, 2)))       # window sweep / block sweep for N > 1024
            if rng.random() < 0.3:
                os.environ["PANSIM_SWEEP_OOP"] = str(int(rng.integers(0, 3)))
            if rng.random() < 0.4:
                os.environ["PANSIM_HEAVY_HGT"] = "1"
                os.environ["PANSIM_HGT_MODE"] = str(int(rng.integers(0, 3)))
            try:
                prm = pa.make_params(seed=seed, n_gen=4, max_distances=50, **kw, **extra)
                if pa.validate(prm)[0] and N >= 2:
                    sim = pa.Simulation(prm)
                    ref = OracleSim(seed=seed, **kw, **extra)
                    sim.run(4)
                    sim.sync()
                    for g in range(4):
                        ref.generation(g)
                    if not (np.array_equal(sim.last_parents(), ref.last_idx) and np.array_equal(sim.core_genome.read_matrix(), ref.core)
                            and np.array_equal(sim.pan_genome.read_matrix(), ref.acc)):
                        bad += 1
                        log("LOOP MISMATCH", t, kw, extra, os.environ.get("PANSIM_HEAVY_HGT"), os.environ.get("PANSIM_HGT_MODE"))
                    sim.close()
            except (pa.PansimError, AssertionError) as e:
                log("LOOP skipped", t, kw, extra, str(e)[:80])
            for k in ("PANSIM_HEAVY_HGT", "PANSIM_HGT_MODE", "PANSIM_WINDOW_SWEEP", "PANSIM_SWEEP_OOP"):
                os.environ.pop(k, None)
    return bad
