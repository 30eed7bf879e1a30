"""The CPU oracle against the committed known-answer vectors (tests/golden/kat.json).

The reference ships no tests; these vectors are hand-derived from its source text
 plus the published Random123 Philox vectors.  They pin the oracle that the
GPU parity tests trust.
"""
import json
import math
import os

import numpy as np
import pytest

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def test_philox_random123_vectors(orc):
    for v in KAT["philox4x32_10"]:
        out = orc.philox([int(x, 16) for x in v["ctr"]], [int(x, 16) for x in v["key"]])
        assert [format(int(x), "08x") for x in out] == v["out"]


def test_hamming_vectors(orc):
    for v in KAT["hamming"]:
        x, y = np.array(v["x"], np.uint8), np.array(v["y"], np.uint8)
        assert orc.hamming(x, y) == v["xor_popcount"]
        d = orc.pairwise_distances(np.stack([x, y]), True, 0, [0], [1])[0]
        assert d == v["core_distance"]


def test_hamming_chunk_and_tail(orc):
    # distances.rs:27-49: 8-byte chunks plus a byte tail, any length
    rng = np.random.default_rng(0)
    for n in range(0, 40):
        x = rng.integers(0, 256, n).astype(np.uint8)
        y = rng.integers(0, 256, n).astype(np.uint8)
        want = sum(bin(int(a) ^ int(b)).count("1") for a, b in zip(x, y))
        assert orc.hamming(x, y) == want
        inter = sum(bin(int(a) & int(b)).count("1") for a, b in zip(x, y))
        uni = sum(bin(int(a) | int(b)).count("1") for a, b in zip(x, y))
        assert orc.jaccard(x, y) == (inter, uni)


def test_jaccard_vectors(orc):
    for v in KAT["jaccard"]:
        a, b = np.array(v["a"], np.uint8), np.array(v["b"], np.uint8)
        assert orc.jaccard(a, b) == (v["intersection"], v["union"])
        for cg, want in v["distance"].items():
            assert orc.pairwise_distances(np.stack([a, b]), False, int(cg), [0], [1])[0] == want


def test_average_distance_rules(orc):
    a = np.array(KAT["jaccard"][0]["a"], np.uint8)
    b = np.array(KAT["jaccard"][0]["b"], np.uint8)
    # identical rows -> 0 -> f64::MIN_POSITIVE (population.rs:774-776)
    out = orc.average_distance(np.stack([a, a, a]), False, 5)
    assert (out == 2.2250738585072014e-308).all()
    # mean over j != i in ascending j
    m = np.stack([a, b, a])
    out = orc.average_distance(m, False, 2)
    d_ab = 0.4285714285714286
    assert out[0] == (d_ab + 0.0) / 2 and out[1] == (d_ab + d_ab) / 2
    # empty union with no core genes -> NaN (0/0)
    z = np.zeros((2, 4), np.uint8)
    assert math.isnan(orc.pairwise_distances(z, False, 0, [0], [1])[0])


def test_int_to_base(orc):
    for k, v in KAT["int_to_base"].items():
        assert orc.lib().orc_int_to_base(int(k)).decode() == v


def test_next_generation_vector(orc):
    v = KAT["next_generation"]
    assert orc.next_generation(np.array(v["pop"], np.uint8), v["sample"]).tolist() == v["next"]


def test_gene_frequencies_vector(orc):
    v = KAT["gene_frequencies"]
    assert orc.gene_frequencies(np.array(v["pop"], np.uint8), v["core_genes"]).tolist() == v["freqs"]


def test_standard_deviation_vector(orc):
    import ctypes as C
    v = KAT["standard_deviation"]
    s, m = C.c_double(), C.c_double()
    orc.lib().orc_standard_deviation(np.array(v["values"]), len(v["values"]), C.byref(s), C.byref(m))
    assert (s.value, m.value) == (v["std"], v["mean"])


def test_rust_display(orc):
    for x, want in KAT["rust_display_f64"]:
        assert orc.fmt_f64(x) == want
    assert orc.fmt_f64(float("nan")) == "NaN"
    assert orc.fmt_f64(float("inf")) == "inf" and orc.fmt_f64(float("-inf")) == "-inf"
    s = orc.fmt_f64(2.2250738585072014e-308)
    assert s.startswith("0.000") and s.endswith("22250738585072014") and float(s) == 2.2250738585072014e-308
    rng = np.random.default_rng(1)
    for x in np.concatenate([rng.random(200), rng.random(50) * 1e-9, rng.random(50) * 1e12]):
        s = orc.fmt_f64(x)
        assert "e" not in s and float(s) == x and len(s) <= len(repr(float(x))) + 25


def test_derived_parameters(orc):
    for v in KAT["derived"]:
        d = orc.derive(orc.make_params(**v["params"]))
        assert d.pan_size == v["pan_size"]
        assert d.avg_gene_freq_adj == v["avg_gene_freq_adj"]
        assert d.avg_gene_num == v["avg_gene_num"]
        assert d.n_core_mutations == v["n_core_mutations"]
        assert d.n_recombinations_core == v["n_recombinations_core"]
        assert d.n_comp == v["n_comp"]
        assert list(d.comp_begin)[:d.n_comp] == v["comp_begin"]
        assert list(d.comp_end)[:d.n_comp] == v["comp_end"]
        assert list(d.n_pan_mutations)[:d.n_comp] == v["n_pan_mutations"]
        assert list(d.n_recombinations_pan)[:d.n_comp] == v["n_recombinations_pan"]
    # a compartment with zero genes is not pushed (main.rs:341, :355)
    d = orc.derive(orc.make_params(prop_genes2=0.0))
    assert d.n_comp == 1 and d.comp_end[0] == 4000
    d = orc.derive(orc.make_params(prop_genes2=1.0))
    assert d.n_comp == 1 and d.comp_begin[0] == 0 and d.n_pan_mutations[0] == 4000000.0
    # avg_gene_freq below the core proportion clamps to 0 (main.rs:266-268)
    d = orc.derive(orc.make_params(avg_gene_freq=0.2))
    assert d.avg_gene_freq_adj == 0.0 and d.avg_gene_num == 0


def _plan_masses(plan):
    """total mass of (mutate to one allele only, mutate to one allele AND receive, receive only) under a plan"""
    T = [t / 2.0**32 for t in plan.T]
    res = plan.R / 64.0
    return plan.k / 64.0 + res * T[0], res * (T[3] - T[2]), res * (T[6] - T[5])


def test_plan_probabilities(orc):
    pr = KAT["probabilities"]
    v = pr["core_hit_p"]
    plan = orc.core_plan(v["lam"], 0.0, v["L"])
    assert plan.T[2] == plan.T[6] and plan.has_events == 1
    a, b, c = _plan_masses(plan)
    assert abs(3 * a - v["p"]) < 2.0**-31 and b == 0 and c == 0
    for f in pr["acc_flip"]:
        assert abs(orc.lib().orc_acc_flip_threshold(f["lam"], f["n"]) / 2.0**32 - f["p"]) < 2.0**-31
    # joint plans: mass of "receives a donor allele" is q, independent of mutation; the symbol-decided part of an allele's
    # mass is k / 64, the residual symbols carry the rest of every class
    for lam_mut, lam_hr, L, krc in ((60000.0, 30000.0, 1200000, (1, 2, 1)), (60000.0, 3000.0, 1200000, (1, 1, 0)),
                                    (60000.0, 0.0, 1200000, (1, 1, 0)), (12000.0, 0.0, 1200000, (0, 1, 0)),
                                    (600000.0, 0.0, 1200000, (8, 2, 3)), (60000.0, 60000.0, 1200000, (0, 7, 1))):
        plan = orc.core_plan(lam_mut, lam_hr, L)
        p, q = -math.expm1(-lam_mut / L), -math.expm1(-lam_hr / L)
        a, b, c = _plan_masses(plan)
        assert abs(a - p * (1 - q) / 3) < 1e-9 and abs(b - p * q / 3) < 1e-9 and abs(c - (1 - p) * q) < 1e-9
        assert (plan.k, plan.R, plan.cshift) == krc and 3 * plan.k + plan.R <= (4 << plan.cshift)
        assert plan.k == math.floor(64 * p * (1 - q) / 3)
        T = list(plan.T)
        assert T == sorted(T) and T[1] - T[0] in (T[0], T[0] + 1, T[0] - 1)
    assert orc.core_plan(0.0, 0.0, 100).has_events == 0


def test_sample_weights_rules(orc):
    N, G = 6, 4
    ng = np.array([1000, 1001, 999, 1000, 990, 1010], np.int32)
    lw = np.zeros(N)
    ones = np.ones(N)
    # neutral defaults: weights proportional to 0.99^(n_genes - avg_gene_num)
    rc, w = orc.sample_weights(ng, lw, G, 1000, ones, False, 0.99, 0.0)
    assert rc == 0
    want = 0.99 ** (ng - 1000.0)
    assert np.allclose(w / w.sum(), want / want.sum(), rtol=1e-12, atol=0)
    # no_control_genome_size: selection weights only -> uniform
    rc, w = orc.sample_weights(ng, lw, G, 1000, ones, True, 0.99, 0.0)
    assert rc == 0 and np.allclose(w / w.sum(), 1.0 / N)
    # G == 0: selection weights stay 1.0 (population.rs:293-296)
    rc, w0 = orc.sample_weights(ng, lw, 0, 1000, ones, True, 0.99, 0.0)
    assert rc == 0 and np.allclose(w0 / w0.sum(), 1.0 / N)
    # ln(penalty) = NaN -> WeightedIndex::new would panic (population.rs:440)
    rc, _ = orc.sample_weights(ng, lw, G, 1000, ones, False, -1.0, 0.0)
    assert rc != 0
    # competition: weight proportional to avg_dist ** strength
    avg = np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    rc, w = orc.sample_weights(np.full(N, 1000, np.int32), lw, G, 1000, avg, False, 0.99, 2.0)
    assert rc == 0 and np.allclose(w / w.sum(), avg**2 / (avg**2).sum(), rtol=1e-12)


def test_neg_inf_reset_rule(orc):
    # population.rs:312-318: a present gene with s = -1 makes the row's log-fitness 0.0
    m = np.array([[1, 1, 0], [0, 1, 0], [1, 0, 1]], np.uint8)
    sel = np.array([-1.0, 0.5, 0.25])
    ng, lw = orc.fitness_terms(m, sel)
    assert ng.tolist() == [2, 1, 2]
    assert lw[0] == 0.0 and lw[2] == 0.0
    assert lw[1] == math.log(1.0 + 0.0) + math.log(1.5) + math.log(1.0)


def test_draw_parents_follows_weights(orc):
    w = np.array([0.0, 1.0, 3.0, 0.0])
    rc, idx = orc.draw_parents(np.tile(w, 2500), 7, 0)
    assert rc == 0
    counts = np.bincount(idx % 4, minlength=4)
    assert counts[0] == 0 and counts[3] == 0
    assert abs(counts[2] / counts[1] - 3.0) < 0.3
    rc, idx2 = orc.draw_parents(np.tile(w, 2500), 7, 0)
    assert np.array_equal(idx, idx2)
    rc, idx3 = orc.draw_parents(np.tile(w, 2500), 7, 1)
    assert not np.array_equal(idx, idx3)


def test_pairs_never_self(orc):
    r1, r2 = orc.sample_pairs(0, 50, 20000)
    assert (r1 != r2).all() and r1.max() < 50 and r2.max() < 50
    assert np.bincount(r1, minlength=50).min() > 250 and np.bincount(r2, minlength=50).min() > 250
    r1, r2 = orc.sample_pairs(0, 2, 100)
    assert (r1 + r2 == 1).all()


def test_selection_coefficients(orc):
    s = orc.selection_coefficients(1, 5000, -0.1, 10.0, 10.0)
    assert (s == 0).all()                                   # neutral (main.rs:287-292)
    s = orc.selection_coefficients(1, 20000, 0.25, 10.0, 5.0)
    pos = s[s > 0]
    neg = s[s < 0]
    assert abs(len(pos) / 20000 - 0.25) < 0.02
    assert abs(pos.mean() - 0.1) < 0.01                     # Exp(10) mean
    assert (neg >= -1.0).all()                              # redrawn while > 1.0 (main.rs:309-311)
    assert abs(-neg.mean() - (0.2 - 1.0 * math.exp(-5) / (1 - math.exp(-5)))) < 0.02


def test_spec_regression_vectors(orc):
    # the build's own keyed specification, frozen by tests/golden/make_spec_regression.py: a change
    # of a stream, counter layout or threshold rule must be deliberate (regenerate the fixture)
    import json, os, sys, zlib
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from orc_sim import OracleSim
    fx = json.load(open(os.path.join(here, "golden", "spec_regression.json")))
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).view(np.uint8))
    for t in fx["poisson_tables"]:
        kmin, thr = orc.poisson_table(t["lambda"])
        assert (kmin, len(thr), int(thr[0]), int(thr[len(thr) // 2]), int(thr[-1]), crc(thr)) == \
               (t["kmin"], t["len"], t["first"], t["middle"], t["last"], t["crc"])
    for s in fx["sims"]:
        sim = OracleSim(seed=s["seed"], **s["params"], **s["extra"])
        for g, want in enumerate(s["generations"]):
            sim.generation(g)
            assert (crc(sim.last_idx), crc(sim.core), crc(sim.acc)) == (want["parents_crc"], want["core_crc"], want["acc_crc"])
        assert [int(x) for x in sim.last_idx[:8]] == s["first_parents"]
        assert [int(x) for x in sim.core[0, :16]] == s["core_row0"]
        assert [int(x) for x in sim.acc[0, :16]] == s["acc_row0"]


def test_oracle_loop_speaks_draw_order_at_its_boundary():
    # tests/orc_sim.py keeps its rows in ascending parent order inside (like the library's engine) and presents them in the
    # reference's order: with every rate at zero a generation is a pure gather, so row k must be row last_idx[k] of the
    # generation before (population.rs:443, main.rs:445-447) -- the same property the GPU suite checks on the library
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from orc_sim import OracleSim
    sim = OracleSim(seed=3, pop_size=60, core_size=40, pan_genes=50, core_genes=10, core_mu=0.0, HR_rate=0.0, HGT_rate=0.0,
                    rate_genes1=0.0, rate_genes2=0.0)
    rng = np.random.default_rng(1)
    sim.core = (1 << rng.integers(0, 4, (60, 40))).astype(np.uint8)
    sim.acc = (rng.random((60, 40)) < 0.5).astype(np.uint8)
    core, acc = sim.core.copy(), sim.acc.copy()
    unsorted_seen = False
    for g in range(4):
        sim.generation(g)
        par = sim.last_idx
        unsorted_seen |= bool((np.diff(par.astype(np.int64)) < 0).any())
        assert (np.diff(sim.internal_idx.astype(np.int64)) >= 0).all()
        assert np.array_equal(sim.core, core[par]) and np.array_equal(sim.acc, acc[par])
        core, acc = sim.core.copy(), sim.acc.copy()
    assert unsorted_seen
