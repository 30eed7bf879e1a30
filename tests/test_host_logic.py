"""CPU-side checks of the product's host code (no GPU compute): parameter derivation,
validation text, seeded host draws, weight arithmetic and formatting of libpansim_hip.so
against the oracle and the golden vectors; the C ABI exports every declared symbol.
"""
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))


def test_library_exports_every_declared_symbol(pa):
    hdr = open(os.path.join(ROOT, "include", "pansim_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    import ctypes
    lib = ctypes.CDLL(pa.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, "declared in include/pansim_hip.h but not exported: %s" % missing
    from pansim_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.ps_abi_version() == 3      # round 4: ps_rccl_*, ps_sim_emulated_link_time, ps_last_sweep_form


def test_no_cpu_fallback_without_device(pa):
    # in the CPU container every compute entry point must fail loudly, never fall back
    if pa.load().ps_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pa.PansimError) as e:
        pa.Population(10, 20, 4, True, 0.0, 0, 0)
    assert e.value.code == -2
    with pytest.raises(pa.PansimError):
        pa.hamming_bitwise_fast([1, 2], [2, 2])
    with pytest.raises(pa.PansimError):
        pa.Simulation(pa.make_params(pop_size=10, core_size=100, pan_genes=60, core_genes=20))
    # the native RCCL exchange provider: librccl.so loads (dlopen) or not, but nothing runs without a device
    lib = pa.load()
    assert lib.ps_rccl_available() in (0, 1)
    ident = np.zeros(128, np.uint8)
    assert lib.ps_rccl_unique_id(ident) in (0, -2)      # (drawing an id needs no device in some RCCL builds)
    import ctypes
    h = ctypes.c_void_p()
    assert lib.ps_rccl_exchange_create(ident, 0, 1, 0, ctypes.byref(h)) == -2 and not h.value
    assert lib.ps_rccl_exchange_create(ident, 3, 2, 0, ctypes.byref(h)) == -1      # rank outside the world
    assert lib.ps_exchange_rccl(None, None, 0, None) == -1


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure: nothing under pansim_amd/ may reference it
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "pansim_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"\boracle\b|orc_", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_derive_matches_golden_and_oracle(pa, orc):
    for v in KAT["derived"]:
        d = pa.derive(pa.make_params(**v["params"]))
        o = orc.derive(orc.make_params(**v["params"]))
        for name in ("pan_size", "avg_gene_freq_adj", "avg_gene_num", "n_core_mutations",
                     "n_recombinations_core", "n_recombinations_pan_total", "n_comp"):
            assert getattr(d, name) == getattr(o, name)
        assert d.avg_gene_num == v["avg_gene_num"] and d.pan_size == v["pan_size"]
        for name in ("comp_begin", "comp_end", "n_pan_mutations", "n_recombinations_pan"):
            assert list(getattr(d, name)) == list(getattr(o, name))
        assert list(d.n_recombinations_pan)[:d.n_comp] == v["n_recombinations_pan"]
    rng = np.random.default_rng(0)
    for _ in range(50):
        kw = dict(pop_size=int(rng.integers(2, 5000)), core_size=int(rng.integers(1, 10**7)),
                  pan_genes=int(rng.integers(100, 10000)), avg_gene_freq=float(rng.random() * 0.99 + 0.01),
                  HR_rate=float(rng.random()), HGT_rate=float(rng.random()), core_mu=float(rng.random()),
                  rate_genes1=float(rng.random() * 3), rate_genes2=float(rng.random() * 2000),
                  prop_genes2=float(rng.random()))
        kw["core_genes"] = int(rng.integers(0, kw["pan_genes"]))
        d, o = pa.derive(pa.make_params(**kw)), orc.derive(orc.make_params(**kw))
        assert (d.pan_size, d.avg_gene_freq_adj, d.avg_gene_num, d.n_core_mutations, d.n_comp) == \
               (o.pan_size, o.avg_gene_freq_adj, o.avg_gene_num, o.n_core_mutations, o.n_comp)
        assert list(d.n_recombinations_pan) == list(o.n_recombinations_pan)
        assert list(d.n_pan_mutations) == list(o.n_pan_mutations)


VALIDATION = [
    (dict(pan_genes=600), "core_genes must be less than or equal to pan_size\n"),               # main.rs:195-198
    (dict(HR_rate=-0.5), "HR_rate and HGT_rate must be above 0.0\nHR_rate: -0.5\nHGT_rate: 0.05\n"),
    (dict(pos_lambda=0.0), "pos_lambda and neg_lambda must be above 0.0\npos_lambda: 0\nneg_lambda: 10\n"),
    (dict(rate_genes2=-1.0), "rate_genes1 and rate_genes2 must be >= 0\nrate_genes1: 1\nrate_genes2: -1\n"),
    (dict(prop_genes2=1.5), "prop_genes2 must be 0.0 <= prop_genes2 <= 1.0\nprop_genes2: 1.5\n"),
    (dict(n_gen=0), "pop_size, core_size, pan_genes, n_gen and max_distances must all be above 1\npop_size: 1000\n"
                    "core_size: 1200000\npan_genes: 6000\nn_gen: 0\nmax_distances: 100000\n"),
    (dict(core_mu=1.5), "core_mu must be between 0.0 and 1.0\ncore_mu: 1.5\n"),
    (dict(avg_gene_freq=0.0), "avg_gene_freq must be above 0.0 and below or equal to 1.0\navg_gene_freq: 0\n"),
]


@pytest.mark.parametrize("kw,text", VALIDATION)
def test_validation_messages(pa, kw, text):
    ok, msg = pa.validate(pa.make_params(**kw))
    assert not ok and msg == text


def test_validation_order_and_defaults(pa):
    assert pa.validate(pa.make_params()) == (True, "")
    # first failing check wins (main.rs:195-247 order)
    ok, msg = pa.validate(pa.make_params(pan_genes=600, HR_rate=-1.0, core_mu=2.0))
    assert msg.startswith("core_genes must be")
    ok, msg = pa.validate(pa.make_params(prop_genes2=2.0, core_mu=2.0))
    assert msg.startswith("prop_genes2 must be")
    p = pa.make_params()
    for k, v in pa.DEFAULTS.items():
        assert getattr(p, k) == v


def test_seeded_host_draws_match_oracle(pa, orc):
    for seed in (0, 1, 2**40 + 17):
        assert np.array_equal(pa.init_vector(seed, True, 5000), orc.init_core_vec(seed, 5000))
        assert np.array_equal(pa.init_vector(seed, True, 100, col_offset=4000), orc.init_core_vec(seed, 5000)[4000:4100])
        assert np.array_equal(pa.init_vector(seed, False, 4000, 0.25), orc.init_acc_vec(seed, 4000, 0.25))
        r1, r2 = pa.sample_pairs(seed, 1000, 5000)
        o1, o2 = orc.sample_pairs(seed, 1000, 5000)
        assert np.array_equal(r1, o1) and np.array_equal(r2, o2)
        for pp in (-0.1, 0.0, 0.3, 1.0):
            assert np.array_equal(pa.selection_coefficients(seed, 3000, pp, 10.0, 4.0),
                                  orc.selection_coefficients(seed, 3000, pp, 10.0, 4.0))
    v = pa.init_vector(0, True, 100000)
    assert set(np.unique(v)) == {1, 2, 4, 8}
    assert abs(pa.init_vector(0, False, 100000, 0.25).mean() - 0.25) < 0.01


def test_sample_weights_and_draws_match_oracle(pa, orc):
    rng = np.random.default_rng(3)
    N, G = 500, 300
    for trial in range(12):
        ng = rng.integers(50, 150, N).astype(np.int32)
        lw = rng.normal(0, 2, N) if trial % 2 else np.zeros(N)
        avg = rng.random(N) + 1e-3 if trial % 3 == 0 else np.ones(N)
        comp = 10.0 if trial % 3 == 0 else 0.0
        noc = bool(trial % 4 == 1)
        w = pa.sample_weights(ng, lw, G, 100, avg, noc, 0.99, comp)
        rc, ow = orc.sample_weights(ng, lw, G, 100, avg, noc, 0.99, comp)
        assert rc == 0 and np.array_equal(w, ow)
        idx = pa.draw_parents(w, 77, trial)
        rc, oidx = orc.draw_parents(ow, 77, trial)
        assert np.array_equal(idx, oidx) and idx.max() < N
    with pytest.raises(pa.PansimError) as e:                 # WeightedIndex::new panic, population.rs:440
        pa.sample_weights(ng, lw, G, 100, np.ones(N), False, -1.0, 0.0)
    assert e.value.code == -4
    # all-zero weights fall back to uniform (population.rs:403, :435-437)
    lw2 = np.zeros(N)
    big = np.full(N, 10**6, np.int32)
    w = pa.sample_weights(big, lw2, G, 0, np.ones(N), False, 1e-300, 0.0)
    rc, ow = orc.sample_weights(big, lw2, G, 0, np.ones(N), False, 1e-300, 0.0)
    assert rc == 0 and np.array_equal(w, ow)


@pytest.mark.parametrize("N", [1, 2, 7, 8, 9, 63, 1023, 1025, 5001])
def test_draw_parents_batches_of_eight_match_oracle(pa, orc, N):
    # the library draws eight parents at a time (one Philox block per two draws, eight binary searches in lockstep
    # through fixed power-of-two steps); the oracle draws them one by one (population.rs:440-443): sizes around the
    # batch and the step boundaries, flat / skewed / mostly-zero weights, draws exactly on cumulative boundaries
    rng = np.random.default_rng(N)
    for trial, w in enumerate((np.ones(N), rng.random(N) ** 8 + 1e-12, np.where(rng.random(N) < 0.9, 0.0, 1.0) + (np.arange(N) == N - 1),
                               np.full(N, 0.5 ** 20))):
        idx = pa.draw_parents(w.astype(np.float64), 5, trial)
        rc, oidx = orc.draw_parents(w.astype(np.float64), 5, trial)
        assert rc == 0 and np.array_equal(idx, oidx) and idx.max() < N


@pytest.mark.parametrize("N", [40000, 65536])
def test_sample_weights_large_populations_match_oracle(pa, orc, N):
    # the library takes the independent exp / ln calls of the three softmaxes on several host threads and N
    # equal inputs through one exp (population.rs:325-393); the oracle runs the plain sequential loops: every
    # weight must still be the same double
    rng = np.random.default_rng(N)
    G = 4000
    cases = [
        (rng.integers(900, 1100, N), np.zeros(N), np.ones(N), 0.0, False),              # the default path (neutral)
        (rng.integers(900, 1100, N), rng.normal(0, 3, N), np.ones(N), 0.0, False),      # selection
        (rng.integers(900, 1100, N), rng.normal(0, 3, N), rng.random(N) + 1e-6, 25.0, True),   # competition, no size control
        (np.full(N, 1000), np.zeros(N), np.full(N, 0.37), 3.0, False),                  # every vector constant
        (rng.integers(0, 4000, N), np.where(rng.random(N) < 0.01, -np.inf, rng.normal(0, 1, N)), np.ones(N), 0.0, False),
    ]
    for ng, lw, avg, comp, noc in cases:
        ng = ng.astype(np.int32)
        w = pa.sample_weights(ng, lw, G, 1000, avg, noc, 0.99, comp)
        rc, ow = orc.sample_weights(ng, lw, G, 1000, avg, noc, 0.99, comp)
        assert rc == 0 and np.array_equal(w, ow)
    # a -0.0 among +0.0 log-weights takes the constant-vector path: same doubles as the plain loop
    lw = np.zeros(N)
    lw[::7] = -0.0
    w = pa.sample_weights(cases[0][0].astype(np.int32), lw, G, 1000, np.ones(N), False, 0.99, 0.0)
    rc, ow = orc.sample_weights(cases[0][0].astype(np.int32), lw, G, 1000, np.ones(N), False, 0.99, 0.0)
    assert rc == 0 and np.array_equal(w, ow)


def test_format_and_small_helpers(pa, orc):
    for x, want in KAT["rust_display_f64"]:
        assert pa.fmt_f64(x) == want
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.random(300), rng.random(100) * 1e-12, rng.random(100) * 1e15, -rng.random(20)])
    for x in xs:
        assert pa.fmt_f64(x) == orc.fmt_f64(x)
    for x in (float("nan"), float("inf"), float("-inf"), 5e-324, 1.7976931348623157e308, 2.2250738585072014e-308):
        assert pa.fmt_f64(x) == orc.fmt_f64(x)
    for k, v in KAT["int_to_base"].items():
        assert pa.int_to_base(int(k)) == v
    v = KAT["standard_deviation"]
    assert pa.standard_deviation(v["values"]) == (v["std"], v["mean"])


def test_poisson_table_matches_oracle_and_poisson(pa, orc):
    # the library's and the oracle's independent restatements give identical integer tables, and the
    # table is the Poisson law: mean/variance = lambda, monotone thresholds ending at 2^32 - 1
    from pansim_amd import _lib
    for lam in (0.003, 0.7, 3.0, 27.0, 299.99999999999994, 2700.0, 27000.0):
        kmin_o, thr_o = orc.poisson_table(lam)
        cap = len(thr_o) + 8
        thr = np.zeros(cap, np.uint32)
        kmin = np.zeros(1, np.uint32)
        n = _lib.load().ps_poisson_table(lam, kmin, thr, cap)
        assert n == len(thr_o) and int(kmin[0]) == kmin_o
        assert np.array_equal(thr[:n], thr_o)
        t = thr_o.astype(np.float64)
        assert (np.diff(t) >= 0).all() and thr_o[-1] == 2**32 - 1
        pmf = np.diff(np.concatenate([[0.0], t])) / 2.0**32
        k = kmin_o + np.arange(len(t))
        mean = (pmf * k).sum()
        var = (pmf * (k - mean) ** 2).sum()
        assert abs(mean - lam) < 1e-6 * max(lam, 1.0) + 1e-8 and abs(var - lam) < 1e-5 * max(lam, 1.0) + 1e-7
        # draws: u = 0 gives the smallest value with mass, u = 2^32 - 1 the largest
        assert orc.poisson_from_table(0, kmin_o, thr_o) == kmin_o + int(np.argmax(thr_o > 0))
        assert orc.poisson_from_table(2**32 - 1, kmin_o, thr_o) == kmin_o + len(thr_o) - 1
    assert _lib.load().ps_poisson_table(0.0, np.zeros(1, np.uint32), np.zeros(8, np.uint32), 8) == 0


@pytest.mark.parametrize("penalty", [0.99, 1.0, 1.03, 1e-300, 0.5])
def test_size_softmax_by_gene_count_matches_the_plain_loops(pa, orc, penalty):
    # round 5: for N >= 4096 the genome-size softmax (population.rs:346-361) does every exp and division once per DISTINCT gene
    # count and looks it up by the count (ps_sample_weights / softmax_size); the oracle runs the plain loops over all N.  Same
    # doubles: penalties on both sides of 1 (the running maximum of the streaming ln_sum_exp moves up or down the counts),
    # exactly 1 (all arguments 0), one that underflows every weight but the lightest genome's; narrow and wide count ranges;
    # a count range too wide for the table (generic path); N just above and below the switch
    rng = np.random.default_rng(int(penalty * 1000) % 9973)
    for N, lo, hi in ((4096, 990, 1010), (4095, 990, 1010), (20000, 0, 4000), (9000, 5, 6), (5000, -3000000, 3000000)):
        ng = rng.integers(lo, hi + 1, N).astype(np.int32)
        ng[0], ng[-1] = hi, lo
        lw = np.where(rng.random(N) < 0.5, 0.0, rng.normal(0, 2, N))
        for avg_num in (1000, 0):
            w = pa.sample_weights(ng, lw, 4000, avg_num, np.ones(N), False, penalty, 0.0)
            rc, ow = orc.sample_weights(ng, lw, 4000, avg_num, np.ones(N), False, penalty, 0.0)
            assert rc == 0 and np.array_equal(w, ow), (N, lo, hi, avg_num)
