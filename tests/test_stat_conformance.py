"""Distributional conformance: the reference's mutation/recombination draw from
thread_rng() (population.rs:493, :517, :596), so only their DISTRIBUTIONS are contractual.  The
keyed dense form the HIP kernels implement (restated in the oracle) is checked here against
the closed forms of SURVEY 8(a) and against the event-driven reference algorithm (orc_ref_*).
All CPU; the GPU path is bit-exact with the dense oracle (tests/test_gpu_parity.py).
"""
import math

import numpy as np


def test_core_mutation_rate_and_alleles(orc):
    N, L, lam = 200, 20000, 1000.0                      # p = 1 - exp(-0.05)
    m0 = np.ones((N, L), np.uint8)
    plan = orc.core_plan(lam, 0.0, L)
    m = orc.mutate_core(m0.copy(), 0, 123, 0, plan)
    p = -math.expm1(-lam / L)
    hit = (m != 1)
    n = N * L
    assert abs(hit.mean() - p) < 5 * math.sqrt(p * (1 - p) / n)
    # mean events per individual = lambda up to the multiple-hit correction (Poisson splitting)
    per_ind = hit.sum(1)
    assert abs(per_ind.mean() - L * p) < 5 * math.sqrt(L * p / N)
    # Binomial dispersion var/mean = 1-p; the per-site counts give 20000 samples of it
    per_site = hit.sum(0)
    assert abs(per_site.var() / per_site.mean() - (1 - p)) < 0.05
    assert abs(per_ind.var() / per_ind.mean() - (1 - p)) < 0.4
    counts = np.array([(m == a).sum() for a in (2, 4, 8)])
    assert (m[hit] != 1).all() and abs(counts / counts.sum() - 1 / 3).max() < 0.005   # never 'A' (App. B.1)
    # different generations / seeds are independent streams
    m2 = orc.mutate_core(m0.copy(), 0, 123, 1, plan)
    both = ((m != 1) & (m2 != 1)).mean()
    assert abs(both - p * p) < 5 * math.sqrt(p * p / n)


def test_core_recombination_rate_donor_and_snapshot(orc):
    N, L, lam_hr = 64, 30000, 3000.0                    # q = 1 - exp(-0.1)
    rng = np.random.default_rng(0)
    # every individual carries a private marker so that the donor of a copied cell is visible
    m0 = np.tile(np.arange(N, dtype=np.uint8)[:, None] + 16, (1, L))
    plan = orc.core_plan(0.0, lam_hr, L)
    m = orc.recombine_core(m0.copy(), 0, 5, 0, plan)
    q = -math.expm1(-lam_hr / L)
    changed = m != m0
    assert abs(changed.mean() - q) < 5 * math.sqrt(q / (N * L))
    donors = (m[changed].astype(int) - 16)
    recips = np.nonzero(changed)[0]
    assert (donors != recips).all()                                  # recipient != donor (population.rs:618)
    hist = np.bincount(donors, minlength=N)
    assert abs(hist / hist.sum() - 1 / N).max() < 0.004             # donors uniform over the others
    # snapshot semantics (population.rs:693-695): the copied allele is the donor's PRE-recombination one
    assert set(np.unique(m)) <= set(range(16, 16 + N))
    # joint plan: mutation and HR of a cell are independent
    plan2 = orc.core_plan(3000.0, 3000.0, L)
    ones = np.ones((N, L), np.uint8)
    mm = orc.mutate_core(ones.copy(), 0, 9, 0, plan2)
    mr = orc.recombine_core(m0.copy(), 0, 9, 0, plan2)
    a, b = (mm != 1), (mr != m0)
    pa_, pb_ = a.mean(), b.mean()
    assert abs((a & b).mean() - pa_ * pb_) < 6 * math.sqrt(pa_ * pb_ / (N * L))


def test_acc_flip_probabilities(orc):
    N, G = 400, 1000
    cb, ce = [0, 900], [900, 1000]
    lam = [900.0, 100000.0]                              # rates 1.0 and 1000 per site (main.rs:348, :361)
    m0 = np.zeros((N, G), np.uint8)
    m = orc.mutate_acc(m0.copy(), 3, 0, cb, ce, lam)
    p1 = (1 - math.exp(-2.0)) / 2                         # 0.43233235838169365
    f1, f2 = m[:, :900].mean(), m[:, 900:].mean()
    assert abs(f1 - p1) < 5 * math.sqrt(p1 * (1 - p1) / (N * 900))
    assert abs(f2 - 0.5) < 5 * math.sqrt(0.25 / (N * 100))
    # a zero rate leaves the compartment untouched (population.rs:480)
    m = orc.mutate_acc(m0.copy(), 3, 0, cb, ce, [0.0, 50.0])
    assert m[:, :900].sum() == 0 and m[:, 900:].sum() > 0


def test_hgt_events(orc):
    N, G = 200, 300
    cb, ce = [0, 200], [200, 300]
    rng = np.random.default_rng(1)
    m0 = (rng.random((N, G)) < 0.2).astype(np.uint8)
    m0[:, 250:] = 0                                       # genes nobody carries can never be gained
    m0[5, :] = 0                                          # a donor with no genes transfers nothing
    lam = [40.0, 10.0]
    m = m0.copy()
    K = orc.recombine_acc(m, 8, 0, cb, ce, lam)
    assert abs(K - N * 50.0) < 6 * math.sqrt(N * 50.0)   # total events ~ Poisson(N * lambda)
    assert (m >= m0).all()                                # gain only (App. B.4)
    assert m[:, 250:].sum() == 0                          # loci come from the donors' present genes
    gained = (m > m0)
    # expected gains: a recipient lacking a gene gains it w.p. 1-exp(-sum_d lam/((N-1) n_d) [d has g])
    n1 = m0[:, :200].sum(1).astype(float)
    rate_g = (lam[0] / (N - 1)) * (m0[:, :200] / np.maximum(n1, 1)[:, None]).sum(0)
    exp_gain = ((1 - m0[:, :200]) * (1 - np.exp(-rate_g))[None, :]).sum()
    got = gained[:, :200].sum()
    assert abs(got - exp_gain) < 6 * math.sqrt(exp_gain)


def test_dense_form_matches_event_driven_reference_algorithm(orc):
    """The reference algorithm (Poisson count per row + weighted-index draws, orc_ref_*) and the
    keyed dense form agree in distribution: per-cell change rates after one generation."""
    p = orc.make_params(pop_size=120, core_size=6000, pan_genes=500, core_genes=100, HR_rate=0.5, HGT_rate=0.5)
    ref = orc.RefSim(p, seed=2, threads=2)
    before_c, before_a = ref.core().copy(), ref.acc().copy()
    ref.generation(0)
    ev_c, ev_a = ref.core().copy(), ref.acc().copy()
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from orc_sim import OracleSim
    dn = OracleSim(seed=2, pop_size=120, core_size=6000, pan_genes=500, core_genes=100, HR_rate=0.5, HGT_rate=0.5)
    assert np.array_equal(dn.core, before_c) and np.array_equal(dn.acc, before_a)   # same clonal start
    dn.generation(0)
    # both started clonal, so gather is invisible and differences are mutations/HR only
    n = before_c.size
    fe, fd = (ev_c != before_c).mean(), (dn.core != before_c).mean()
    # event form: a hit may rewrite the same allele (prob 1/3 when the site is not 'A'); identical
    # in both forms, so the visible-change rates must agree
    assert abs(fe - fd) < 6 * math.sqrt(fd / n)
    for a in (2, 4, 8):
        ce_, cd_ = (ev_c == a).mean(), (dn.core == a).mean()
        assert abs(ce_ - cd_) < 6 * math.sqrt(cd_ / n)
    # accessory: flip rates per compartment and overall gene frequency
    g1 = dn.d.comp_end[0]
    for sl in (slice(0, g1), slice(g1, dn.G)):
        fe, fd = (ev_a[:, sl] != before_a[:, sl]).mean(), (dn.acc[:, sl] != before_a[:, sl]).mean()
        cells = before_a[:, sl].size
        assert abs(fe - fd) < 6 * math.sqrt(max(fd * (1 - fd), 1e-4) / cells) + 0.003


def test_parent_draw_distribution(orc):
    N = 50
    rng = np.random.default_rng(7)
    w = rng.random(N)
    counts = np.zeros(N)
    for gen in range(400):
        rc, idx = orc.draw_parents(w, 1, gen)
        counts += np.bincount(idx, minlength=N)
    exp = w / w.sum() * counts.sum()
    chi2 = ((counts - exp) ** 2 / exp).sum()
    assert chi2 < 110                                      # 49 dof, far tail


def test_poisson_sampler_moments(orc):
    for mean in (0.5, 3.0, 9.9, 10.0, 50.0, 3000.0, 2.7e6):
        ks = np.array([orc.lib().orc_poisson(mean, 42, 21, g, None) for g in range(4000)], float)
        assert abs(ks.mean() - mean) < 6 * math.sqrt(mean / 4000)
        assert abs(ks.var() / mean - 1) < 0.15
