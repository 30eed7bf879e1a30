"""GPU parity: every operator of the C ABI (HIP kernels) against the CPU oracle on the same
seeded inputs.  Integer / byte / index work must be bit-exact; f64 outputs are produced by
identical IEEE expression shapes and are compared exactly as well.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_core(rng, N, L):
    return (1 << rng.integers(0, 4, size=(N, L))).astype(np.uint8)


def _rand_acc(rng, N, G, p=0.3):
    return (rng.random((N, G)) < p).astype(np.uint8)


# ----------------------------------------------------------------------------- D-kern
def test_hamming_jaccard_kat(pa):
    # SURVEY 8(c) H1, H2, J1 (hand-derived from distances.rs:22-77)
    x = [1, 2, 4, 8, 1, 2, 4, 8, 1]
    y = [1, 2, 4, 8, 1, 2, 4, 8, 2]
    assert pa.hamming_bitwise_fast(x, y) == 2
    x = [1, 2, 4, 8, 1, 2, 4, 8, 1, 2, 4, 8, 1, 2, 4, 8, 8]
    y = [2, 2, 4, 1, 1, 2, 8, 8, 1, 2, 4, 8, 1, 4, 4, 8, 1]
    assert pa.hamming_bitwise_fast(x, y) == 10
    a = [1, 1, 0, 0, 1, 0, 0, 0, 1]
    b = [1, 0, 1, 0, 1, 0, 0, 0, 0]
    assert pa.jaccard_distance_fast(a, b) == (2, 5)
    assert pa.hamming_bitwise_fast([], []) == 0


def test_hamming_jaccard_random(pa, orc):
    rng = np.random.default_rng(1)
    for n in (1, 7, 8, 9, 1000, 123457):
        x = rng.integers(0, 256, n).astype(np.uint8)
        y = rng.integers(0, 256, n).astype(np.uint8)
        assert pa.hamming_bitwise_fast(x, y) == orc.hamming(x, y)
        assert pa.jaccard_distance_fast(x, y) == orc.jaccard(x, y)


# ----------------------------------------------------------------------------- S-state
@pytest.mark.parametrize("N,L", [(2, 1), (100, 777), (1000, 130), (1024, 65), (1500, 200), (4100, 33)])
def test_core_load_read_roundtrip(pa, N, L):
    rng = np.random.default_rng(N * 7 + L)
    m = _rand_core(rng, N, L)
    pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
    pop.load_matrix(m)
    assert np.array_equal(pop.read_matrix(), m)
    pop.close()


@pytest.mark.parametrize("N,G", [(2, 1), (100, 400), (1000, 4000), (130, 64), (65, 129)])
def test_acc_load_read_roundtrip(pa, N, G):
    rng = np.random.default_rng(N * 3 + G)
    m = _rand_acc(rng, N, G)
    pop = pa.Population(N, G, 2, False, 0.5, 0, 10)
    pop.load_matrix(m)
    assert np.array_equal(pop.read_matrix(), m)
    pop.close()


def test_clonal_init_matches_oracle(pa, orc):
    N, L, G = 100, 5000, 400
    core = pa.Population(N, L, 4, True, 0.0, 5, 0)
    assert np.array_equal(core.read_matrix(), np.tile(orc.init_core_vec(5, L), (N, 1)))
    acc = pa.Population(N, G, 2, False, 0.25, 5, 200)
    assert np.array_equal(acc.read_matrix(), np.tile(orc.init_acc_vec(5, G, 0.25), (N, 1)))
    # a site shard draws the same global vector
    sh = pa.Population(N, 1000, 4, True, 0.0, 5, 0, col_offset=3000, global_cols=L)
    assert np.array_equal(sh.read_matrix(), np.tile(orc.init_core_vec(5, L)[3000:4000], (N, 1)))


# ----------------------------------------------------------------------------- G-gather
@pytest.mark.parametrize("N,L", [(3, 5), (100, 1200), (1000, 300), (1500, 100)])
def test_next_generation_core(pa, orc, N, L):
    rng = np.random.default_rng(N + L)
    m = _rand_core(rng, N, L)
    sample = rng.integers(0, N, N).astype(np.uint32)
    pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
    pop.load_matrix(m)
    pop.next_generation(sample)
    assert np.array_equal(pop.read_matrix(), orc.next_generation(m, sample))


def test_next_generation_kat(pa):
    # sample=[2,2,0]
    m = np.array([[1, 2, 4], [8, 8, 8], [2, 1, 2]], np.uint8)
    pop = pa.Population(3, 3, 4, True, 0.0, 0, 0)
    pop.load_matrix(m)
    pop.next_generation([2, 2, 0])
    assert pop.read_matrix().tolist() == [[2, 1, 2], [2, 1, 2], [1, 2, 4]]


@pytest.mark.parametrize("N,G", [(100, 400), (1000, 700), (130, 64)])
def test_next_generation_acc(pa, orc, N, G):
    rng = np.random.default_rng(N + G)
    m = _rand_acc(rng, N, G)
    sample = rng.integers(0, N, N).astype(np.uint32)
    pop = pa.Population(N, G, 2, False, 0.5, 0, 0)
    pop.load_matrix(m)
    pop.next_generation(sample)
    assert np.array_equal(pop.read_matrix(), orc.next_generation(m, sample))


# ----------------------------------------------------------------------------- M-core / R-core
CORE_CASES = [
    # N, L_local, L_global, offset, lam_mut, lam_hr
    (100, 1200, 12000, 0, 600.0, 30.0),
    (100, 500, 12000, 7000, 600.0, 6000.0),
    (1000, 400, 1200000, 1000000, 60000.0, 3000.0),
    (1000, 200, 4000, 0, 2000.0, 2000.0),        # heavy rates: many candidates per chunk
    (1500, 150, 3000, 100, 150.0, 15.0),         # block sweep, two segments per row
    (5000, 60, 60000, 0, 3000.0, 3000.0),        # block sweep, five segments, partial last one
    (20000, 9, 900, 0, 45.0, 200.0),             # block sweep, one row per workgroup iteration
    (70000, 5, 500, 0, 5.0, 1.0),                # block sweep, 69 segments: 16 waves, two batches for some
    (65536, 3, 1200000, 77, 60000.0, 3000.0),    # cfg4 population at the default rates (LDS exactly full)
    (17, 64, 64, 0, 64.0, 0.0),
    (2, 300, 300, 0, 30.0, 30.0),
]


@pytest.mark.parametrize("N,L,LG,off,lm,lh", CORE_CASES)
def test_core_operators_match_oracle(pa, orc, N, L, LG, off, lm, lh):
    rng = np.random.default_rng(N * 31 + L)
    m0 = _rand_core(rng, N, L)
    plan = orc.core_plan(lm, lh, LG)
    seed, gen = 1234567890123, 7
    pop = pa.Population(N, L, 4, True, 0.0, seed, 0, col_offset=off, global_cols=LG)
    pop.set_rates([lm], [lh])
    # mutate only
    pop.load_matrix(m0)
    pop.mutate_alleles(gen)
    want_mut = orc.mutate_core(m0.copy(), off, seed, gen, plan)
    got = pop.read_matrix()
    assert np.array_equal(got, want_mut)
    assert (got != m0).any()
    # recombine only, on the post-mutation state
    pop.recombine(gen)
    want_rec = orc.recombine_core(want_mut.copy(), off, seed, gen, plan)
    assert np.array_equal(pop.read_matrix(), want_rec)
    # fused == gather, mutate, recombine in order
    sample = rng.integers(0, N, N).astype(np.uint32)
    pop.load_matrix(m0)
    pop.step(gen + 1, sample, True)
    w = orc.next_generation(m0, sample)
    orc.mutate_core(w, off, seed, gen + 1, plan)
    orc.recombine_core(w, off, seed, gen + 1, plan)
    assert np.array_equal(pop.read_matrix(), w)
    # fused without recombination
    pop.load_matrix(m0)
    pop.step(gen + 2, sample, False)
    w = orc.next_generation(m0, sample)
    orc.mutate_core(w, off, seed, gen + 2, plan)
    assert np.array_equal(pop.read_matrix(), w)
    pop.close()


def test_core_mutation_never_writes_A(pa):
    # `1 >> value` always selects [2,4,8]
    N, L = 200, 2000
    pop = pa.Population(N, L, 4, True, 0.0, 3, 0)
    pop.load_matrix(np.ones((N, L), np.uint8))
    pop.set_rates([L * 0.5], [0.0])
    pop.mutate_alleles(0)
    m = pop.read_matrix()
    assert set(np.unique(m)) <= {1, 2, 4, 8}
    changed = m[m != 1]
    assert changed.size > 0 and set(np.unique(changed)) == {2, 4, 8}


# ----------------------------------------------------------------------------- M-acc / R-acc
ACC_CASES = [
    # N, G, comps(begin,end), lam_mut, lam_rec
    (100, 400, [(0, 360), (360, 400)], [360.0, 40000.0], [27.0, 3.0]),
    (1000, 4000, [(0, 3600), (3600, 4000)], [3600.0, 400000.0], [2700.0, 299.99999999999994]),
    (130, 200, [(0, 200)], [20.0], [500.0]),
    (70, 129, [(0, 64), (64, 129)], [0.0, 10.0], [5.0, 0.0]),
    (50, 300, [(0, 100), (100, 300)], [1.0, 1.0], [2000.0, 2000.0]),
    (60, 9000, [(0, 8100), (8100, 9000)], [10.0, 10.0], [300.0, 50.0]),   # 141 row words: 3 chunks of group entries
]


@pytest.mark.parametrize("N,G,comps,lm,lr", ACC_CASES)
def test_acc_operators_match_oracle(pa, orc, N, G, comps, lm, lr):
    rng = np.random.default_rng(N * 13 + G)
    m0 = _rand_acc(rng, N, G, 0.25)
    m0[0, :] = 0            # a donor with no genes at all (population.rs:672)
    cb = [c[0] for c in comps]
    ce = [c[1] for c in comps]
    seed, gen = 42, 3
    pop = pa.Population(N, G, 2, False, 0.25, seed, 10)
    pop.set_rates(lm, lr, cb, ce)
    pop.load_matrix(m0)
    pop.mutate_alleles(gen)
    want = orc.mutate_acc(m0.copy(), seed, gen, cb, ce, lm)
    assert np.array_equal(pop.read_matrix(), want)
    pop.recombine(gen)
    want2 = want.copy()
    orc.recombine_acc(want2, seed, gen, cb, ce, lr)
    got2 = pop.read_matrix()
    assert np.array_equal(got2, want2)
    # every HGT form applies the same keyed events: atomics (donor lists in LDS / in global scratch)
    # and the binned two-pass form with one recipient partition and with ~N/21 partitions
    row_bytes = 8 * ((G + 63) // 64)
    for tune in ({"hgt_mode": 1}, {"hgt_mode": 1, "hgt_list_in_global": 1}, {"hgt_mode": 2},
                 {"hgt_mode": 2, "hgt_bin_list_in_global": 1},
                 {"hgt_mode": 2, "lds_limit": min(160 * 1024, max(1024 + 21 * row_bytes, 2 * G + 8192))},
                 {"hgt_mode": 2, "hgt_slices": 3}):
        alt = pa.Population(N, G, 2, False, 0.25, seed, 10)
        for key, val in tune.items():
            alt.set_tuning(key, val)
        alt.set_rates(lm, lr, cb, ce)
        alt.load_matrix(want)
        alt.recombine(gen)
        assert np.array_equal(alt.read_matrix(), want2)
        alt.close()
    assert (got2 >= want).all()          # HGT never clears a gene
    # gene-major view stayed coherent with the individual-major one
    assert np.array_equal(pop.gene_frequencies()[:G], want2.sum(0) / N)
    sample = rng.integers(0, N, N).astype(np.uint32)
    pop.load_matrix(m0)
    pop.step(gen + 1, sample, True)
    w = orc.next_generation(m0, sample)
    orc.mutate_acc(w, seed, gen + 1, cb, ce, lm)
    orc.recombine_acc(w, seed, gen + 1, cb, ce, lr)
    assert np.array_equal(pop.read_matrix(), w)
    pop.close()


# ----------------------------------------------------------------------------- P-*
def test_fitness_terms_and_sample_indices(pa, orc):
    N, G = 300, 500
    rng = np.random.default_rng(5)
    m = _rand_acc(rng, N, G, 0.4)
    sel = orc.selection_coefficients(9, G, 0.3, 10.0, 10.0)
    sel[7] = -1.0                      # the -inf reset rule (population.rs:312-318)
    pop = pa.Population(N, G, 2, False, 0.4, 11, 20)
    pop.load_matrix(m)
    ng, lw = pop.fitness_terms(sel)
    ong, olw = orc.fitness_terms(m, sel)
    assert np.array_equal(ng, ong)
    assert np.array_equal(lw, olw)
    assert (lw[m[:, 7] == 1] == 0.0).all()
    for comp, avg in ((0.0, np.ones(N)), (5.0, rng.random(N) + 0.1)):
        for no_control in (False, True):
            idx = pop.sample_indices(4, 200, avg, sel, False, no_control, 0.99, comp)
            rc, oidx = orc.sample_indices(m, 11, 4, 200, avg, sel, no_control, 0.99, comp)
            assert rc == 0
            assert np.array_equal(idx, oidx)
    # neutral coefficients
    z = np.zeros(G)
    idx = pop.sample_indices(0, 200, np.ones(N), z)
    rc, oidx = orc.sample_indices(m, 11, 0, 200, np.ones(N), z)
    assert np.array_equal(idx, oidx)


# ----------------------------------------------------------------------------- D-pairs / D-avg / F-freq
@pytest.mark.parametrize("N,L,P", [(100, 1203, 5000), (1000, 777, 40000), (1100, 300, 3000),
                                   (1500, 333, 2000), (6000, 40, 1000)])
def test_pairwise_core(pa, orc, N, L, P):
    rng = np.random.default_rng(N + L + P)
    m = _rand_core(rng, N, L)
    r1, r2 = orc.sample_pairs(3, N, P)
    pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
    pop.load_matrix(m)
    got = pop.pairwise_distances(P, r1, r2)
    assert np.array_equal(got, orc.pairwise_distances(m, True, 0, r1, r2))
    (cnt,) = pop.pairwise_counts(r1, r2)
    assert np.array_equal(cnt, orc.pairwise_hamming_counts(m, 0, L, r1, r2))


@pytest.mark.parametrize("N,L,P", [(100, 1203, 5000), (1000, 777, 40000), (300, 5000, 1000), (1500, 333, 2000)])
def test_pairwise_core_kernels_agree(pa, orc, N, L, P):
    # sampled-pair kernels (2-bit packed form for one-hot matrices, nibble form) and all-pairs
    # tiles + lookup give the same integers
    rng = np.random.default_rng(N + L)
    m = _rand_core(rng, N, L)
    r1, r2 = orc.sample_pairs(5, N, P)
    want = orc.pairwise_hamming_counts(m, 0, L, r1, r2)
    for mode in (1, 2, 3, 5, 6, 7):
        pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
        pop.set_tuning("pair_mode", mode)
        pop.load_matrix(m)
        assert np.array_equal(pop.pairwise_counts(r1, r2)[0], want)
        assert pop.last_pair_form() == {1: 1, 2: 7, 3: 3, 5: 2, 6: 6, 7: 8}[mode]
        pop.close()


@pytest.mark.parametrize("N,L", [(2, 1), (33, 130), (257, 517), (300, 4100), (700, 9000), (1030, 2500)])
def test_allpairs_matrix_core_form(pa, orc, N, L):
    # the i8 MFMA all-pairs form (one-hot X X^T, population.rs:787-837 / distances.rs:22-52): every ordered pair
    # (i, j) incl. i == j and both orders, populations and site counts that are not multiples of the 256-tiles /
    # 128-site chunks, rows that differ in a structured (asymmetric) way so that a row <-> column swap or a wrong
    # accumulator row map shows
    rng = np.random.default_rng(N * 31 + L)
    m = _rand_core(rng, N, L)
    for i in range(N):                       # individual i agrees with individual 0 on its first (13 i) % L sites
        k = (13 * i) % (L + 1)
        m[i, :k] = m[0, :k]
    ii, jj = np.meshgrid(np.arange(N, dtype=np.uint32), np.arange(N, dtype=np.uint32), indexing="ij")
    r1, r2 = ii.ravel(), jj.ravel()
    if r1.size > 300000:
        sel = rng.choice(r1.size, 300000, replace=False)
        r1, r2 = r1[sel], r2[sel]
    want = orc.pairwise_hamming_counts(m, 0, L, r1, r2)
    pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
    pop.load_matrix(m)
    for mode, form in ((6, 6), (2, 7), (7, 8)):          # i8, the block-scaled FP4 form on one-hot nibbles, on +-1 features
        pop.set_tuning("pair_mode", mode)
        (got,) = pop.pairwise_counts(np.ascontiguousarray(r1), np.ascontiguousarray(r2))
        assert pop.last_pair_form() == form
        assert np.array_equal(got, want), "pair form %d" % form
    # a matrix that is not one-hot falls back to the xor + popcount tiles
    m[0, 0] = 3
    pop.load_matrix(m)
    (got,) = pop.pairwise_counts(np.ascontiguousarray(r1), np.ascontiguousarray(r2))
    assert pop.last_pair_form() == 2
    assert np.array_equal(got, orc.pairwise_hamming_counts(m, 0, L, r1, r2))
    pop.close()


def test_pairwise_core_arbitrary_bytes(pa, orc):
    # not one-hot: the generic kernel must reproduce popcount(x ^ y) / 2 (population.rs:817)
    N, L, P = 64, 99, 500
    rng = np.random.default_rng(0)
    m = rng.integers(0, 256, (N, L)).astype(np.uint8)
    r1, r2 = orc.sample_pairs(1, N, P)
    pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
    pop.load_matrix(m)
    assert np.array_equal(pop.pairwise_distances(P, r1, r2), orc.pairwise_distances(m, True, 0, r1, r2))


@pytest.mark.parametrize("N,L,P", [(130, 700, 3000), (1000, 513, 20000)])
def test_pairwise_core_nibble_bytes(pa, orc, N, L, P):
    # bytes below 16 that are not one-hot (0, 3, 5, 15 ...): the nibble kernels, sampled and all-pairs
    rng = np.random.default_rng(N + L)
    m = rng.integers(0, 16, (N, L)).astype(np.uint8)
    r1, r2 = orc.sample_pairs(2, N, P)
    want = orc.pairwise_hamming_counts(m, 0, L, r1, r2)
    for mode in (0, 1, 2):
        pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
        pop.set_tuning("pair_mode", mode)
        pop.load_matrix(m)
        (cnt,) = pop.pairwise_counts(r1, r2)
        assert np.array_equal(cnt, want)
        pop.close()


def test_pairwise_nibble_counter_cap(pa, orc):
    # worst case of the nibble kernel's 16-bit range counters: rows of 0x0F against rows of 0x00 differ in
    # 4 bits per site, so a range may hold at most 16383 sites (a cap sized for one-hot bytes -- 2 bits per
    # site -- let 79 tiles x 256 sites x 4 bits = 80896 carry into the neighbouring pair's counter)
    N, L, P = 130, 40000, 4000
    m = np.zeros((N, L), np.uint8)
    m[::2, :] = 0x0F
    r1, r2 = orc.sample_pairs(3, N, P)
    want = orc.pairwise_hamming_counts(m, 0, L, r1, r2)
    assert want.max() == 4 * L
    for ranges in (1, 2, 0):
        pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
        pop.set_tuning("pair_mode", 1)
        pop.set_tuning("pair_ranges", ranges)
        pop.load_matrix(m)
        assert np.array_equal(pop.pairwise_counts(r1, r2)[0], want)
        pop.close()


def test_pair_list_cache_follows_lds_limit(pa, orc):
    # the cached device copy of a pair list is sorted (with a thread table) only when the tiled kernel can
    # run; changing "lds_limit" between two calls with the same list must rebuild it
    N, L, P = 1000, 600, 5000
    rng = np.random.default_rng(8)
    m = _rand_core(rng, N, L)
    r1, r2 = orc.sample_pairs(4, N, P)
    want = orc.pairwise_hamming_counts(m, 0, L, r1, r2)
    pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
    pop.load_matrix(m)
    pop.set_tuning("pair_mode", 1)
    pop.set_tuning("lds_limit", 16 * 1024)     # no LDS tile for N = 1000: caller's order, no thread table
    assert np.array_equal(pop.pairwise_counts(r1, r2)[0], want)
    pop.set_tuning("lds_limit", 160 * 1024)    # tiled kernel: needs the sorted list
    assert np.array_equal(pop.pairwise_counts(r1, r2)[0], want)
    pop.set_tuning("lds_limit", 16 * 1024)
    assert np.array_equal(pop.pairwise_counts(r1, r2)[0], want)
    pop.close()


def test_population_api_does_not_disturb_sim_fitness_table(pa, orc):
    # ps_sim keeps its own ln(1+s) table: a Population-API call on the sim's accessory handle with other
    # selection coefficients must not change the parents the loop draws afterwards
    from orc_sim import OracleSim
    kw = dict(pop_size=90, core_size=600, pan_genes=300, core_genes=100)
    extra = dict(prop_positive=0.4)
    sim = pa.Simulation(pa.make_params(seed=5, n_gen=4, max_distances=100, **kw, **extra))
    ref = OracleSim(seed=5, **kw, **extra)
    sim.run(2)
    sim.sync()
    other = np.linspace(-0.5, 0.5, sim.pan_genome.ncols)
    sim.pan_genome.fitness_terms(other)
    sim.run(2)
    sim.sync()
    for g in range(4):
        ref.generation(g)
    assert np.array_equal(sim.last_parents(), ref.last_idx)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    sim.close()


def test_pairwise_kat(pa):
    # SURVEY 8(c) H1 -> 0.1111111111111111 ; J1 -> 0.6 / 0.4285714285714286 / 0.0014962593516208988
    x = [1, 2, 4, 8, 1, 2, 4, 8, 1]
    y = [1, 2, 4, 8, 1, 2, 4, 8, 2]
    pop = pa.Population(2, 9, 4, True, 0.0, 0, 0)
    pop.load_matrix(np.array([x, y], np.uint8))
    assert pop.pairwise_distances(1, [0], [1])[0] == 0.1111111111111111
    a = [1, 1, 0, 0, 1, 0, 0, 0, 1]
    b = [1, 0, 1, 0, 1, 0, 0, 0, 0]
    for cg, want in ((0, 0.6), (2, 0.4285714285714286), (2000, 0.0014962593516208988)):
        acc = pa.Population(2, 9, 2, False, 0.5, 0, cg)
        acc.load_matrix(np.array([a, b], np.uint8))
        assert acc.pairwise_distances(1, [0], [1])[0] == want
    # identical rows -> average distance clamps to f64::MIN_POSITIVE (population.rs:774-776)
    acc = pa.Population(3, 9, 2, False, 0.5, 0, 5)
    acc.load_matrix(np.array([a, a, a], np.uint8))
    assert (acc.average_distance() == 2.2250738585072014e-308).all()
    # empty union and no core genes -> NaN
    acc = pa.Population(2, 4, 2, False, 0.5, 0, 0)
    acc.load_matrix(np.zeros((2, 4), np.uint8))
    assert np.isnan(acc.pairwise_distances(1, [0], [1])[0])


@pytest.mark.parametrize("N,G,cg", [(100, 400, 200), (1000, 4000, 2000), (77, 65, 0), (300, 700, 50)])
def test_pairwise_average_freq_acc(pa, orc, N, G, cg):
    rng = np.random.default_rng(N + G)
    m = _rand_acc(rng, N, G, 0.25)
    r1, r2 = orc.sample_pairs(8, N, 3000)
    pop = pa.Population(N, G, 2, False, 0.25, 0, cg)
    pop.load_matrix(m)
    assert np.array_equal(pop.pairwise_distances(3000, r1, r2), orc.pairwise_distances(m, False, cg, r1, r2))
    assert np.array_equal(pop.gene_frequencies(), orc.gene_frequencies(m, cg))
    if N <= 300:
        assert np.array_equal(pop.average_distance(), orc.average_distance(m, False, cg))
    assert pop.calc_gene_freq() == orc.lib().orc_calc_gene_freq(m, N, G)


def test_gene_frequencies_kat(pa):
    m = np.array([[1, 0, 1], [1, 0, 0], [1, 1, 0], [1, 0, 0]], np.uint8)
    pop = pa.Population(4, 3, 2, False, 0.5, 0, 2)
    pop.load_matrix(m)
    assert pop.gene_frequencies().tolist() == [1.0, 0.25, 0.25, 1.0, 1.0]


# ----------------------------------------------------------------------------- W-out
def test_write_matrices(pa, orc, tmp_path):
    N, L, G = 5, 7, 6
    rng = np.random.default_rng(2)
    mc = _rand_core(rng, N, L)
    mc[0, 0] = 3                       # not an allele -> 'N' (population.rs:160)
    ma = _rand_acc(rng, N, G)
    core = pa.Population(N, L, 4, True, 0.0, 0, 2)
    core.load_matrix(mc)
    acc = pa.Population(N, G, 2, False, 0.5, 0, 2)
    acc.load_matrix(ma)
    core.write(str(tmp_path / "got"))
    acc.write(str(tmp_path / "got"))
    orc.lib().orc_write_matrix(mc, N, L, 1, 2, str(tmp_path / "want").encode())
    orc.lib().orc_write_matrix(ma, N, G, 0, 2, str(tmp_path / "want").encode())
    for suffix in ("_core_genome.csv", "_pangenome.csv"):
        assert (tmp_path / ("got" + suffix)).read_text() == (tmp_path / ("want" + suffix)).read_text()
    first = (tmp_path / "got_core_genome.csv").read_text().splitlines()[0]
    assert first.startswith("N,") and len(first.split(",")) == L
    assert (tmp_path / "got_pangenome.csv").read_text().splitlines()[0].startswith("1,1,")


@pytest.mark.parametrize("N,L", [(130, 777), (1000, 129), (3, 64), (70, 1)])
def test_write_core_matrix_tiles(pa, orc, tmp_path, N, L):
    # the device-side text expansion across tile boundaries, byte for byte against the oracle writer
    rng = np.random.default_rng(N + L)
    m = _rand_core(rng, N, L)
    m[rng.integers(0, N), rng.integers(0, L)] = 7          # not an allele -> 'N'
    core = pa.Population(N, L, 4, True, 0.0, 0, 0)
    core.load_matrix(m)
    core.write(str(tmp_path / "got"))
    orc.lib().orc_write_matrix(m, N, L, 1, 0, str(tmp_path / "want").encode())
    assert (tmp_path / "got_core_genome.csv").read_bytes() == (tmp_path / "want_core_genome.csv").read_bytes()


# ----------------------------------------------------------------------------- L-loop
SIM_CASES = [
    dict(pop_size=100, core_size=12000, pan_genes=600, core_genes=200),                   # config 1'
    dict(pop_size=100, core_size=3000, pan_genes=600, core_genes=200, HR_rate=0.5, HGT_rate=0.5),
    dict(pop_size=64, core_size=2000, pan_genes=300, core_genes=100, HR_rate=0.0, HGT_rate=0.0),
    dict(pop_size=1200, core_size=900, pan_genes=500, core_genes=100),                     # block path
]


@pytest.mark.parametrize("kw", SIM_CASES)
def test_generation_loop_matches_oracle(pa, orc, kw):
    from orc_sim import OracleSim
    sim = pa.Simulation(pa.make_params(seed=3, n_gen=6, max_distances=2000, **kw))
    ref = OracleSim(seed=3, **kw)
    r1, r2 = orc.sample_pairs(3, kw["pop_size"], 2000)
    assert np.array_equal(sim.range1, r1) and np.array_equal(sim.range2, r2)
    for g in range(6):
        sim.run(1)
        sim.sync()
        ref.generation(g)
        assert np.array_equal(sim.last_parents(), ref.last_idx)
    assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    core_d, acc_d = sim.final_distances()
    cg = kw["core_genes"]
    assert np.array_equal(core_d, orc.pairwise_distances(ref.core, True, cg, r1, r2))
    assert np.array_equal(acc_d, orc.pairwise_distances(ref.acc, False, cg, r1, r2))
    assert np.array_equal(sim.pan_genome.gene_frequencies(), orc.gene_frequencies(ref.acc, cg))
    sim.close()


def test_generation_loop_selection_and_competition(pa, orc):
    from orc_sim import OracleSim
    kw = dict(pop_size=80, core_size=1500, pan_genes=400, core_genes=100)
    extra = dict(prop_positive=0.3, competition_strength=50.0)
    sim = pa.Simulation(pa.make_params(seed=9, n_gen=5, max_distances=300, **kw, **extra))
    ref = OracleSim(seed=9, **kw, **extra)
    assert np.array_equal(sim.selection_weights, ref.sel)
    # run the 5 generations without host synchronisation in between
    sim.run(5)
    sim.sync()
    for g in range(5):
        ref.generation(g)
    assert np.array_equal(sim.last_parents(), ref.last_idx)
    assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)


def test_site_shards_equal_unsharded(pa):
    # 8(e): a site shard keyed on global site indices reproduces its columns of the full run
    kw = dict(pop_size=100, core_size=6001, pan_genes=600, core_genes=200, HR_rate=0.3)
    full = pa.Simulation(pa.make_params(seed=1, n_gen=4, max_distances=1000, **kw))
    full.run(4)
    want = full.core_genome.read_matrix()
    (cnt_full,) = full.core_genome.pairwise_counts(full.range1, full.range2)
    parts, cnt = [], np.zeros_like(cnt_full)
    for r in range(3):
        sh = pa.Simulation(pa.make_params(seed=1, n_gen=4, max_distances=1000, shard_rank=r, shard_count=3, **kw))
        sh.run(4)
        parts.append(sh.core_genome.read_matrix())
        cnt += sh.core_genome.pairwise_counts(sh.range1, sh.range2)[0]
        assert np.array_equal(sh.pan_genome.read_matrix(), full.pan_genome.read_matrix())
        sh.close()
    assert np.array_equal(np.concatenate(parts, axis=1), want)
    assert np.array_equal(cnt, cnt_full)
    full.close()


# ----------------------------------------------------------------------------- kernel variants
@pytest.mark.parametrize("N,L,lm,lh", [(1000, 300, 20000.0, 20000.0), (1000, 200, 150.0, 15.0), (100, 700, 300.0, 300.0)])
def test_block_sweep_equals_wave_sweep(pa, orc, N, L, lm, lh):
    # the block-per-row sweep (large populations) and the wave-per-row sweep (with its queue
    # overflowing at the heavy rates) implement the same keyed arithmetic
    rng = np.random.default_rng(N + L)
    m0 = _rand_core(rng, N, L)
    sample = rng.integers(0, N, N).astype(np.uint32)
    plan = orc.core_plan(lm, lh, 1000)
    want = orc.next_generation(m0, sample)
    orc.mutate_core(want, 0, 5, 2, plan)
    orc.recombine_core(want, 0, 5, 2, plan)
    for force, inline in ((0, 0), (1, 0), (1, 1)):
        for bpc, rows in ((1, 2), (8, 3), (6, 4)):
            pop = pa.Population(N, L, 4, True, 0.0, 5, 0, global_cols=1000)
            pop.set_tuning("force_block_sweep", force)
            pop.set_tuning("force_inline_sweep", inline)
            pop.set_tuning("sweep_blocks_per_cu", bpc)
            pop.set_tuning("sweep_rows", rows)
            pop.set_rates([lm], [lh])
            pop.load_matrix(m0)
            pop.step(2, sample, True)
            assert np.array_equal(pop.read_matrix(), want)
            pop.close()


@pytest.mark.parametrize("N,L,tune", [(5000, 40, {"block_waves": 8}), (5000, 40, {"block_waves": 16}),
                                      (3000, 50, {"lds_limit": 12000}), (9000, 30, {"lds_limit": 40000}),
                                      (2500, 64, {"block_waves": 16, "lds_limit": 30000}),
                                      (5000, 40, {"no_block_preload": 1}), (9000, 30, {"no_block_preload": 1, "block_waves": 16}),
                                      (3000, 50, {"block_waves": 4, "lds_limit": 9000}),
                                      (5000, 40, {"block_batch": 2}), (9000, 30, {"block_batch": 2, "block_waves": 8})])
def test_block_sweep_geometries(pa, orc, N, L, tune):
    # workgroup size, rows per iteration and the 4- or 2-segment batches are host choices that
    # must not change the result
    rng = np.random.default_rng(N + L)
    m0 = _rand_core(rng, N, L)
    sample = rng.integers(0, N, N).astype(np.uint32)
    lm, lh = 0.05 * L, 0.02 * L
    plan = orc.core_plan(lm, lh, L)
    want = orc.next_generation(m0, sample)
    orc.mutate_core(want, 0, 9, 3, plan)
    orc.recombine_core(want, 0, 9, 3, plan)
    pop = pa.Population(N, L, 4, True, 0.0, 9, 0)
    for k, v in tune.items():
        pop.set_tuning(k, v)
    pop.set_rates([lm], [lh])
    pop.load_matrix(m0)
    pop.step(3, sample, True)
    assert np.array_equal(pop.read_matrix(), want)
    pop.close()


@pytest.mark.parametrize("N,L,tune", [(1000, 60, {}), (700, 33, {"sweep_rows": 2}), (3000, 40, {}),
                                      (1000, 50, {"force_block_sweep": 1}), (9000, 24, {"block_batch": 2}),
                                      (2048, 30, {"force_inline_sweep": 1})])
def test_sweeps_carry_bytes_above_15(pa, orc, N, L, tune):
    # ps_load_matrix accepts any byte: a matrix with bytes 16 / 200 / 255 must come through gather, mutation and HR
    # like in the oracle (a former build of the sweeps borrowed bits 4-7 of the child byte in LDS).
    rng = np.random.default_rng(N * 13 + L)
    m0 = _rand_core(rng, N, L)
    for v in (16, 17, 128, 200, 255, 0, 3):
        for _ in range(40):
            m0[rng.integers(0, N), rng.integers(0, L)] = v
    sample = rng.integers(0, N, N).astype(np.uint32)
    LG = 1200000
    lm, lh = 0.05 * LG, 0.0025 * LG          # cfg2's per-site rates
    plan = orc.core_plan(lm, lh, LG)
    want = orc.next_generation(m0, sample)
    orc.mutate_core(want, 100, 21, 5, plan)
    orc.recombine_core(want, 100, 21, 5, plan)
    pop = pa.Population(N, L, 4, True, 0.0, 21, 0, col_offset=100, global_cols=LG)
    for k, v in tune.items():
        pop.set_tuning(k, v)
    pop.set_rates([lm], [lh])
    pop.load_matrix(m0)
    pop.step(5, sample, True)
    assert np.array_equal(pop.read_matrix(), want)
    # the three calls in order agree as well
    pop.load_matrix(m0)
    pop.next_generation(sample)
    pop.mutate_alleles(5)
    pop.recombine(5)
    assert np.array_equal(pop.read_matrix(), want)
    pop.close()


@pytest.mark.parametrize("total,krc", [(0.0640, (1, 1, 0)), (0.0650, (1, 2, 1)), (0.0039, (0, 1, 0)), (0.0485, (0, 4, 0)),
                                       (0.1300, (2, 2, 1)), (0.1350, (2, 3, 2))])
@pytest.mark.parametrize("N,offset,ascending", [(1000, 40, False), (1024, 41, False), (777, 42, False), (17, 43, False),
                                                (5000, 40, True), (4097, 43, True), (2100, 42, False)])
def test_candidate_push_on_both_sides_of_the_plan_limits(pa, orc, N, offset, ascending, total, krc):
    # The wave / window / block sweeps have two builds of their candidate push and dense pass (core_kernels.h): plans whose
    # events all sit in symbols with n = 0 (cshift = 0: one candidate word per row pair) and plans that reach n = 1
    # (cshift = 1, cfg3's rates: two more mask words, bit 0 of n in the queue entry); above that (cshift >= 2) the
    # queue-free inline kernel runs.  Per-site event rates on both sides of both limits, a very sparse plan (most lanes
    # without a candidate), a plan without symbol-decided mutations (k = 0: every event through the residual symbols),
    # shards that start at every position inside a 4-site block group (offset mod 4: the first and the last batch of the
    # wave / window sweeps then lie partly outside the shard), ragged last lanes (N = 777, 17, 4097) and all three sweeps
    # (ascending parents + N > 1024 = the window sweep, unsorted = the block sweep).
    LG, L = 1200000, 26
    rng = np.random.default_rng(N * 31 + offset + int(total * 1e4))
    m0 = _rand_core(rng, N, L)
    sample = rng.integers(0, N, N).astype(np.uint32)
    if ascending:
        sample = np.sort(sample)
    lm, lh = (total - 0.002) * LG, 0.002 * LG
    plan = orc.core_plan(lm, lh, LG)
    assert (plan.k, plan.R, plan.cshift) == krc
    want = orc.next_generation(m0, sample)
    orc.mutate_core(want, offset, 77, 11, plan)
    orc.recombine_core(want, offset, 77, 11, plan)
    pop = pa.Population(N, L, 4, True, 0.0, 77, 0, col_offset=offset, global_cols=LG)
    pop.set_rates([lm], [lh])
    pop.load_matrix(m0)
    pop.step(11, sample, True)
    assert np.array_equal(pop.read_matrix(), want)
    pop.close()


@pytest.mark.parametrize("N,L,tune", [(1000, 60, {}), (1000, 45, {"sweep_rows": 2}), (1000, 33, {"sweep_rows": 4}),
                                      (3000, 40, {}), (1000, 50, {"force_block_sweep": 1}), (9000, 24, {"block_batch": 2}),
                                      (9000, 20, {"no_block_preload": 1}), (5000, 40, {"block_waves": 4}),
                                      (20000, 6, {}), (1500, 30, {"sweep_out_of_place": 1, "force_block_sweep": 1})])
@pytest.mark.parametrize("ops", ["step", "calls"])
def test_full_queues_fall_back_to_the_queue_free_redo(pa, orc, N, L, tune, ops):
    # A full candidate queue or HR list must not drop events (round 2: a sticky PS_ERR_STATE with the matrix already
    # modified): the wave redoes its batch / the workgroup its row group by the queue-free method.  `sweep_queue_cap`
    # makes the kernels treat their queues as 8 entries long, so the redo path runs for (nearly) every batch.
    rng = np.random.default_rng(N * 7 + L)
    m0 = _rand_core(rng, N, L)
    sample = rng.integers(0, N, N).astype(np.uint32)
    LG = 1200000
    lm, lh = 0.05 * LG, 0.02 * LG
    plan = orc.core_plan(lm, lh, LG)
    want = orc.next_generation(m0, sample)
    orc.mutate_core(want, 7, 3, 11, plan)
    orc.recombine_core(want, 7, 3, 11, plan)
    pop = pa.Population(N, L, 4, True, 0.0, 3, 0, col_offset=7, global_cols=LG)
    for k, v in tune.items():
        pop.set_tuning(k, v)
    pop.set_tuning("sweep_queue_cap", 8)
    pop.set_rates([lm], [lh])
    pop.load_matrix(m0)
    if ops == "step":
        pop.step(11, sample, True)
    else:
        pop.next_generation(sample)
        pop.mutate_alleles(11)
        pop.recombine(11)
    assert np.array_equal(pop.read_matrix(), want)
    pop.close()


@pytest.mark.parametrize("N,G,cap", [(700, 1000, 1), (2500, 4000, 7), (130, 300, 3)])
def test_full_hgt_bins_take_the_overflow_image(pa, orc, N, G, cap):
    # binned HGT with bins far too small for the events: nothing may be dropped (population.rs:544-751, accessory path)
    rng = np.random.default_rng(N + G)
    a = (rng.random((N, G)) < 0.3).astype(np.uint8)
    cb, ce, lr = [0, G // 2], [G // 2, G], [60.0, 9.0]
    for gen in (4, 5):                      # twice: the overflow image must be clean again for the second launch
        want = a.copy()
        orc.recombine_acc(want, 31, gen, cb, ce, lr)
        acc = pa.Population(N, G, 2, False, 0.3, 31, 0)
        acc.set_tuning("hgt_mode", 2)
        acc.set_tuning("hgt_bin_cap", cap)
        acc.set_rates([0.0, 0.0], lr, cb, ce)
        acc.load_matrix(a)
        acc.recombine(gen)
        mid = acc.read_matrix()
        assert np.array_equal(mid, want)
        want2 = mid.copy()
        orc.recombine_acc(want2, 31, gen + 10, cb, ce, lr)
        acc.recombine(gen + 10)
        assert np.array_equal(acc.read_matrix(), want2)
        acc.close()


@pytest.mark.parametrize("N,G,mode", [(2, 1, 2), (65, 129, 2), (300, 4000, 2), (1000, 700, 0), (130, 64, 1)])
def test_pairwise_accessory_all_pairs_form(pa, orc, N, G, mode):
    # accessory Jaccard numerators (distances.rs:55-77) for more sampled pairs than a quarter of all pairs: all
    # intersections from LDS tiles + lookup, unions from the row counts -- every ordered pair incl. i == j
    rng = np.random.default_rng(N * 5 + G)
    m = _rand_acc(rng, N, G, 0.4)
    m[0, :] = 0                                # an empty row: union = the other row's count, NaN distance with itself at core_genes 0
    ii, jj = np.meshgrid(np.arange(N, dtype=np.uint32), np.arange(N, dtype=np.uint32), indexing="ij")
    r1, r2 = np.ascontiguousarray(ii.ravel()), np.ascontiguousarray(jj.ravel())
    if r1.size > 200000:
        sel = rng.choice(r1.size, 200000, replace=False)
        r1, r2 = np.ascontiguousarray(r1[sel]), np.ascontiguousarray(r2[sel])
    pop = pa.Population(N, G, 2, False, 0.4, 0, 0)
    pop.set_tuning("pair_mode", mode)
    pop.load_matrix(m)
    inter, uni = pop.pairwise_counts(r1, r2)
    a, b = m[r1].astype(bool), m[r2].astype(bool)
    assert np.array_equal(inter, (a & b).sum(1).astype(np.uint32))
    assert np.array_equal(uni, (a | b).sum(1).astype(np.uint32))
    assert np.array_equal(pop.pairwise_distances(r1.size, r1, r2), orc.pairwise_distances(m, False, 0, r1, r2), equal_nan=True)
    pop.close()


@pytest.mark.parametrize("N,L,lh,tune", [(1030, 40, 0.02, {}), (3000, 33, 0.02, {}), (9000, 20, 0.0, {}), (20000, 7, 0.1, {}),
                                         (5000, 30, 0.02, {"sweep_queue_cap": 8}), (4097, 25, 0.02, {"window_sweep": 0}),
                                         (65536, 3, 0.02, {}), (3000, 20, 0.02, {"sweep_out_of_place": 1})])
def test_step_with_ascending_parents_takes_the_window_sweep(pa, orc, N, L, lh, tune):
    # ps_step on a wide population with an ASCENDING sample: the window sweep (a wave gathers its 1024-child segment from a
    # window of the parent row, HR donors recomputed from the old generation), incl. a skewed sample whose windows exceed the
    # row buffer (second launch), a full queue (redo path) and the block sweep for comparison -- always the oracle's result
    rng = np.random.default_rng(N * 3 + L)
    m0 = _rand_core(rng, N, L)
    LG = 1200000
    lm = 0.05 * LG
    lhr = lh * LG
    plan = orc.core_plan(lm, lhr, LG)
    for skew in (False, True):
        if skew:       # most children from the first tenth of the parents, the rest spread thin: wide windows at the end
            sample = np.sort(np.concatenate([rng.integers(0, max(1, N // 10), N - N // 20), rng.integers(0, N, N // 20)])).astype(np.uint32)
        else:
            sample = np.sort(rng.integers(0, N, N)).astype(np.uint32)
        want = orc.next_generation(m0, sample)
        orc.mutate_core(want, 11, 5, 2, plan)
        if lhr > 0:
            orc.recombine_core(want, 11, 5, 2, plan)
        pop = pa.Population(N, L, 4, True, 0.0, 5, 0, col_offset=11, global_cols=LG)
        for k, v in tune.items():
            pop.set_tuning(k, v)
        pop.set_rates([lm], [lhr])
        pop.load_matrix(m0)
        pop.step(2, sample, lhr > 0)
        assert np.array_equal(pop.read_matrix(), want), "skew %s" % skew
        pop.step(3, sample, lhr > 0)           # a second generation on top (the buffers have swapped)
        want2 = orc.next_generation(want, sample)
        orc.mutate_core(want2, 11, 5, 3, plan)
        if lhr > 0:
            orc.recombine_core(want2, 11, 5, 3, plan)
        assert np.array_equal(pop.read_matrix(), want2)
        pop.close()


def test_wide_segments_at_any_generation_number(pa, orc):
    # The window sweep's second ("wide") launch leaves at once unless the first launch of the SAME generation flagged a
    # segment for it: one flag per generation parity, set by the first launch, cleared by the first launch of the generation
    # before (core_kernels.h).  Generation numbers here are neither consecutive nor alternating in parity
    # (7, 12, 14, 15, 15, 40), wide and narrow generations interleave, so a stale or a missing flag would leave the children
    # of a wide segment unwritten -- every state is compared with the oracle's.
    N, L, LG = 6000, 18, 1200000
    rng = np.random.default_rng(77)
    m = _rand_core(rng, N, L)
    lm, lhr = 0.05 * LG, 0.02 * LG
    plan = orc.core_plan(lm, lhr, LG)
    pop = pa.Population(N, L, 4, True, 0.0, 9, 0, col_offset=5, global_cols=LG)
    pop.set_rates([lm], [lhr])
    pop.load_matrix(m)
    narrow = np.sort(rng.integers(0, N, N)).astype(np.uint32)
    # parents far apart inside one 1024-child segment: half of the children from the first 50 parents, half from the last 50
    wide = np.sort(np.concatenate([rng.integers(0, 50, N // 2), rng.integers(N - 50, N, N - N // 2)])).astype(np.uint32)
    for gen, sample in ((7, wide), (12, narrow), (14, wide), (15, wide), (15, narrow), (40, wide)):
        pop.step(gen, sample, True)
        assert pop.last_sweep_form() == 3
        m = orc.next_generation(m, sample)
        orc.mutate_core(m, 5, 9, gen, plan)
        orc.recombine_core(m, 5, 9, gen, plan)
        assert np.array_equal(pop.read_matrix(), m), gen
    pop.close()


# ----------------------------------------------------------------------------- BASELINE full sizes
def _crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8))


def test_config2_full_size_generation_matches_oracle(pa, orc):
    # BASELINE configs[1]: --pop_size 1000 --core_size 1200000 --pan_genes 6000 --seed 0.
    # One whole generation (select, gather, mutate, HR, HGT) bit-exact against the oracle.
    from orc_sim import OracleSim
    kw = dict(pop_size=1000, core_size=1200000, pan_genes=6000, core_genes=2000)
    sim = pa.Simulation(pa.make_params(seed=0, n_gen=2, max_distances=20000, **kw))
    ref = OracleSim(seed=0, **kw)
    sim.run(1)
    sim.sync()
    ref.generation(0)
    assert np.array_equal(sim.last_parents(), ref.last_idx)
    got = sim.core_genome.read_matrix()
    assert _crc(got) == _crc(ref.core) and np.array_equal(got, ref.core)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    core_d, acc_d = sim.final_distances()
    assert np.array_equal(core_d, orc.pairwise_distances(ref.core, True, 2000, sim.range1, sim.range2))
    assert np.array_equal(acc_d, orc.pairwise_distances(ref.acc, False, 2000, sim.range1, sim.range2))
    sim.close()


def test_config2_full_size_properties(pa):
    # size-independent properties at the full benchmark size, over several generations
    kw = dict(pop_size=1000, core_size=1200000, pan_genes=6000, core_genes=2000)
    runs = []
    for _ in range(2):
        sim = pa.Simulation(pa.make_params(seed=0, n_gen=8, max_distances=50000, **kw))
        sim.run(8)
        core_d, acc_d = sim.final_distances()
        (cnt,) = sim.core_genome.pairwise_counts(sim.range1, sim.range2)
        runs.append((sim.last_parents(), core_d, acc_d, cnt, sim.pan_genome.gene_frequencies()))
        if not runs[1:]:
            m = sim.core_genome.read_matrix()
            assert np.isin(m, (1, 2, 4, 8)).all()                     # alleles stay one-hot
            assert (cnt % 2 == 0).all()                                # so every mismatch counts 2 (population.rs:817)
            i, j = int(sim.range1[0]), int(sim.range2[0])
            assert cnt[0] == 2 * int((m[i] != m[j]).sum())
            full_crc = [_crc(m[:, :600000]), _crc(m[:, 600000:])]
        sim.close()
    for a, b in zip(runs[0], runs[1]):                                 # deterministic replay under --seed
        assert np.array_equal(a, b)
    # two site shards reproduce their halves of the unsharded run and their counts add up
    cnt_sum = np.zeros_like(runs[0][3])
    for r in range(2):
        sh = pa.Simulation(pa.make_params(seed=0, n_gen=8, max_distances=50000, shard_rank=r, shard_count=2, **kw))
        sh.run(8)
        sh.sync()
        assert _crc(sh.core_genome.read_matrix()) == full_crc[r]
        cnt_sum += sh.core_genome.pairwise_counts(sh.range1, sh.range2)[0]
        sh.close()
    assert np.array_equal(cnt_sum, runs[0][3])


def test_config3_rates_full_width_rows(pa, orc):
    # BASELINE configs[2]: HR_rate = HGT_rate = 0.5 (donor-copy path stressed), full population,
    # a slice of the genome keyed at its global site offset
    from orc_sim import OracleSim
    kw = dict(pop_size=1000, core_size=1200000, pan_genes=6000, core_genes=2000, HR_rate=0.5, HGT_rate=0.5)
    ref = OracleSim(seed=0, site_begin=700000, site_end=720000, **kw)
    p = pa.make_params(seed=0, n_gen=3, max_distances=1000, shard_rank=35, shard_count=60, **kw)
    sim = pa.Simulation(p)
    assert sim.core_genome.ncols == 20000
    for g in range(2):
        sim.run(1)
        ref.generation(g)
    sim.sync()
    assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    sim.close()


# ----------------------------------------------------------------------------- randomized differential
@pytest.mark.parametrize("trial", range(30))
def test_random_operator_sequences(pa, orc, trial):
    """Random sizes, rates and call sequences: after every call the HBM state must equal the oracle's."""
    rng = np.random.default_rng(1000 + trial)
    N = int(rng.choice([2, 3, 17, 64, 100, 129, 640, 1000, 1024, 1025, 1500, 2100]))
    L = int(rng.integers(1, 400))
    LG = L + int(rng.integers(0, 1000))
    off = int(rng.integers(0, LG - L + 1))
    G = int(rng.integers(1, 300))
    seed = int(rng.integers(0, 2**63))
    lm = float(rng.choice([0.0, 0.3, 5.0, 0.05 * LG, 0.3 * LG]))
    lh = float(rng.choice([0.0, 0.2, 0.05 * LG, 0.2 * LG]))
    g1 = int(rng.integers(0, G + 1))
    comps = [(0, g1), (g1, G)] if 0 < g1 < G else [(0, G)]
    cb, ce = [c[0] for c in comps], [c[1] for c in comps]
    am = [float(rng.choice([0.0, 0.5, 1.0, 1000.0])) * (e - b) for b, e in comps]
    ar = [float(rng.choice([0.0, 1.0, 30.0, 400.0])) for _ in comps]
    core = pa.Population(N, L, 4, True, 0.0, seed, 0, col_offset=off, global_cols=LG)
    acc = pa.Population(N, G, 2, False, 0.3, seed, 7)
    core.set_rates([lm], [lh])
    acc.set_rates(am, ar, cb, ce)
    if rng.random() < 0.5:
        core.set_tuning("sweep_rows", int(rng.integers(2, 5)))
        core.set_tuning("sweep_blocks_per_cu", int(rng.integers(1, 9)))
    if rng.random() < 0.3:
        core.set_tuning("force_block_sweep", 1)
    if rng.random() < 0.3:
        acc.set_tuning("hgt_mode", int(rng.integers(0, 3)))
    plan = orc.core_plan(lm, lh, LG)
    mc = _rand_core(rng, N, L)
    ma = _rand_acc(rng, N, G, 0.3)
    core.load_matrix(mc)
    acc.load_matrix(ma)
    for step in range(8):
        op = rng.choice(["gather", "mutate", "recombine", "step", "step_norec"])
        gen = int(rng.integers(0, 2**31))
        sample = rng.integers(0, N, N).astype(np.uint32)
        if op == "gather":
            core.next_generation(sample)
            acc.next_generation(sample)
            mc, ma = orc.next_generation(mc, sample), orc.next_generation(ma, sample)
        elif op == "mutate":
            core.mutate_alleles(gen)
            acc.mutate_alleles(gen)
            orc.mutate_core(mc, off, seed, gen, plan)
            orc.mutate_acc(ma, seed, gen, cb, ce, am)
        elif op == "recombine":
            core.recombine(gen)
            acc.recombine(gen)
            orc.recombine_core(mc, off, seed, gen, plan)
            orc.recombine_acc(ma, seed, gen, cb, ce, ar)
        else:
            rec = op == "step"
            core.step(gen, sample, rec)
            acc.step(gen, sample, rec)
            mc, ma = orc.next_generation(mc, sample), orc.next_generation(ma, sample)
            orc.mutate_core(mc, off, seed, gen, plan)
            orc.mutate_acc(ma, seed, gen, cb, ce, am)
            if rec:
                orc.recombine_core(mc, off, seed, gen, plan)
                orc.recombine_acc(ma, seed, gen, cb, ce, ar)
        assert np.array_equal(core.read_matrix(), mc), (trial, step, op, N, L, lm, lh)
        assert np.array_equal(acc.read_matrix(), ma), (trial, step, op, N, G, am, ar)
    P = 300
    if N >= 2:
        r1, r2 = orc.sample_pairs(seed % 1000, N, P)
        assert np.array_equal(core.pairwise_counts(r1, r2)[0], orc.pairwise_hamming_counts(mc, 0, L, r1, r2))
        assert np.array_equal(acc.pairwise_distances(P, r1, r2), orc.pairwise_distances(ma, False, 7, r1, r2))
    assert np.array_equal(acc.gene_frequencies(), orc.gene_frequencies(ma, 7))
    core.close()
    acc.close()


# ----------------------------------------------------------------------------- error behaviour
def test_error_codes_replace_reference_panics(pa):
    pop = pa.Population(10, 50, 4, True, 0.0, 0, 0)
    with pytest.raises(pa.PansimError) as e:             # rates are a precondition of the keyed operators
        pop.mutate_alleles(0)
    assert e.value.code == -6
    with pytest.raises(pa.PansimError) as e:             # parent index out of range (ndarray would panic)
        pop.next_generation(np.full(10, 10, np.uint32))
    assert e.value.code == -1
    with pytest.raises(pa.PansimError):                  # pair index out of range
        pop.pairwise_counts([0, 11], [1, 2])
    with pytest.raises(pa.PansimError):                  # unknown tuning key
        pop.set_tuning("no_such_key", 1)
    with pytest.raises(pa.PansimError):                  # Poisson::new(lambda < 0).unwrap() (population.rs:484)
        pop.set_rates([-1.0], [0.0])
    one = pa.Population(1, 20, 4, True, 0.0, 0, 0)
    with pytest.raises(pa.PansimError):                  # Uniform::new(0, 0) panics for pop_size 1 (population.rs:584)
        one.set_rates([1.0], [1.0])
    one.set_rates([5.0], [0.0])
    one.mutate_alleles(0)                                # mutation alone is fine for a single individual
    acc = pa.Population(10, 30, 2, False, 0.5, 0, 0)
    with pytest.raises(pa.PansimError) as e:             # ln(penalty) = NaN -> WeightedIndex::new panics (:440)
        acc.sample_indices(0, 10, np.ones(10), np.zeros(30), False, False, -1.0, 0.0)
    assert e.value.code == -4
    with pytest.raises(pa.PansimError):                  # gene_frequencies is an accessory-matrix method in main()
        pop.gene_frequencies()
    with pytest.raises(pa.PansimError):                  # pop_size 1 cannot draw pairs (main.rs:421 panics)
        pa.Simulation(pa.make_params(pop_size=1, core_size=100, pan_genes=60, core_genes=20))
    # a failed generation leaves an error, not a crash
    sim = pa.Simulation(pa.make_params(pop_size=20, core_size=200, pan_genes=60, core_genes=20, genome_size_penalty=-1.0))
    with pytest.raises(pa.PansimError) as e:
        sim.run(1)
    assert e.value.code == -4
    sim.close()


def test_zero_accessory_genes(pa, orc):
    # pan_genes == core_genes: the accessory matrix has no columns (population.rs:293-296 keeps weights 1.0)
    from orc_sim import OracleSim
    kw = dict(pop_size=50, core_size=700, pan_genes=100, core_genes=100, avg_gene_freq=1.0)
    sim = pa.Simulation(pa.make_params(seed=2, n_gen=3, max_distances=200, **kw))
    ref = OracleSim(seed=2, **kw)
    sim.run(3)
    sim.sync()
    for g in range(3):
        ref.generation(g)
    assert np.array_equal(sim.last_parents(), ref.last_idx)
    assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    core_d, acc_d = sim.final_distances()
    assert np.array_equal(acc_d, orc.pairwise_distances(ref.acc, False, 100, sim.range1, sim.range2))
    assert (acc_d == 0.0).all()
    sim.close()


# ----------------------------------------------------------------------------- long runs
@pytest.mark.parametrize("kw,extra,gens", [
    (dict(pop_size=300, core_size=4000, pan_genes=1200, core_genes=300), dict(), 150),
    (dict(pop_size=200, core_size=2500, pan_genes=900, core_genes=250, HR_rate=0.4, HGT_rate=0.4),
     dict(prop_positive=0.2, competition_strength=20.0), 60),
    (dict(pop_size=1100, core_size=600, pan_genes=500, core_genes=100), dict(genome_size_penalty=0.9), 40),
])
@pytest.mark.parametrize("env", [{}, {"PANSIM_HEAVY_HGT": "1", "PANSIM_HGT_MODE": "2"}])
def test_long_run_stays_bit_exact(pa, orc, monkeypatch, kw, extra, gens, env):
    # many generations without host synchronisation in between (slot ring, event ordering, the
    # counter sets of the sweep, the heavy-HGT turn-taking): every parent draw must still match
    from orc_sim import OracleSim
    for k, v in env.items():       # force the cfg3 schedule (HGT and sweep take turns) and the binned HGT kernels
        monkeypatch.setenv(k, v)
    sim = pa.Simulation(pa.make_params(seed=77, n_gen=gens, max_distances=500, **kw, **extra))
    ref = OracleSim(seed=77, **kw, **extra)
    done = 0
    for chunk in (gens // 3, gens - gens // 3):
        sim.run(chunk)
        sim.sync()
        for g in range(done, done + chunk):
            ref.generation(g)
        done += chunk
        assert np.array_equal(sim.last_parents(), ref.last_idx)
        assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
        assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    sim.close()


def test_hgt_binned_auto_many_partitions(pa, orc):
    # 1.2e7 expected events at N = 3000: the binned form is chosen by itself, with the real LDS
    # limit (10 recipient partitions of 300 rows), and must still equal the oracle bit for bit
    N, G = 3000, 4000
    cb, ce = [0, 3600], [3600, 4000]
    lr = [3600.0, 400.0]
    rng = np.random.default_rng(5)
    m0 = _rand_acc(rng, N, G, 0.3)
    m0[7, :] = 0
    pop = pa.Population(N, G, 2, False, 0.3, 11, 0)
    pop.set_rates([0.0, 0.0], lr, cb, ce)
    pop.load_matrix(m0)
    pop.recombine(4)
    want = m0.copy()
    K = orc.recombine_acc(want, 11, 4, cb, ce, lr)
    assert K > 1.1e7
    assert np.array_equal(pop.read_matrix(), want)
    pop.close()


def test_spec_regression_vectors_on_gpu(pa):
    # the frozen vectors of tests/golden/spec_regression.json, reproduced by the HIP path alone
    import json, os, zlib
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spec_regression.json")))
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).view(np.uint8))
    for s in fx["sims"]:
        sim = pa.Simulation(pa.make_params(seed=s["seed"], n_gen=len(s["generations"]), max_distances=10,
                                           **s["params"], **s["extra"]))
        for g, want in enumerate(s["generations"]):
            sim.run(1)
            sim.sync()
            assert crc(sim.last_parents()) == want["parents_crc"]
            assert crc(sim.core_genome.read_matrix()) == want["core_crc"]
            assert crc(sim.pan_genome.read_matrix()) == want["acc_crc"]
        sim.close()
