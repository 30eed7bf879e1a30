"""A SECOND restatement of the reference's deterministic functions, written from the Rust text alone (not from
oracle/pansim_oracle.c) in numpy / plain Python, and a randomised cross-check of the C oracle against it.

The oracle is parity-unpinned (the reference ships no vectors and cannot be built here: DESIGN.md 5), so a slip made once
would sit in the oracle AND in the library that is tested against it.  This file narrows what "unpinned" can hide: the
same functions read a second time, in another language and another shape (vectorised where the C loops, loops where the C
is vectorised).  It does not change the parity grade: nothing here is reference output.

Restated: distances.rs:22-52 (hamming_bitwise_fast), :55-77 (jaccard_distance_fast); population.rs:114-151
(get_distance), :753-784 (average_distance), :787-837 (pairwise_distances), :840-863 (gene_frequencies), :87-94
(standard_deviation), :450-465 (next_generation), :270-437 (the weights of sample_indices).

ln_sum_exp (logsumexp 0.1, not under /root/reference) is written BOTH ways -- the two-pass max-shift form and the one-pass
streaming form -- and the test states where they agree to the bit and where they do not (test_ln_sum_exp_forms).
"""
import math

import numpy as np
import pytest

MIN_POSITIVE = 2.2250738585072014e-308      # f64::MIN_POSITIVE (population.rs:774-776)


# --------------------------------------------------------------------------- distances.rs
def _popcount_u64(v):
    v = v.copy()
    n = np.zeros(v.shape, np.uint32)
    while v.any():
        n += (v & np.uint64(1)).astype(np.uint32)
        v >>= np.uint64(1)
    return n


def hamming_bitwise_fast(x, y):
    """distances.rs:22-52: count_ones of the XOR of native-endian u64 chunks, then of the remainder bytes; u32"""
    x, y = np.asarray(x, np.uint8), np.asarray(y, np.uint8)
    assert len(x) == len(y)                                         # :23
    whole = len(x) // 8 * 8
    d = 0
    if whole:
        d += int(_popcount_u64(x[:whole].view(np.uint64) ^ y[:whole].view(np.uint64)).sum())      # :27-39
    if len(x) % 8:                                                  # :41-49
        d += sum(bin(int(a) ^ int(b)).count("1") for a, b in zip(x[whole:], y[whole:]))
    return d & 0xFFFFFFFF


def jaccard_distance_fast(x, y):
    """distances.rs:55-77: (count_ones of AND, count_ones of OR) over u64 chunks + remainder bytes"""
    x, y = np.asarray(x, np.uint8), np.asarray(y, np.uint8)
    assert len(x) == len(y)
    whole = len(x) // 8 * 8
    xi, yi = x[:whole].view(np.uint64), y[:whole].view(np.uint64)
    inter = int(_popcount_u64(xi & yi).sum()) if whole else 0
    union = int(_popcount_u64(xi | yi).sum()) if whole else 0
    for a, b in zip(x[whole:], y[whole:]):
        inter += bin(int(a) & int(b)).count("1")
        union += bin(int(a) | int(b)).count("1")
    return inter, union


# --------------------------------------------------------------------------- population.rs
def pair_distance(row1, row2, core, core_genes, ncols, matches=None):
    """the closure of get_distance (population.rs:134-146, with `matches`) / pairwise_distances (:813-830, without)"""
    if core:
        distance = hamming_bitwise_fast(row1, row2) // 2           # u32 integer division FIRST (:135, :817)
        return float(distance) / float(ncols)
    inter, union = jaccard_distance_fast(row1, row2)
    with np.errstate(invalid="ignore", divide="ignore"):
        if matches is None:
            q = np.float64(float(inter) + float(core_genes)) / np.float64(float(union) + float(core_genes))
        else:
            q = np.float64(float(inter) + matches + float(core_genes)) / np.float64(float(union) + matches + float(core_genes))
        return float(np.float64(1.0) - q)


def pairwise_distances(pop, core, core_genes, r1, r2):
    return np.array([pair_distance(pop[i], pop[j], core, core_genes, pop.shape[1]) for i, j in zip(r1, r2)], np.float64)


def average_distance(pop, core, core_genes):
    """population.rs:753-784: per i the fold (0.0, 0) over j != i in ascending j, sum / count, exact 0 -> MIN_POSITIVE"""
    n = pop.shape[0]
    out = np.zeros(n)
    for i in range(n):
        s, c = 0.0, 0
        for j in range(n):
            if j == i:
                continue                                            # :126-128
            s, c = s + pair_distance(pop[i], pop[j], core, core_genes, pop.shape[1], matches=0.0), c + 1
        with np.errstate(invalid="ignore", divide="ignore"):
            f = float(np.float64(s) / np.float64(c))
        out[i] = MIN_POSITIVE if f == 0.0 else f
    return out


def gene_frequencies(pop, core_genes):
    """population.rs:840-863"""
    n = float(pop.shape[0])
    return np.array([float(int(pop[:, g].astype(np.uint64).sum())) / n for g in range(pop.shape[1])] + [1.0] * core_genes)


def standard_deviation(values):
    """population.rs:83-94: sequential sums, population variance; returns (std, mean)"""
    s = 0.0
    for v in values:
        s += v
    mean = s / float(len(values))
    ss = 0.0
    for v in values:
        ss += (v - mean) * (v - mean)                               # powi(2)
    return math.sqrt(ss / float(len(values))), mean


def ln_sum_exp_two_pass(xs):
    """max-shift form: m = max, m + ln(sum exp(x - m))"""
    m = -math.inf
    for x in xs:
        m = max(m, x)
    if m == -math.inf:
        return -math.inf
    s = 0.0
    for x in xs:
        s += math.exp(x - m)
    return m + math.log(s)


def ln_sum_exp_streaming(xs):
    """one-pass form: running maximum m and running sum r of exp(x - m), rescaled when the maximum moves"""
    m, r = -math.inf, 0.0
    for x in xs:
        if x <= m:
            r += math.exp(x - m)
        else:
            r = r * math.exp(m - x) + 1.0 if m != -math.inf else 1.0
            m = x
    return m + math.log(r) if r > 0.0 else -math.inf


def _softmax(xs, lse):
    """`.map(|x| (x - lse).exp())`, sequential sum, `w / sum` (population.rs:328-340, :357-361, :378-382)"""
    e = [math.exp(x - lse) for x in xs]
    s = 0.0
    for v in e:
        s += v
    with np.errstate(invalid="ignore", divide="ignore"):
        return [float(np.float64(v) / np.float64(s)) if v != -math.inf else 0.0 for v in e]


def sample_weights(pop, sel, avg_gene_num, avg_dists, no_control, penalty, competition, lse=ln_sum_exp_streaming):
    """population.rs:282-437: the weight vector handed to WeightedIndex::new"""
    n, G = pop.shape
    num_genes = [int(pop[i].astype(np.int64).sum()) for i in range(n)]                       # :282-291
    w = [1.0] * n                                                                          # :293
    if G > 0:                                                                              # :296
        w = []
        for i in range(n):
            logs = []
            for g in range(G):
                v = 1.0 + sel[g] * float(pop[i, g])
                logs.append(math.log(v) if v > 0.0 else (-math.inf if v == 0.0 else math.nan))      # f64::ln
            if -math.inf in logs:                                                          # :312
                w.append(0.0)
            else:
                s = 0.0
                for v in logs:
                    s += v                                                                 # :317
                w.append(s)
        w = _softmax(w, lse(w))                                                            # :325-340
    if not no_control:                                                                     # :345
        diff = [float(k - avg_gene_num) for k in num_genes]                                # :347-351 (i32 subtraction, then f64)
        lp = math.log(penalty) if penalty > 0.0 else (-math.inf if penalty == 0.0 else math.nan)
        sd = [d * lp for d in diff]                                                        # :355
        sd = _softmax(sd, lse(sd))
        weights = [sd[i] * w[i] for i in range(n)]                                         # :365-369
    else:
        weights = list(w)                                                                  # :371
    cd = [competition * (math.log(a) if a > 0.0 else (-math.inf if a == 0.0 else math.nan)) for a in avg_dists]      # :375
    cd = _softmax(cd, lse(cd))
    weights = [weights[i] * cd[i] for i in range(n)]                                       # :389-393
    mx = -math.inf
    for v in weights:
        mx = max(mx, v) if not math.isnan(v) else mx                                       # f64::max ignores a NaN operand
    if mx == 0.0:                                                                          # :435-437
        weights = [1.0] * n
    return np.array(weights), np.array(num_genes, np.int32)


# --------------------------------------------------------------------------- the cross-checks
def test_distance_kernels_against_the_second_restatement(orc):
    rng = np.random.default_rng(1)
    for n in list(range(0, 20)) + [63, 64, 65, 1000, 1201]:
        for _ in range(3):
            x, y = rng.integers(0, 256, n, dtype=np.uint8), rng.integers(0, 256, n, dtype=np.uint8)
            assert orc.hamming(x, y) == hamming_bitwise_fast(x, y)
            assert tuple(orc.jaccard(x, y)) == jaccard_distance_fast(x, y)
    # the hand-derived vectors of SURVEY 8(c) through this restatement too
    assert hamming_bitwise_fast([1, 2, 4, 8, 1, 2, 4, 8, 1], [1, 2, 4, 8, 1, 2, 4, 8, 2]) == 2
    assert jaccard_distance_fast([1, 1, 0, 0, 1, 0, 0, 0, 1], [1, 0, 1, 0, 1, 0, 0, 0, 0]) == (2, 5)
    assert pair_distance(np.array([1, 1, 0, 0, 1, 0, 0, 0, 1], np.uint8), np.array([1, 0, 1, 0, 1, 0, 0, 0, 0], np.uint8),
                         False, 2000, 9) == 0.0014962593516208988


@pytest.mark.parametrize("core", [True, False])
def test_pairwise_and_average_distances_against_the_second_restatement(orc, core):
    rng = np.random.default_rng(2 + core)
    for n, cols, cg in [(2, 9, 0), (7, 17, 3), (23, 130, 2000), (40, 64, 1)]:
        if core:
            pop = (1 << rng.integers(0, 4, (n, cols))).astype(np.uint8)
        else:
            pop = (rng.random((n, cols)) < 0.3).astype(np.uint8)
            pop[1] = pop[0]                                          # identical rows: distance 0 / the MIN_POSITIVE clamp
        P = 200
        r1 = rng.integers(0, n, P).astype(np.uint32)
        r2 = rng.integers(0, n, P).astype(np.uint32)
        assert np.array_equal(orc.pairwise_distances(pop, core, cg, r1, r2), pairwise_distances(pop, core, cg, r1, r2), equal_nan=True)
        if not core or n <= 23:
            assert np.array_equal(orc.average_distance(pop, core, cg), average_distance(pop, core, cg), equal_nan=True)
    # an empty union with no core genes is 0 / 0 = NaN (population.rs:824-830), and stays NaN
    z = np.zeros((3, 5), np.uint8)
    assert np.isnan(pairwise_distances(z, False, 0, [0], [1])).all() and np.isnan(orc.pairwise_distances(z, False, 0, np.array([0], np.uint32), np.array([1], np.uint32))).all()


def test_frequencies_gather_and_sigma_against_the_second_restatement(orc):
    rng = np.random.default_rng(5)
    pop = (rng.random((37, 50)) < 0.4).astype(np.uint8)
    assert np.array_equal(orc.gene_frequencies(pop, 7), gene_frequencies(pop, 7))
    sample = rng.integers(0, 37, 37).astype(np.uint32)
    assert np.array_equal(orc.next_generation(pop, sample), pop[sample])           # population.rs:450-465
    vals = rng.random(101)
    from oracle import oracle as o
    if hasattr(o, "standard_deviation"):
        assert tuple(o.standard_deviation(vals)) == standard_deviation(list(vals))


def test_ln_sum_exp_forms():
    # equal inputs (neutral selection, no competition: two of the three softmaxes at the defaults): both forms are
    # exactly x + ln(n).  Distinct inputs: they agree to a few ulp, not always to the bit -- the streaming form rounds the
    # running sum at every rescale.  The oracle and the library use the streaming form; the difference is at
    # most the last bits of lse, i.e. a relative 1e-16 in the weights.
    for n in (1, 2, 1000, 65536):
        for x in (0.0, -3.25, 17.0):
            assert ln_sum_exp_two_pass([x] * n) == ln_sum_exp_streaming([x] * n) == x + math.log(float(n))
    rng = np.random.default_rng(9)
    worst = 0.0
    differ = 0
    for _ in range(300):
        xs = list(rng.normal(0.0, 3.0, int(rng.integers(2, 200))))
        a, b = ln_sum_exp_two_pass(xs), ln_sum_exp_streaming(xs)
        differ += a != b
        worst = max(worst, abs(a - b) / max(abs(a), 1e-300))
    assert worst < 1e-14            # a few ulp at most
    assert differ > 0               # ... but NOT bit-identical in general: which form the crate uses matters at the last bit
    assert ln_sum_exp_two_pass([-math.inf, -math.inf]) == ln_sum_exp_streaming([-math.inf, -math.inf]) == -math.inf


@pytest.mark.parametrize("case", range(8))
def test_sample_weights_against_the_second_restatement(orc, case):
    rng = np.random.default_rng(40 + case)
    n, G = int(rng.integers(2, 40)), int(rng.integers(1, 60))
    pop = (rng.random((n, G)) < 0.35).astype(np.uint8)
    sel = np.zeros(G) if case % 4 == 0 else np.where(rng.random(G) < 0.3, rng.exponential(0.1, G), -np.minimum(rng.exponential(0.1, G), 1.0))
    if case % 4 == 1:
        sel[int(rng.integers(0, G))] = -1.0                                       # ln(0) = -inf: the reset rule (:312-318)
    avg = np.ones(n) if case % 2 == 0 else rng.random(n) * 0.3 + 1e-3
    comp = 0.0 if case % 2 == 0 else 5.0
    no_control = case % 3 == 2
    penalty = 0.99
    agn = int(G * 0.35)
    num, logw = orc.fitness_terms(pop, sel)
    rc, got = orc.sample_weights(num, logw, G, agn, avg, no_control, penalty, comp)
    assert rc == 0
    want, want_num = sample_weights(pop, sel, agn, list(avg), no_control, penalty, comp)
    assert np.array_equal(num, want_num)
    assert np.array_equal(got, want), (got, want)
    # the max-shift form of ln_sum_exp gives the same weights to ~1e-15 (bit-identical for the neutral vectors)
    alt, _ = sample_weights(pop, sel, agn, list(avg), no_control, penalty, comp, lse=ln_sum_exp_two_pass)
    assert np.allclose(alt, want, rtol=1e-13, atol=0.0)
    if case % 4 == 0 and comp == 0.0:
        assert np.array_equal(alt, want)


# --------------------------------------------------------------------------- the authors' scripted competition strengths
AUTHORS_STRENGTHS = (0.0, 100.0, 1e4, 1e8)      # scripts/run_pansim_benchmark.sh:235-278


def _authors_like_population(rng, n, G, clones=0):
    """an accessory matrix like the one the authors' flags produce after a generation (one compartment, every gene flipped
    with p = 0.43 from a clonal start), optionally with `clones` exact copies of row 0"""
    base = (rng.random(G) < 0.21).astype(np.uint8)
    pop = np.where(rng.random((n, G)) < 0.4323, 1 - base, base).astype(np.uint8)
    for k in range(1, clones + 1):
        pop[k] = pop[0]
    return pop


@pytest.mark.parametrize("strength", AUTHORS_STRENGTHS)
def test_authors_competition_strengths_against_the_second_restatement(orc, strength):
    # population.rs:374-393 at the strengths the authors script: at 1e4 the softmax of strength * ln(avg) leaves one
    # individual (or a tie) with all the weight, at 1e8 every other entry underflows to exactly 0
    rng = np.random.default_rng(77)
    n, G, cg = 48, 320, 140
    pop = _authors_like_population(rng, n, G)
    avg = orc.average_distance(pop, False, cg)
    assert np.array_equal(avg, average_distance(pop, False, cg))
    sel = np.zeros(G)                                                # --prop_positive -0.1: neutral genes (main.rs:287-292)
    agn = int(round(0.2086 * G))
    num, logw = orc.fitness_terms(pop, sel)
    rc, got = orc.sample_weights(num, logw, G, agn, avg, False, 0.99, strength)
    assert rc == 0
    want, _ = sample_weights(pop, sel, agn, list(avg), False, 0.99, strength)
    assert np.array_equal(got, want)
    nz = int((got > 0.0).sum())
    if strength >= 1e8:
        assert nz == 1 and int(np.argmax(got)) == int(np.argmax(avg))         # winner takes all: the most distant individual
    elif strength >= 1e4:
        assert got.max() / got.sum() > 0.5 and np.sort(got)[n // 2] < 1e-12 * got.max()        # a handful of individuals hold the mass
    else:
        assert nz == n
    # WeightedIndex over that vector (population.rs:440-443): the oracle's draws only ever name individuals with weight
    rc, idx = orc.draw_parents(got, 5, 3)
    assert rc == 0 and (got[idx] > 0.0).all()


@pytest.mark.parametrize("strength", AUTHORS_STRENGTHS)
def test_clonal_population_under_competition_is_uniform(orc, strength):
    # all rows identical (the clonal start, or the generation after a single-parent sweep without a mutation): every
    # distance is exactly 0, average_distance returns MIN_POSITIVE for everyone (population.rs:774-776), strength * ln of
    # it is the same finite number for all (-7.08e10 at 1e8) and the competition softmax is exactly uniform
    rng = np.random.default_rng(3)
    n, G, cg = 33, 90, 10
    pop = np.tile((rng.random(G) < 0.3).astype(np.uint8), (n, 1))
    avg = orc.average_distance(pop, False, cg)
    assert (avg == MIN_POSITIVE).all() and np.array_equal(avg, average_distance(pop, False, cg))
    num, logw = orc.fitness_terms(pop, np.zeros(G))
    rc, got = orc.sample_weights(num, logw, G, int(pop[0].sum()), avg, False, 0.99, strength)
    want, _ = sample_weights(pop, np.zeros(G), int(pop[0].sum()), list(avg), False, 0.99, strength)
    assert rc == 0 and np.array_equal(got, want)
    assert (got == got[0]).all() and got[0] > 0.0


def test_all_zero_weights_fall_back_to_uniform(orc):
    # population.rs:403, :435-437: when every product underflows to 0 the weights become 1.0.  Reached when the one
    # individual the saturated competition softmax keeps is one the genome-size softmax has already zeroed
    # (--genome_size_penalty 1e-300: two genes more than the others costs exp(-1381))
    G, cg = 40, 5
    pop = np.zeros((6, G), np.uint8)
    pop[:, :10] = 1
    pop[5, 10:14] = 1            # the most distant individual AND the one with four extra genes
    avg = orc.average_distance(pop, False, cg)
    assert int(np.argmax(avg)) == 5
    num, logw = orc.fitness_terms(pop, np.zeros(G))
    rc, got = orc.sample_weights(num, logw, G, 10, avg, False, 1e-300, 1e8)
    want, _ = sample_weights(pop, np.zeros(G), 10, list(avg), False, 1e-300, 1e8)
    assert rc == 0 and np.array_equal(got, want) and (got == 1.0).all()
    # one step away from the fallback: without the genome-size term the winner keeps its weight
    rc, got = orc.sample_weights(num, logw, G, 10, avg, True, 1e-300, 1e8)
    want, _ = sample_weights(pop, np.zeros(G), 10, list(avg), True, 1e-300, 1e8)
    assert rc == 0 and np.array_equal(got, want) and (got > 0.0).sum() == 1 and got[5] > 0.0
