"""GPU parity at the populations of BASELINE configs[3] (--pop_size 65536) and configs[4]
(--pop_size 8192 --max_distances 33554432 --print_matrices): whole generations through ps_sim_run,
the distance phase in its large-N forms and the CLI, each against the CPU oracle.  The core genome is
a site slice keyed at its global offset (a shard of the 1.2 M sites), so the oracle finishes in
seconds; the accessory matrix, the parents and the pair list are full size.
Reference paths: population.rs:270-465 (selection, gather), :486-751 (gain/loss, HR, HGT),
:753-837 (distances), main.rs:413-427, :467-499, :550-553."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config4_population_generations_match_oracle(pa, orc):
    # BASELINE configs[3]: --pop_size 65536 --core_size 1200000 --pan_genes 6000 (defaults otherwise).
    # Shard 12500 of 25000 = sites [600000, 600048); 2 generations: 2e8 HGT events each through the binned
    # HGT kernels (202 recipient partitions), acc_step at W = 1024 words, the block sweep with the LDS full,
    # the turn-taking schedule of the loop; then the distance phase of 100 000 pairs at N = 65536.
    from orc_sim import OracleSim
    kw = dict(pop_size=65536, core_size=1200000, pan_genes=6000, core_genes=2000)
    sim = pa.Simulation(pa.make_params(seed=0, n_gen=2, max_distances=100000, shard_rank=12500, shard_count=25000, **kw))
    assert sim.core_genome.ncols == 48
    ref = OracleSim(seed=0, site_begin=600000, site_end=600048, **kw)
    for g in range(2):
        sim.run(1)
        sim.sync()
        ref.generation(g)
        assert np.array_equal(sim.last_parents(), ref.last_idx), "parents differ at generation %d" % g
    assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    r1, r2 = orc.sample_pairs(0, 65536, 100000)
    assert np.array_equal(sim.range1, r1) and np.array_equal(sim.range2, r2)
    (cnt,) = sim.core_genome.pairwise_counts(r1, r2)
    assert np.array_equal(cnt, orc.pairwise_hamming_counts(ref.core, 0, 48, r1, r2))
    acc_d = sim.pan_genome.pairwise_distances(r1.size, r1, r2)
    assert np.array_equal(acc_d, orc.pairwise_distances(ref.acc, False, 2000, r1, r2))
    assert np.array_equal(sim.pan_genome.gene_frequencies(), orc.gene_frequencies(ref.acc, 2000))
    sim.close()


@pytest.mark.parametrize("mode", [0, 4])
def test_large_population_sampled_pairs(pa, orc, mode):
    # N = 65536 with enough sites for several tiles of the transposed form (and a ragged tail): the
    # sampled-pair kernel that serves populations too wide for an LDS tile (population.rs:787-837)
    N, L, P = 65536, 1100, 30000
    rng = np.random.default_rng(65536)
    base = (1 << rng.integers(0, 4, L)).astype(np.uint8)
    m = np.tile(base, (N, 1))
    mut = rng.random((N, L)) < 0.05
    m[mut] = (1 << rng.integers(0, 4, int(mut.sum()))).astype(np.uint8)
    r1, r2 = orc.sample_pairs(7, N, P)
    r2[:5] = r1[:5]                                   # a pair of an individual with itself counts 0
    want = orc.pairwise_hamming_counts(m, 0, L, r1, r2)
    pop = pa.Population(N, L, 4, True, 0.0, 0, 0)
    pop.set_tuning("pair_mode", mode)
    pop.load_matrix(m)
    (cnt,) = pop.pairwise_counts(r1, r2)
    assert np.array_equal(cnt, want)
    assert (cnt[:5] == 0).all() and cnt.max() > 0
    pop.close()


def test_config5_population_all_pairs_distances(pa, orc):
    # BASELINE configs[4]: --pop_size 8192 --max_distances 2^22 (the run itself uses 2^25; the kernels
    # are the same: all-pairs tiles + lookup, chosen by cost), 2000-site shard at its global offset,
    # 2 generations, then core counts and accessory distances of all 4.2 M sampled pairs
    from orc_sim import OracleSim
    kw = dict(pop_size=8192, core_size=1200000, pan_genes=6000, core_genes=2000)
    P = 1 << 22
    sim = pa.Simulation(pa.make_params(seed=0, n_gen=2, max_distances=P, shard_rank=300, shard_count=600, **kw))
    assert sim.core_genome.ncols == 2000
    ref = OracleSim(seed=0, site_begin=600000, site_end=602000, **kw)
    sim.run(2)
    sim.sync()
    for g in range(2):
        ref.generation(g)
    assert np.array_equal(sim.last_parents(), ref.last_idx)
    assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    r1, r2 = orc.sample_pairs(0, 8192, P)
    assert np.array_equal(sim.range1, r1) and np.array_equal(sim.range2, r2)
    (cnt,) = sim.core_genome.pairwise_counts(r1, r2)
    assert np.array_equal(cnt, orc.pairwise_hamming_counts(ref.core, 0, 2000, r1, r2))
    acc_d = sim.pan_genome.pairwise_distances(P, r1, r2)
    assert np.array_equal(acc_d, orc.pairwise_distances(ref.acc, False, 2000, r1, r2))
    sim.close()


def test_config5_cli_print_matrices(pa, orc, tmp_path):
    # `pansim --pop_size 8192 --max_distances <2^20> --print_matrices` on a short genome: the six output
    # files byte for byte (main.rs:321-331, :467-499, :531-553; population.rs:865-897)
    import ctypes as C
    from orc_sim import OracleSim
    exe = os.path.join(ROOT, "pansim_amd", "pansim")
    N, L, P, cg = 8192, 300, 1 << 20, 50
    kw = dict(pop_size=N, core_size=L, pan_genes=250, core_genes=cg)
    args = []
    for k, v in kw.items():
        args += ["--" + k, str(v)]
    r = subprocess.run([exe, *args, "--n_gen", "2", "--seed", "4", "--max_distances", str(P), "--outpref",
                        str(tmp_path / "run"), "--print_matrices", "--print_dist", "--print_selection"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    ref = OracleSim(seed=4, **kw)
    r1, r2 = orc.sample_pairs(4, N, P)
    per_gen = []
    for g in range(2):
        ref.generation(g)
        cd = orc.pairwise_distances(ref.core, True, cg, r1, r2)
        ad = orc.pairwise_distances(ref.acc, False, cg, r1, r2)
        row = []
        for d in (cd, ad):
            s, m = C.c_double(), C.c_double()
            orc.lib().orc_standard_deviation(d, d.size, C.byref(s), C.byref(m))
            row += [m.value, s.value]
        per_gen.append(row)
    want = "".join("%s\t%s\n" % (orc.fmt_f64(c), orc.fmt_f64(a)) for c, a in zip(cd, ad))
    assert (tmp_path / "run.tsv").read_text() == want
    want = "".join("%s\n" % orc.fmt_f64(x) for x in orc.gene_frequencies(ref.acc, cg))
    assert (tmp_path / "run_freqs.txt").read_text() == want
    want = "".join("\t".join(orc.fmt_f64(x) for x in row) + "\n" for row in per_gen)
    assert (tmp_path / "run_per_gen.tsv").read_text() == want
    want = "\n".join(orc.fmt_f64(x) for x in ref.sel) + "\n"
    assert (tmp_path / "run_selection.tsv").read_text() == want
    orc.lib().orc_write_matrix(ref.core, N, L, 1, cg, str(tmp_path / "want").encode())
    orc.lib().orc_write_matrix(ref.acc, N, 200, 0, cg, str(tmp_path / "want").encode())
    for suffix in ("_core_genome.csv", "_pangenome.csv"):
        assert (tmp_path / ("run" + suffix)).read_bytes() == (tmp_path / ("want" + suffix)).read_bytes()


@pytest.mark.parametrize("N,G,cg", [(1000, 4000, 2000), (2100, 130, 7), (4100, 70, 3), (8200, 130, 7), (9000, 64, 0)])
def test_average_distance_large_populations(pa, orc, N, G, cg):
    # D-avg (population.rs:753-784): the N x N matrix form at the cfg2 population (32 x 32 tiles), at populations
    # with enough 64 x 64 tiles, and the streaming kernel that serves N > 8192, all against the oracle's left-to-right fold
    rng = np.random.default_rng(N + G)
    m = (rng.random((N, G)) < 0.3).astype(np.uint8)
    m[5] = m[6]                                       # a zero distance inside the fold
    pop = pa.Population(N, G, 2, False, 0.3, 0, cg)
    pop.load_matrix(m)
    assert np.array_equal(pop.average_distance(), orc.average_distance(m, False, cg))
    pop.close()


@pytest.mark.parametrize("N,G,cg,nb", [(2, 9, 0, 0), (33, 65, 0, 1), (130, 257, 3, 2), (1000, 4000, 2000, 1), (2100, 130, 7, 2),
                                       (4100, 4000, 2000, 0), (9000, 64, 0, 0), (20000, 300, 11, 0),
                                       (300, 5000, 17, 1), (700, 9000, 5, 2), (1000, 4000, 2000, 4), (333, 700, 1, 4),
                                       (390, 300, 3, 4), (130, 900, 2, 4), (600, 400, 9, 4)])
def test_average_distance_on_the_matrix_cores(pa, orc, N, G, cg, nb):
    # D-avg with the intersections as a {0, 1} X X^T on the FP4 matrix cores and the ordered f64 fold in the accumulator
    # layout (acc_average_distance_mfma_kernel; population.rs:753-784, :114-151): ragged populations and gene counts (fewer
    # and more 256-gene chunks than the sixteen epilogue groups that ride behind them), both
    # fragment counts per wave, a zero distance and an empty row inside the fold, cg = 0 (an empty union is NaN and must
    # stay out of the fold only where j == i), and row shards against slices of the whole; nb = 4 (a wave of phase 1 stores
    # 128 whole rows) with shards of 130 / 43 / 200 rows, i.e. bands that 64-row rounding would leave short
    rng = np.random.default_rng(N * 7 + G)
    m = (rng.random((N, G)) < 0.3).astype(np.uint8)
    if N > 6:
        m[5] = m[6]                                   # a zero distance inside the fold
        m[3] = 0                                      # an individual without genes
    want = orc.average_distance(m, False, cg)
    pop = pa.Population(N, G, 2, False, 0.3, 0, cg)
    pop.load_matrix(m)
    K = 3 if N >= 3 else 2
    # form 2: contraction + fold in one kernel (round 4); form 3: two phases (round 5: contraction on every SIMD -> u16
    # counts, then division + ordered fold, acc_intersections_mfma_kernel / acc_average_from_counts_kernel); 0 = the
    # library's choice for row shards and wide populations (= 3)
    for form in (2, 3, 0):
        pop.set_tuning("davg_form", form)
        pop.set_tuning("davg_nb", nb)
        pop.set_tuning("davg_ib", 32 if form == 3 else 0)       # phase 2 with 32 individuals per workgroup, then its own choice (16 here)
        if form:
            assert np.array_equal(pop.average_distance(), want, equal_nan=True), form
        got = np.concatenate([pop.average_distance_rows(N * r // K, N * (r + 1) // K - N * r // K) for r in range(K)])
        assert np.array_equal(got, want, equal_nan=True), form
        # the lean quotient (rcp + Newton + residual: the arithmetic core of the IEEE division) against the compiler's
        # own f64 division inside the same kernel
        pop.set_tuning("davg_plain_division", 1)
        if form:
            assert np.array_equal(pop.average_distance(), want, equal_nan=True), form
        assert np.array_equal(pop.average_distance_rows(1, N - 1), want[1:], equal_nan=True), form
        pop.set_tuning("davg_plain_division", 0)
    pop.set_tuning("davg_form", 3)
    if N == 20000:
        pop.set_tuning("davg_form", 1)                # the LDS-tile popcount kernel agrees
        assert np.array_equal(pop.average_distance(), want, equal_nan=True)
    # rows outside the population are refused (PS_ERR_INVALID), like every bad argument of the boundary
    for first, count in ((N, 1), (0, N + 1), (N - 1, 2), (0, 0)):
        with pytest.raises(pa.PansimError) as e:
            pop.average_distance_rows(first, count)
        assert e.value.code == -1
    pop.close()


def test_randomised_stress_subset(pa, orc):
    # a time-boxed subset of scripts/stress_parity.py (the long run): random geometries of the distance,
    # HGT and sweep kernels and short generation loops, every result compared with the oracle
    from stress_trials import run
    lines = []
    bad = run(40, seed=20261003, log=lambda *a: lines.append(" ".join(map(str, a))))
    assert bad == 0, "\n".join(lines)


@pytest.mark.parametrize("device_draw", ["0", "1"])
def test_parent_draw_on_host_and_device_agree(pa, orc, monkeypatch, device_draw):
    # P-draw (population.rs:440-443): the N weighted draws run on the device for pop_size >= 4096 and on the host
    # below; PANSIM_DEVICE_DRAW forces either.  Selection and competition on, so that the weights are not uniform.
    from orc_sim import OracleSim
    monkeypatch.setenv("PANSIM_DEVICE_DRAW", device_draw)
    kw = dict(pop_size=4100 if device_draw == "0" else 333, core_size=700, pan_genes=500, core_genes=100)
    extra = dict(prop_positive=0.3, competition_strength=5.0)
    sim = pa.Simulation(pa.make_params(seed=13, n_gen=4, max_distances=200, **kw, **extra))
    ref = OracleSim(seed=13, **kw, **extra)
    for g in range(4):
        sim.run(1)
        ref.generation(g)
        assert np.array_equal(sim.last_parents(), ref.last_idx), "generation %d" % g
    assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    sim.close()


def _max_window(idx):
    """widest parent window of a 1024-child segment (core_sweep_window_kernel: > 1536 bytes goes to the wide launch)"""
    n = len(idx)
    return max(int(idx[min(c + 1023, n - 1)]) - (int(idx[c]) & ~15) + 1 for c in range(0, n, 1024))


@pytest.mark.parametrize("kw,extra,window", [
    (dict(pop_size=5000, core_size=240, pan_genes=300, core_genes=40, HR_rate=0.3, HGT_rate=0.05), dict(prop_positive=0.5, pos_lambda=0.4), 1),
    (dict(pop_size=5000, core_size=240, pan_genes=300, core_genes=40, HR_rate=0.3, HGT_rate=0.05), dict(prop_positive=0.5, pos_lambda=0.4), 0),
    (dict(pop_size=3100, core_size=500, pan_genes=200, core_genes=100, HR_rate=0.6, HGT_rate=0.0), dict(), 1),
    (dict(pop_size=9000, core_size=100, pan_genes=120, core_genes=20, HR_rate=0.0, HGT_rate=0.1), dict(prop_positive=0.3, pos_lambda=0.3), 1),
])
def test_window_sweep_of_wide_populations(pa, orc, kw, extra, window):
    # The generation loop stores the children in ascending parent order, and populations wider than one wavefront take
    # the window sweep (a wave gathers from a ~1.1 KB window of the parent row; HR donors recomputed from the old
    # generation and patched in): with and without selection strong enough that some windows exceed the row buffer (the
    # second, "wide" launch), with HR off, and against the block sweep (window_sweep = 0) -- always the oracle's run
    from orc_sim import OracleSim
    os.environ["PANSIM_WINDOW_SWEEP"] = str(window)
    try:
        sim = pa.Simulation(pa.make_params(seed=13, n_gen=5, max_distances=300, **kw, **extra))
    finally:
        os.environ.pop("PANSIM_WINDOW_SWEEP", None)
    ref = OracleSim(seed=13, **kw, **extra)
    widest = 0
    for g in range(5):
        sim.run(1)
        sim.sync()
        ref.generation(g)
        # (draw order at the boundary; inside, the children sit in ascending parent order: what the window sweep works from)
        assert np.array_equal(sim.last_parents(), ref.last_idx)
        assert (np.diff(ref.internal_idx.astype(np.int64)) >= 0).all()
        widest = max(widest, _max_window(ref.internal_idx))
        assert np.array_equal(sim.core_genome.read_matrix(), ref.core), "generation %d" % g
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    if extra:
        assert widest > 1536, "the selection of this test no longer produces a wide window (%d)" % widest
    core_d, acc_d = sim.final_distances()
    assert np.array_equal(core_d, orc.pairwise_distances(ref.core, True, kw["core_genes"], sim.range1, sim.range2))
    sim.close()


# ----------------------------------------------------------------------------- full-size properties of configs[2..4]
# Siblings of test_gpu_parity.py::test_config2_full_size_properties: the whole 1.2 M-site genome at each BASELINE
# population, checked through size-independent properties with device-side reductions only (no 78 GB read-back):
# one-hot alleles <=> every pair's xor-popcount is even (population.rs:817), deterministic replay under --seed, 8 site
# shards' numerators summing to the unsharded ones (the shards are keyed at their global offsets, so this compares the
# full-length launch -- row ranges per XCD group, chunk counters, second buffer -- with eight short ones), HGT gain-only.
def _shard_counts_sum(pa, kw, gens, P, n_shards, seed=0):
    total = None
    for r in range(n_shards):
        sh = pa.Simulation(pa.make_params(seed=seed, n_gen=gens, max_distances=P, shard_rank=r, shard_count=n_shards, **kw))
        sh.run(gens)
        (c,) = sh.core_genome.pairwise_counts(sh.range1, sh.range2)
        total = c.astype(np.uint64) if total is None else total + c
        sh.close()
    return total


def test_config3_full_size_properties(pa):
    # BASELINE configs[2]: HR_rate = HGT_rate = 0.5 at the full 1.2 M sites (cshift = 1: the wave sweep's second build;
    # the binned HGT taking turns with the sweep)
    kw = dict(pop_size=1000, core_size=1200000, pan_genes=6000, core_genes=2000, HR_rate=0.5, HGT_rate=0.5)
    P, gens = 50000, 4
    runs = []
    for _ in range(2):
        sim = pa.Simulation(pa.make_params(seed=0, n_gen=gens, max_distances=P, **kw))
        sim.run(gens)
        (cnt,) = sim.core_genome.pairwise_counts(sim.range1, sim.range2)
        runs.append((sim.last_parents(), cnt, sim.pan_genome.gene_frequencies(), sim.pan_genome.read_matrix()))
        if len(runs) == 1:
            assert sim.core_genome.last_sweep_form() == 1                # PS_SWEEP_FORM_WAVE
            assert (cnt % 2 == 0).all() and cnt.max() > 0                # alleles stay one-hot
            # HGT never clears a gene (population.rs:632): one more recombination on the same state only adds bits
            before = runs[0][3]
            sim.pan_genome.recombine(gens)
            after = sim.pan_genome.read_matrix()
            assert (after >= before).all() and after.sum() > before.sum()
        sim.close()
    for a, b in zip(runs[0], runs[1]):                                   # deterministic replay under --seed
        assert np.array_equal(a, b)
    assert np.array_equal(_shard_counts_sum(pa, kw, gens, P, 8), runs[0][1].astype(np.uint64))


def test_config4_full_size_properties(pa):
    # BASELINE configs[3]: --pop_size 65536 with all 1.2 M sites on one GPU (2 x 78.6 GB: the window sweep is out of
    # place), two generations; nothing of the core matrix comes back to the host
    import torch
    free, _total = torch.cuda.mem_get_info()
    if free < 170e9:
        # LOUD: a skipped full-size cfg4 test must not read as coverage -- the reason goes to
        # stderr (pytest -rs / -s show it), into gpurun_out/ for the builder, and PANSIM_REQUIRE_FULL_SIZE=1 turns it into a failure
        msg = "test_config4_full_size_properties SKIPPED: needs 2 x 78.6 GB of HBM, only %.1f GB free" % (free / 1e9)
        import sys
        print("\n*** " + msg + " ***", file=sys.stderr, flush=True)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "SKIPPED_full_size_cfg4.txt"), "w").write(msg + "\n")
        except OSError:
            pass
        assert not os.environ.get("PANSIM_REQUIRE_FULL_SIZE"), msg
        pytest.skip(msg)
    kw = dict(pop_size=65536, core_size=1200000, pan_genes=6000, core_genes=2000)
    P, gens = 100000, 2
    runs = []
    for _ in range(2):
        sim = pa.Simulation(pa.make_params(seed=0, n_gen=gens, max_distances=P, **kw))
        sim.run(gens)
        (cnt,) = sim.core_genome.pairwise_counts(sim.range1, sim.range2)
        runs.append((sim.last_parents(), cnt, sim.pan_genome.gene_frequencies()))
        if len(runs) == 1:
            assert sim.core_genome.last_sweep_form() == 3                # PS_SWEEP_FORM_WINDOW
            assert sim.core_genome.last_pair_form() == 4                 # transposed strings, two streamed per pair
            assert (cnt % 2 == 0).all() and cnt.max() > 0
            par = runs[0][0]
            assert (np.diff(par.astype(np.int64)) < 0).any()             # draw order at the boundary (population.rs:443), not the engine's
        sim.close()
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)
    # eight site shards of 150 000 sites (what the 8 ranks of the north-star run hold), one after the other
    assert np.array_equal(_shard_counts_sum(pa, kw, gens, P, 8), runs[0][1].astype(np.uint64))


def test_config5_full_size_distance_forms_agree(pa):
    # BASELINE configs[4]: --pop_size 8192 --max_distances 33554432 at the full 1.2 M sites: the 2^25-pair phase through
    # the block-scaled FP4 matrix-core form (the default), its signed variant, the i8 matrix-core form and the xor + popcount tiles must give
    # the same integers; two generations first so that the population is not clonal
    kw = dict(pop_size=8192, core_size=1200000, pan_genes=6000, core_genes=2000)
    P, gens = 1 << 25, 2
    sim = pa.Simulation(pa.make_params(seed=0, n_gen=gens, max_distances=P, **kw))
    sim.run(gens)
    sim.sync()
    assert sim.core_genome.last_sweep_form() == 3
    got = {}
    for mode, form in ((0, 7), (7, 8), (6, 6), (5, 2)):
        sim.core_genome.set_tuning("pair_mode", mode)
        (cnt,) = sim.core_genome.pairwise_counts(sim.range1, sim.range2)
        assert sim.core_genome.last_pair_form() == form
        got[mode] = cnt
    assert (got[0] % 2 == 0).all() and got[0].max() > 0
    assert np.array_equal(got[0], got[6]) and np.array_equal(got[0], got[5]) and np.array_equal(got[0], got[7])
    # a pair of an individual with itself would count 0; the pair list never holds one (main.rs:413-427)
    assert (sim.range1 != sim.range2).all()
    # replay: the same generations again give the same numerators and gene frequencies
    freqs = sim.pan_genome.gene_frequencies()
    parents = sim.last_parents()
    sim.close()
    sim = pa.Simulation(pa.make_params(seed=0, n_gen=gens, max_distances=P, **kw))
    sim.run(gens)
    (cnt,) = sim.core_genome.pairwise_counts(sim.range1, sim.range2)
    assert np.array_equal(cnt, got[0]) and np.array_equal(sim.pan_genome.gene_frequencies(), freqs)
    assert np.array_equal(sim.last_parents(), parents)
    sim.close()
    assert np.array_equal(_shard_counts_sum(pa, kw, gens, 1 << 20, 8), got[0][:1 << 20].astype(np.uint64))


@pytest.mark.parametrize("hgt,env", [(0.0, {}), (0.4, {"PANSIM_HEAVY_HGT": "1", "PANSIM_HGT_MODE": "2"})])
def test_average_distance_ahead_of_the_sweep(pa, orc, monkeypatch, hgt, env):
    # --competition_strength with a matrix-core D-avg (forced here with davg_form 3; the library's own choice for N > 8192 and
    # for row shards): D-avg of generation g + 1 is computed right behind HGT(g) and AHEAD of sweep(g).  Same parents and
    # matrices as the oracle whether the generations run in one call or one by one; and a matrix loaded between two calls
    # must not be served the distances of the old one (the result carries the matrix's edit epoch)
    from orc_sim import OracleSim
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    kw = dict(pop_size=700, core_size=900, pan_genes=420, core_genes=120, HR_rate=0.1, HGT_rate=hgt)
    extra = dict(competition_strength=6.0, prop_positive=0.2)
    gens = 7
    ref = OracleSim(seed=12, **kw, **extra)
    a = pa.Simulation(pa.make_params(seed=12, n_gen=gens, max_distances=100, **kw, **extra))
    b = pa.Simulation(pa.make_params(seed=12, n_gen=gens, max_distances=100, **kw, **extra))
    for s in (a, b):
        s.pan_genome.set_tuning("davg_form", 3)
    a.run(4)
    a.sync()
    for g in range(4):
        b.run(1)
        b.sync()
        ref.generation(g)
        assert np.array_equal(b.last_parents(), ref.last_idx), "generation %d" % g
    assert np.array_equal(a.last_parents(), ref.last_idx)
    assert np.array_equal(a.pan_genome.read_matrix(), ref.acc) and np.array_equal(b.pan_genome.read_matrix(), ref.acc)
    # another matrix arrives between two generations: the distances computed ahead belong to the old one
    rng = np.random.default_rng(5)
    m = (rng.random(ref.acc.shape) < 0.4).astype(np.uint8)
    for s in (a, b):
        s.pan_genome.load_matrix(m)
    ref.acc = m.copy()
    for g in range(4, gens):
        a.run(1)
        b.run(1)
        ref.generation(g)
        assert np.array_equal(a.last_parents(), ref.last_idx), "generation %d" % g
        assert np.array_equal(b.last_parents(), ref.last_idx), "generation %d" % g
    assert np.array_equal(a.pan_genome.read_matrix(), ref.acc)
    assert np.array_equal(a.core_genome.read_matrix(), ref.core)
    a.close()
    b.close()


@pytest.mark.parametrize("N", [300, 5000])
def test_output_rows_follow_the_draws(pa, N):
    # The reference's child k is the child of draw k (population.rs:443, main.rs:445-447); the engine stores the children in
    # ascending parent order inside.  With every mutation and recombination rate at zero a generation is a pure gather, so
    # -- whatever the internal order, and without the oracle -- row k of both matrices after a generation must be row
    # last_parents()[k] of the matrices before it, the rows of the .csv files must be the rows read back, and the distances of
    # the run's fixed pair list must be those of rows (range1[k], range2[k]) of that matrix.  N = 300: wave sweep; 5000: window.
    kw = dict(pop_size=N, core_size=700, pan_genes=300, core_genes=100, core_mu=0.0, HR_rate=0.0, HGT_rate=0.0,
              rate_genes1=0.0, rate_genes2=0.0)
    sim = pa.Simulation(pa.make_params(seed=3, n_gen=4, max_distances=2000, **kw))
    rng = np.random.default_rng(N)
    core = (1 << rng.integers(0, 4, (N, 700))).astype(np.uint8)
    acc = (rng.random((N, 200)) < 0.4).astype(np.uint8)
    sim.core_genome.load_matrix(core)
    sim.pan_genome.load_matrix(acc)
    unsorted_seen = False
    for g in range(4):
        sim.run(1)
        sim.sync()
        par = sim.last_parents()
        unsorted_seen |= bool((np.diff(par.astype(np.int64)) < 0).any())
        new_core, new_acc = sim.core_genome.read_matrix(), sim.pan_genome.read_matrix()
        assert np.array_equal(new_core, core[par]) and np.array_equal(new_acc, acc[par]), "generation %d" % g
        core, acc = new_core, new_acc
    assert unsorted_seen                               # draw order, not the engine's ascending order
    cnt = sim.core_genome.pairwise_counts(sim.range1, sim.range2)[0]
    want = np.array([2 * int((core[i] != core[j]).sum()) for i, j in zip(sim.range1, sim.range2)], np.uint32)
    assert np.array_equal(cnt, want)
    core_d, acc_d = sim.final_distances()
    assert np.array_equal(core_d, (want // 2) / 700.0)
    sim.close()
