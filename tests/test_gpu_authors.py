"""The reference authors' own scripted workload (scripts/run_pansim_benchmark.sh:2-12, :235-278): N = 1000,
--pan_genes 4400 --core_genes 1342 --avg_gene_freq 0.45 --core_mu 0.019, ONE accessory compartment (--prop_genes2 0.0),
--HR_rate 0 --HGT_rate 0 (both `recombine` calls skipped, main.rs:459-464), --pos_lambda / --neg_lambda 100, --verbose,
--threads 4 and --competition_strength in {0, 100, 1e4, 1e8} -- with --core_size shrunk from 1 342 000 (the oracle walks
every cell).  At 1e4 / 1e8 the competition softmax saturates (population.rs:374-393): one individual parents the whole
next generation, the window / wave sweeps gather from a single column, and D-avg must agree with the oracle to the last
bit because strength * ln(avg) amplifies one ulp to a different winner."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pansim_amd", "pansim")
STRENGTHS = (0.0, 100.0, 1e4, 1e8)
AUTHORS = dict(pop_size=1000, pan_genes=4400, core_genes=1342, avg_gene_freq=0.45, core_mu=0.019, HR_rate=0.0, HGT_rate=0.0,
               rate_genes1=1.0, rate_genes2=1000.0, prop_genes2=0.0)
EXTRA = dict(prop_positive=-0.1, pos_lambda=100.0, neg_lambda=100.0)
MIN_POSITIVE = 2.2250738585072014e-308


@pytest.mark.parametrize("strength", STRENGTHS)
def test_authors_flags_generation_loop(pa, orc, strength):
    from orc_sim import OracleSim
    kw = dict(AUTHORS, core_size=2684)
    gens, P = 22, 3000
    sim = pa.Simulation(pa.make_params(seed=0, n_gen=gens, max_distances=P, competition_strength=strength, **kw, **EXTRA))
    ref = OracleSim(seed=0, competition_strength=strength, **kw, **EXTRA)
    assert ref.d.n_comp == 1 and ref.cb == [0] and ref.ce == [3058]            # one compartment (main.rs:341, :355)
    distinct = []
    for g in range(gens):
        sim.run(1)
        sim.sync()
        ref.generation(g)
        assert np.array_equal(sim.last_parents(), ref.last_idx), "parents differ at generation %d" % g
        distinct.append(len(np.unique(ref.last_idx)))
    assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    r1, r2 = orc.sample_pairs(0, 1000, P)
    core_d, acc_d = sim.final_distances()
    assert np.array_equal(core_d, orc.pairwise_distances(ref.core, True, 1342, r1, r2))
    assert np.array_equal(acc_d, orc.pairwise_distances(ref.acc, False, 1342, r1, r2))
    assert np.array_equal(sim.pan_genome.gene_frequencies(), orc.gene_frequencies(ref.acc, 1342))
    # what the strengths do (generation 0 starts clonal: uniform draws whatever the strength)
    if strength >= 1e8:
        assert distinct[1:] == [1] * (gens - 1), distinct
    elif strength >= 1e4:
        assert max(distinct[1:]) <= 5, distinct
    else:
        assert min(distinct) > 400, distinct
    sim.close()


def test_authors_flags_unbatched_equals_batched(pa):
    # the same run without a host synchronisation between generations (the CLI's form): same parents, same matrices
    kw = dict(AUTHORS, core_size=2000)
    a = pa.Simulation(pa.make_params(seed=4, n_gen=12, max_distances=100, competition_strength=1e4, **kw, **EXTRA))
    b = pa.Simulation(pa.make_params(seed=4, n_gen=12, max_distances=100, competition_strength=1e4, **kw, **EXTRA))
    a.run(12)
    a.sync()
    for g in range(12):
        b.run(1)
        b.sync()
    assert np.array_equal(a.last_parents(), b.last_parents())
    assert np.array_equal(a.pan_genome.read_matrix(), b.pan_genome.read_matrix())
    assert np.array_equal(a.core_genome.read_matrix(), b.core_genome.read_matrix())
    a.close()
    b.close()


@pytest.mark.parametrize("strength", ["0.0", "100", "10000", "100000000"])
def test_authors_command_line(pa, orc, tmp_path, strength):
    # the script's command line as written (flag order, --threads 4, --verbose, strengths spelled as the script spells
    # them), --core_size shrunk: stdout and the two output files against the oracle
    from orc_sim import OracleSim
    n_gen, core_size = 21, 2684
    args = ["--n_gen", n_gen, "--pop_size", 1000, "--core_size", core_size, "--pan_genes", 4400, "--core_genes", 1342,
            "--avg_gene_freq", 0.45, "--threads", 4, "--core_mu", 0.019, "--HR_rate", 0.0, "--HGT_rate", 0.0, "--rate_genes1", 1.0,
            "--rate_genes2", 1000, "--prop_genes2", 0.0, "--prop_positive", -0.1, "--pos_lambda", 100, "--neg_lambda", 100,
            "--outpref", tmp_path / "run", "--verbose", "--competition_strength", strength]
    r = subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    ref = OracleSim(seed=0, competition_strength=float(strength), **dict(AUTHORS, core_size=core_size), **EXTRA)
    lines = ["avg_gene_freq adjusted to %s" % orc.fmt_f64(ref.d.avg_gene_freq_adj)]
    for g in range(n_gen):
        ref.generation(g)
        lines += ["Finished gen: %d" % (g + 1), "avg_gene_freq: %s" % orc.fmt_f64(orc.lib().orc_calc_gene_freq(ref.acc, 1000, 3058))]
    assert r.stdout.splitlines() == lines                                          # main.rs:269-271, :522-526
    r1, r2 = orc.sample_pairs(0, 1000, 100000)
    cd = orc.pairwise_distances(ref.core, True, 1342, r1, r2)
    ad = orc.pairwise_distances(ref.acc, False, 1342, r1, r2)
    want = "".join("%s\t%s\n" % (orc.fmt_f64(c), orc.fmt_f64(a)) for c, a in zip(cd, ad))
    assert (tmp_path / "run.tsv").read_text() == want                              # main.rs:474-482
    want = "".join("%s\n" % orc.fmt_f64(x) for x in orc.gene_frequencies(ref.acc, 1342))
    assert (tmp_path / "run_freqs.txt").read_text() == want                        # main.rs:487-497
    assert sorted(os.listdir(tmp_path)) == ["run.tsv", "run_freqs.txt"]


@pytest.mark.parametrize("strength", [1e4, 1e8])
def test_clonal_collapse_cycle(pa, orc, strength):
    # the cycle VERDICT round 4 asks to cross: saturated softmax -> a single parent -> (no gain/loss event in the whole
    # population that generation) all rows identical -> every distance exactly 0 -> MIN_POSITIVE (population.rs:774-776)
    # -> strength * ln equal for all -> uniform draws again.  Gain / loss slowed to ~1 event per generation in the whole
    # population so that both states alternate within 40 generations.
    from orc_sim import OracleSim
    N, G_all, cg = 96, 420, 120
    kw = dict(pop_size=N, core_size=640, pan_genes=G_all, core_genes=cg, avg_gene_freq=0.5, core_mu=0.019, HR_rate=0.0, HGT_rate=0.0,
              rate_genes1=1.0 / (N * (G_all - cg)), rate_genes2=1000.0, prop_genes2=0.0)
    gens = 40
    sim = pa.Simulation(pa.make_params(seed=6, n_gen=gens, max_distances=200, competition_strength=strength, **kw, **EXTRA))
    ref = OracleSim(seed=6, competition_strength=strength, **kw, **EXTRA)
    clonal = single = 0
    for g in range(gens):
        avg = orc.average_distance(ref.acc, False, cg)
        got = sim.pan_genome.average_distance()
        assert np.array_equal(got, avg)
        clonal += bool((avg == MIN_POSITIVE).all())
        sim.run(1)
        sim.sync()
        ref.generation(g)
        assert np.array_equal(sim.last_parents(), ref.last_idx), "parents differ at generation %d" % g
        single += len(np.unique(ref.last_idx)) == 1
    assert clonal >= 2 and single >= 2 and clonal < gens, (clonal, single)
    assert np.array_equal(sim.pan_genome.read_matrix(), ref.acc)
    assert np.array_equal(sim.core_genome.read_matrix(), ref.core)
    sim.close()


def test_all_zero_weights_fall_back_to_uniform_on_the_device_path(pa, orc):
    # population.rs:403, :435-437 through ps_sample_weights (the host half of P-size / P-comp): the contrived vector of
    # tests/test_oracle_independent.py::test_all_zero_weights_fall_back_to_uniform
    G, cg = 40, 5
    pop = np.zeros((6, G), np.uint8)
    pop[:, :10] = 1
    pop[5, 10:14] = 1
    avg = orc.average_distance(pop, False, cg)
    num, logw = orc.fitness_terms(pop, np.zeros(G))
    for no_control in (False, True):
        rc, want = orc.sample_weights(num, logw, G, 10, avg, no_control, 1e-300, 1e8)
        got = np.zeros(6)
        pa._lib.check(pa.load().ps_sample_weights(num, logw, 6, G, 10, avg, int(no_control), 1e-300, 1e8, got))
        assert rc == 0 and np.array_equal(got, want)
        assert (got == 1.0).all() if not no_control else ((got > 0.0).sum() == 1 and got[5] > 0.0)
