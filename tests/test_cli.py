"""Command-line conformance of the `pansim` executable with the reference's clap definition
(main.rs:17-152), validation (main.rs:195-247: message on stdout, exit status 0, no files)
and output files (main.rs:321-331, :467-499, :531-553)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pansim_amd", "pansim")


@pytest.fixture(scope="module")
def exe(pa):
    if not os.path.exists(EXE):
        import __graft_entry__
        __graft_entry__.build()
    return EXE


def run(exe, *args, cwd=None):
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True, cwd=cwd, timeout=600)


def test_version_and_help(exe):
    r = run(exe, "--version")
    assert r.returncode == 0 and r.stdout == "pansim 0.1.0\n"          # main.rs:18
    r = run(exe, "--help")
    assert r.returncode == 0
    for flag, default in (("pop_size", "1000"), ("core_size", "1200000"), ("pan_genes", "6000"), ("core_genes", "2000"),
                          ("avg_gene_freq", "0.5"), ("n_gen", "100"), ("max_distances", "100000"), ("core_mu", "0.05"),
                          ("HR_rate", "0.05"), ("HGT_rate", "0.05"), ("rate_genes1", "1.0"), ("rate_genes2", "1000.0"),
                          ("prop_genes2", "0.1"), ("prop_positive", "-0.1"), ("pos_lambda", "10.0"), ("neg_lambda", "10.0"),
                          ("seed", "0"), ("outpref", "distances"), ("threads", "1"), ("genome_size_penalty", "0.99"),
                          ("competition_strength", "0.0")):
        assert "--%s <%s>" % (flag, flag) in r.stdout and "[default: %s]" % default in " ".join(r.stdout.split())
    for switch in ("print_dist", "print_matrices", "print_selection", "verbose", "no_control_genome_size"):
        assert "--%s\n" % switch in r.stdout


def test_help_is_the_reference_binarys_own_output_byte_for_byte(exe):
    # The one golden text the reference holds: `pansim --help` of a real Pansim binary (clap 3), pasted into its README
    # (/root/reference/README.md:40-138; extracted to tests/golden/help_usage.txt by tests/golden/make_help_usage.py).
    # From `USAGE:` on the build prints exactly that: options in clap's order (byte order, uppercase first, -h / -V by their
    # letters), help text and "[default: ...]" filled to 100 columns.  This check does not involve the oracle.
    want = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "help_usage.txt")).read()
    for flag in ("--help", "-h"):
        r = run(exe, flag)
        assert r.returncode == 0 and r.stderr == ""
        assert r.stdout[r.stdout.index("USAGE:"):] == want
        assert r.stdout.startswith("pansim 0.1.0\nSamuel Horsfield shorsfield@ebi.ac.uk\nRuns Wright-Fisher simulation")   # main.rs:17-20
        assert "--gpus" not in r.stdout and "MI355X" not in r.stdout
    # the two flags the reference does not have are listed by --help-extensions only
    r = run(exe, "--help-extensions")
    assert r.returncode == 0 and want.rstrip("\n") in r.stdout and "--gpus <gpus>" in r.stdout and "--reference_seed_stream" in r.stdout


def test_config1_as_written_is_the_reference_early_return(exe, tmp_path):
    # BASELINE configs[0] literally: --pan_genes 600 with the default --core_genes 2000
    r = run(exe, "--pop_size", 100, "--core_size", 12000, "--pan_genes", 600, "--n_gen", 50, "--seed", 0, cwd=tmp_path)
    assert r.returncode == 0
    assert r.stdout == "core_genes must be less than or equal to pan_size\n"
    assert os.listdir(tmp_path) == []


@pytest.mark.parametrize("args,first_line", [
    (["--HGT_rate", "-1"], "HR_rate and HGT_rate must be above 0.0"),
    (["--neg_lambda", "0"], "pos_lambda and neg_lambda must be above 0.0"),
    (["--rate_genes1", "-0.5"], "rate_genes1 and rate_genes2 must be >= 0"),
    (["--prop_genes2", "1.01"], "prop_genes2 must be 0.0 <= prop_genes2 <= 1.0"),
    (["--pop_size", "0"], "pop_size, core_size, pan_genes, n_gen and max_distances must all be above 1"),
    (["--pop_size", "0.4"], "pop_size, core_size, pan_genes, n_gen and max_distances must all be above 1"),
    (["--core_mu", "1.5"], "core_mu must be between 0.0 and 1.0"),
    (["--avg_gene_freq", "0"], "avg_gene_freq must be above 0.0 and below or equal to 1.0"),
])
def test_validation_failures_exit_zero(exe, tmp_path, args, first_line):
    r = run(exe, *args, cwd=tmp_path)
    assert r.returncode == 0 and r.stdout.splitlines()[0] == first_line
    assert os.listdir(tmp_path) == []


def test_prop_positive_accepts_hyphen_value_and_floats_round(exe, tmp_path):
    # main.rs:91 allow_hyphen_values; main.rs:155-156 f64-then-round ("1e3" is accepted)
    r = run(exe, "--prop_positive", "-0.5", "--pop_size", "1e3", "--core_mu", "7", cwd=tmp_path)
    assert r.returncode == 0 and r.stdout.startswith("core_mu must be between")


def test_bad_arguments(exe):
    assert run(exe, "--no_such_flag").returncode == 2
    assert run(exe, "--pop_size").returncode == 2
    assert run(exe, "--pop_size", "abc").returncode == 101      # value_of_t().unwrap() panic
    assert run(exe, "--max_distances", "1.5").returncode == 101  # usize parse (main.rs:169)
    assert run(exe, "--seed", "-3").returncode != 0


@pytest.mark.gpu
def test_outputs_match_oracle(exe, orc, tmp_path):
    from orc_sim import OracleSim
    kw = dict(pop_size=60, core_size=900, pan_genes=300, core_genes=100)
    args = []
    for k, v in kw.items():
        args += ["--" + k, v]
    r = run(exe, *args, "--n_gen", 4, "--seed", 11, "--max_distances", 500, "--outpref", tmp_path / "run",
            "--print_matrices", "--print_dist", "--print_selection", "--verbose", "--prop_positive", "0.2",
            "--threads", 3)
    assert r.returncode == 0, r.stderr
    ref = OracleSim(seed=11, prop_positive=0.2, **kw)
    r1, r2 = orc.sample_pairs(11, 60, 500)
    per_gen, verbose_lines = [], ["avg_gene_freq adjusted to %s" % orc.fmt_f64(ref.d.avg_gene_freq_adj)]
    import ctypes as C
    for g in range(4):
        ref.generation(g)
        cd = orc.pairwise_distances(ref.core, True, 100, r1, r2)
        ad = orc.pairwise_distances(ref.acc, False, 100, r1, r2)
        row = []
        for d in (cd, ad):
            s, m = C.c_double(), C.c_double()
            orc.lib().orc_standard_deviation(d, d.size, C.byref(s), C.byref(m))
            row += [m.value, s.value]
        per_gen.append(row)
        verbose_lines += ["Finished gen: %d" % (g + 1),
                          "avg_gene_freq: %s" % orc.fmt_f64(orc.lib().orc_calc_gene_freq(ref.acc, 60, 200))]
    assert r.stdout.splitlines() == verbose_lines                                   # main.rs:269-271, :522-526
    want = "".join("%s\t%s\n" % (orc.fmt_f64(c), orc.fmt_f64(a)) for c, a in zip(cd, ad))
    assert (tmp_path / "run.tsv").read_text() == want                              # main.rs:474-482
    want = "".join("%s\n" % orc.fmt_f64(x) for x in orc.gene_frequencies(ref.acc, 100))
    assert (tmp_path / "run_freqs.txt").read_text() == want                        # main.rs:487-497
    want = "".join("\t".join(orc.fmt_f64(x) for x in row) + "\n" for row in per_gen)
    assert (tmp_path / "run_per_gen.tsv").read_text() == want                      # main.rs:531-548
    want = "\n".join(orc.fmt_f64(x) for x in ref.sel) + "\n"
    assert (tmp_path / "run_selection.tsv").read_text() == want                    # main.rs:321-331
    orc.lib().orc_write_matrix(ref.core, 60, 900, 1, 100, str(tmp_path / "want").encode())
    orc.lib().orc_write_matrix(ref.acc, 60, 200, 0, 100, str(tmp_path / "want").encode())
    for suffix in ("_core_genome.csv", "_pangenome.csv"):
        assert (tmp_path / ("run" + suffix)).read_text() == (tmp_path / ("want" + suffix)).read_text()
    # no header, tab separated, as scripts/plot_distances.R:15-16 reads it
    first = (tmp_path / "run.tsv").read_text().splitlines()[0].split("\t")
    assert len(first) == 2 and all(0.0 <= float(x) <= 1.0 for x in first)


@pytest.mark.gpu
def test_reference_seed_stream_flag(exe, tmp_path):
    # SURVEY 8f-3, opt-in and unpinned: --reference_seed_stream draws _selection.tsv (main.rs:289-331) from the
    # reference's own ChaCha12 stream as restated in tests/test_reference_stream.py; without the flag the build's
    # Philox stream is used
    import ctypes as C
    import pansim_amd as pa
    from test_reference_stream import selection
    base = ["--pop_size", 40, "--core_size", 200, "--pan_genes", 260, "--core_genes", 60, "--n_gen", 2, "--seed", 9,
            "--prop_positive", 0.25, "--max_distances", 20, "--print_selection"]
    r = run(exe, *base, "--reference_seed_stream", "--outpref", tmp_path / "ref")
    assert r.returncode == 0, r.stderr
    got = (tmp_path / "ref_selection.tsv").read_text().split("\n")
    want = selection(9, 200, 0.25, 10.0, 10.0)
    assert got[-1] == "" and len(got) == 201
    assert got[:-1] == [pa.fmt_f64(x) for x in want]
    r = run(exe, *base, "--outpref", tmp_path / "own")
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "own_selection.tsv").read_text() != (tmp_path / "ref_selection.tsv").read_text()
