"""Host-side mirror of the reference's `Population` API (pansim/src/population.rs:164-897)
over the C ABI of libpansim_hip.so.  Method names and argument meaning follow the
reference so that parity tests read like tests of the reference would.

All computation happens in the HIP library; this module only marshals buffers.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, check


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def hamming_bitwise_fast(x, y):
    """distances.rs:22-52"""
    x = np.ascontiguousarray(x, np.uint8)
    y = np.ascontiguousarray(y, np.uint8)
    if x.size != y.size:
        raise ValueError("slices must have the same length (distances.rs:24 assert_eq)")
    out = C.c_uint32()
    check(_lib.load().ps_hamming_bitwise_fast(x, y, x.size, C.byref(out)))
    return out.value


def jaccard_distance_fast(x, y):
    """distances.rs:55-77 -> (intersection, union)"""
    x = np.ascontiguousarray(x, np.uint8)
    y = np.ascontiguousarray(y, np.uint8)
    if x.size != y.size:
        raise ValueError("slices must have the same length (distances.rs:56 assert_eq)")
    a, b = C.c_uint32(), C.c_uint32()
    check(_lib.load().ps_jaccard_distance_fast(x, y, x.size, C.byref(a), C.byref(b)))
    return a.value, b.value


def standard_deviation(values):
    """population.rs:87-94 -> (std, mean)"""
    v = _f64(values)
    s, m = C.c_double(), C.c_double()
    check(_lib.load().ps_standard_deviation(v, v.size, C.byref(s), C.byref(m)))
    return s.value, m.value


def int_to_base(n):
    """population.rs:154-162"""
    return _lib.load().ps_int_to_base(int(n)).decode()


def fmt_f64(v):
    """Rust `{}` Display of an f64 (main.rs:481, :496, :546)."""
    buf = C.create_string_buffer(512)
    rc = _lib.load().ps_fmt_f64(float(v), buf, 512)
    if rc < 0:
        check(rc)
    return buf.value.decode()


def init_vector(seed, core, ncols, avg_gene_freq=0.0, col_offset=0):
    out = np.zeros(ncols, np.uint8)
    check(_lib.load().ps_init_vector(int(seed), int(bool(core)), int(col_offset), int(ncols),
                                     float(avg_gene_freq), out))
    return out


def sample_weights(num_genes, logw, n_genes, avg_gene_num, avg_pairwise_dists,
                   no_control_genome_size, genome_size_penalty, competition_strength):
    """population.rs:293-437 (host half of sample_indices)"""
    ng = np.ascontiguousarray(num_genes, np.int32)
    w = np.zeros(ng.size, np.float64)
    check(_lib.load().ps_sample_weights(ng, _f64(logw), ng.size, int(n_genes), int(avg_gene_num),
                                        _f64(avg_pairwise_dists), int(no_control_genome_size),
                                        float(genome_size_penalty), float(competition_strength), w))
    return w


def draw_parents(weights, seed, generation):
    """population.rs:440-443"""
    w = _f64(weights)
    idx = np.zeros(w.size, np.uint32)
    check(_lib.load().ps_draw_parents(w, w.size, int(seed), int(generation), idx))
    return idx


class Population:
    """`struct Population` (population.rs:164-170) with its HBM state.

    Population::new (population.rs:181-190) takes (size, allele_count, max_variants, core,
    avg_gene_freq, rng, core_genes, acc_sampling_vec); `rng` becomes `seed` (the build's
    seeded host stream) and `acc_sampling_vec` is accepted and ignored, as in the reference
    (population.rs:189, :216).
    """

    def __init__(self, size, allele_count, max_variants, core, avg_gene_freq, seed, core_genes,
                 acc_sampling_vec=None, *, col_offset=0, global_cols=None, device=-1, init_vec=None,
                 _handle=None, _owned=True):
        self._lib = _lib.load()
        self.core = bool(core)
        self.size = int(size)
        self.ncols = int(allele_count)
        self.core_genes = int(core_genes)
        self.seed = int(seed)
        self.global_cols = self.ncols if global_cols is None else int(global_cols)
        self._owned = _owned
        if _handle is not None:
            self._h = C.c_void_p(_handle)
            return
        if self.core and max_variants != 4:
            raise ValueError("core populations use 4 variants (main.rs:375)")
        cfg = Config(self.size, self.ncols, self.global_cols, int(col_offset), self.core_genes,
                     self.seed, int(self.core), int(device))
        if init_vec is None:
            init_vec = init_vector(seed, core, self.ncols, avg_gene_freq, col_offset)
        init_vec = np.ascontiguousarray(init_vec, np.uint8)
        if init_vec.size != self.ncols:
            raise ValueError("init_vec must have allele_count entries")
        self._h = C.c_void_p()
        check(self._lib.ps_population_create(C.byref(cfg), init_vec.ctypes.data_as(C.c_void_p),
                                             C.byref(self._h)))

    # -- lifetime ---------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value and self._owned:
            self._lib.ps_population_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ------------------------------------------------------------------
    def load_matrix(self, rows):
        rows = np.ascontiguousarray(rows, np.uint8)
        if rows.shape != (self.size, self.ncols):
            raise ValueError("matrix must be (size, allele_count)")
        check(self._lib.ps_load_matrix(self._h, rows))

    def read_matrix(self):
        out = np.zeros((self.size, self.ncols), np.uint8)
        check(self._lib.ps_read_matrix(self._h, out))
        return out

    def set_rates(self, mutations_vec, recombinations_vec, comp_begin=None, comp_end=None):
        """The `mutations_vec` / `recombinations_vec` of mutate_alleles / recombine
        (population.rs:469, :546) and the gene range of each compartment's weight mask."""
        lm, lr = _f64(mutations_vec), _f64(recombinations_vec)
        if comp_begin is None:
            comp_begin, comp_end = [0], [self.global_cols]
        b = np.ascontiguousarray(comp_begin, np.uint64)
        e = np.ascontiguousarray(comp_end, np.uint64)
        check(self._lib.ps_set_rates(self._h, lm.size, lm, lr, b, e))

    # -- per-generation operators ---------------------------------------------------
    def next_generation(self, sample):
        """population.rs:450-465"""
        s = _u32(sample)
        if s.size != self.size:
            raise ValueError("sample must have one parent per individual")
        check(self._lib.ps_next_generation(self._h, s))

    def mutate_alleles(self, generation):
        """population.rs:467-542"""
        check(self._lib.ps_mutate_alleles(self._h, int(generation)))

    def recombine(self, generation):
        """population.rs:544-751"""
        check(self._lib.ps_recombine(self._h, int(generation)))

    def step(self, generation, sample, do_recombine=True):
        """next_generation + mutate_alleles + recombine fused (main.rs:445-464)"""
        s = _u32(sample)
        check(self._lib.ps_step(self._h, int(generation), s, int(bool(do_recombine))))

    def sample_indices(self, generation, avg_gene_num, avg_pairwise_dists, selection_coefficients,
                       verbose=False, no_control_genome_size=False, genome_size_penalty=0.99,
                       competition_strength=0.0):
        """population.rs:270-448"""
        idx = np.zeros(self.size, np.uint32)
        check(self._lib.ps_sample_indices(self._h, int(generation), int(avg_gene_num),
                                          _f64(avg_pairwise_dists), _f64(selection_coefficients),
                                          int(verbose), int(no_control_genome_size),
                                          float(genome_size_penalty), float(competition_strength), idx))
        return idx

    def fitness_terms(self, selection_coefficients):
        """population.rs:282-322 -> (num_genes, log_sum)"""
        ng = np.zeros(self.size, np.int32)
        lw = np.zeros(self.size, np.float64)
        check(self._lib.ps_fitness_terms(self._h, _f64(selection_coefficients), ng, lw))
        return ng, lw

    # -- measurements -----------------------------------------------------------------
    def average_distance(self):
        """population.rs:753-784"""
        out = np.zeros(self.size, np.float64)
        check(self._lib.ps_average_distance(self._h, out))
        return out

    def average_distance_rows(self, first, count):
        """rows [first, first + count) of average_distance (one rank's share of a row-sharded D-avg)"""
        out = np.zeros(int(count), np.float64)
        check(self._lib.ps_average_distance_rows(self._h, int(first), int(count), out))
        return out

    def pairwise_distances(self, max_distances, range1, range2):
        """population.rs:787-837"""
        r1, r2 = _u32(range1)[:max_distances], _u32(range2)[:max_distances]
        out = np.zeros(int(max_distances), np.float64)
        check(self._lib.ps_pairwise_distances(self._h, int(max_distances), np.ascontiguousarray(r1),
                                              np.ascontiguousarray(r2), out))
        return out

    def pairwise_counts(self, range1, range2):
        """integer numerators of pairwise_distances (host arrays)"""
        r1, r2 = _u32(range1), _u32(range2)
        a = np.zeros(r1.size, np.uint32)
        b = np.zeros(r1.size, np.uint32)
        check(self._lib.ps_pairwise_counts(self._h, r1.size, r1, r2, a.ctypes.data_as(C.c_void_p),
                                           b.ctypes.data_as(C.c_void_p), 0))
        return (a,) if self.core else (a, b)

    def pairwise_counts_device(self, range1, range2, out_a_ptr, out_b_ptr=0):
        """integer numerators written to caller-owned DEVICE memory (raw pointers)"""
        r1, r2 = _u32(range1), _u32(range2)
        check(self._lib.ps_pairwise_counts(self._h, r1.size, r1, r2, C.c_void_p(out_a_ptr),
                                           C.c_void_p(out_b_ptr), 1))

    def set_donor_shard(self, shard_rank, shard_count, fn=None):
        """HGT donors [N r / K, N (r + 1) / K) only (ps_set_donor_shard); `fn`: an _lib.EXCHANGE_FN object or None
        (own donors' events only).  The caller keeps `fn` alive."""
        ptr = C.cast(fn, C.c_void_p) if fn is not None else None
        self._exchange_fn = fn
        check(self._lib.ps_set_donor_shard(self._h, int(shard_rank), int(shard_count), ptr, None))

    def last_pair_form(self):
        """kernel form of the last core pair-count call (include/pansim_hip.h, PS_PAIR_FORM_*)"""
        return int(self._lib.ps_last_pair_form(self._h))

    def last_sweep_form(self):
        """kernel of the last core sweep launch (include/pansim_hip.h, PS_SWEEP_FORM_*)"""
        return int(self._lib.ps_last_sweep_form(self._h))

    def gene_frequencies(self):
        """population.rs:840-863"""
        out = np.zeros(self.ncols + self.core_genes, np.float64)
        check(self._lib.ps_gene_frequencies(self._h, out))
        return out

    def calc_gene_freq(self):
        """population.rs:244-268"""
        v = C.c_double()
        check(self._lib.ps_calc_gene_freq(self._h, C.byref(v)))
        return v.value

    def write(self, outpref):
        """population.rs:865-897"""
        check(self._lib.ps_write(self._h, str(outpref).encode()))

    def sync(self):
        check(self._lib.ps_sync(self._h))

    def set_tuning(self, key, value):
        """launch tuning / test hooks of ps_set_tuning (no reference counterpart)"""
        check(self._lib.ps_set_tuning(self._h, key.encode(), int(value)))
