"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (pansim_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpansim_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pansim_oracle.c")
    hdr = os.path.join(_HERE, "pansim_oracle.h")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Params(C.Structure):
    _fields_ = [("pop_size", C.c_uint64), ("core_size", C.c_uint64), ("pan_genes", C.c_uint64),
                ("core_genes", C.c_uint64), ("avg_gene_freq", C.c_double), ("HR_rate", C.c_double),
                ("HGT_rate", C.c_double), ("core_mu", C.c_double), ("rate_genes1", C.c_double),
                ("rate_genes2", C.c_double), ("prop_genes2", C.c_double)]


class Derived(C.Structure):
    _fields_ = [("pan_size", C.c_uint64), ("avg_gene_freq_adj", C.c_double),
                ("avg_gene_num", C.c_int32), ("n_core_mutations", C.c_double),
                ("n_recombinations_core", C.c_double), ("n_recombinations_pan_total", C.c_double),
                ("n_comp", C.c_int), ("comp_begin", C.c_uint64 * 2), ("comp_end", C.c_uint64 * 2),
                ("n_pan_mutations", C.c_double * 2), ("n_recombinations_pan", C.c_double * 2)]


class CorePlan(C.Structure):
    _fields_ = [("T", C.c_uint32 * 7), ("has_events", C.c_uint32), ("k", C.c_uint32), ("R", C.c_uint32), ("cshift", C.c_uint32)]


_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u64, u32, i32, f64, ci = C.c_uint64, C.c_uint32, C.c_int32, C.c_double, C.c_int
    sig = {
        "orc_philox4x32_10": (None, [_u32p, _u32p, _u32p]),
        "orc_hs_f64": (f64, [u64, u32, u32, u64]),
        "orc_hs_u32": (u32, [u64, u32, u32, u64]),
        "orc_poisson": (u64, [f64, u64, u32, u32, C.POINTER(u64)]),
        "orc_derive": (None, [C.POINTER(Params), C.POINTER(Derived)]),
        "orc_core_plan_make": (None, [f64, f64, u64, C.POINTER(CorePlan)]),
        "orc_acc_flip_threshold": (u32, [f64, u64]),
        "orc_init_core_vec": (None, [u64, u64, _u8p]),
        "orc_init_acc_vec": (None, [u64, u64, f64, _u8p]),
        "orc_replicate": (None, [_u8p, u64, u64, _u8p]),
        "orc_selection_coefficients": (u64, [u64, u64, f64, f64, f64, _f64p]),
        "orc_sample_pairs": (None, [u64, u64, u64, _u32p, _u32p]),
        "orc_hamming_bitwise_fast": (u32, [_u8p, _u8p, C.c_size_t]),
        "orc_jaccard_distance_fast": (None, [_u8p, _u8p, C.c_size_t, C.POINTER(u32), C.POINTER(u32)]),
        "orc_pairwise_distances": (None, [_u8p, u64, u64, ci, u64, u64, _u32p, _u32p, _f64p]),
        "orc_pairwise_hamming_counts": (None, [_u8p, u64, u64, u64, u64, u64, _u32p, _u32p, _u32p]),
        "orc_average_distance": (None, [_u8p, u64, u64, ci, u64, _f64p]),
        "orc_gene_frequencies": (None, [_u8p, u64, u64, u64, _f64p]),
        "orc_calc_gene_freq": (f64, [_u8p, u64, u64]),
        "orc_next_generation": (None, [_u8p, u64, u64, _u32p, _u8p]),
        "orc_standard_deviation": (None, [_f64p, u64, C.POINTER(f64), C.POINTER(f64)]),
        "orc_int_to_base": (C.c_char, [C.c_uint8]),
        "orc_fitness_terms": (None, [_u8p, u64, u64, _f64p, _i32p, _f64p]),
        "orc_sample_weights": (ci, [_i32p, _f64p, u64, u64, i32, _f64p, ci, f64, f64, _f64p]),
        "orc_draw_parents": (ci, [_f64p, u64, u64, u32, _u32p]),
        "orc_sample_indices": (ci, [_u8p, u64, u64, u64, u32, i32, _f64p, _f64p, ci, f64, f64, _u32p]),
        "orc_mutate_core": (None, [_u8p, u64, u64, u64, u64, u32, C.POINTER(CorePlan)]),
        "orc_recombine_core": (None, [_u8p, u64, u64, u64, u64, u32, C.POINTER(CorePlan)]),
        "orc_mutate_acc": (None, [_u8p, u64, u64, u64, u32, ci, _u64p, _u64p, _f64p]),
        "orc_recombine_acc": (u64, [_u8p, u64, u64, u64, u32, ci, _u64p, _u64p, _f64p]),
        "orc_poisson_table": (u32, [f64, _u32p, _u32p, u32]),
        "orc_poisson_from_table": (u32, [u32, u32, _u32p, u32]),
        "orc_fmt_f64": (ci, [f64, C.c_char_p, C.c_size_t]),
        "orc_write_matrix": (ci, [_u8p, u64, u64, ci, u64, C.c_char_p]),
        "orc_ref_create": (C.c_void_p, [C.POINTER(Params), u64, ci]),
        "orc_ref_destroy": (None, [C.c_void_p]),
        "orc_ref_generation": (ci, [C.c_void_p, u32]),
        "orc_ref_set_competition": (None, [C.c_void_p, C.c_double]),
        "orc_ref_core": (C.c_void_p, [C.c_void_p]),
        "orc_ref_acc": (C.c_void_p, [C.c_void_p]),
        "orc_ref_pairwise": (None, [C.c_void_p, ci, u64, _u32p, _u32p, _f64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# --------------------------------------------------------------------------- helpers
def philox(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def make_params(pop_size=1000, core_size=1200000, pan_genes=6000, core_genes=2000,
                avg_gene_freq=0.5, HR_rate=0.05, HGT_rate=0.05, core_mu=0.05, rate_genes1=1.0,
                rate_genes2=1000.0, prop_genes2=0.1):
    return Params(pop_size, core_size, pan_genes, core_genes, avg_gene_freq, HR_rate, HGT_rate,
                  core_mu, rate_genes1, rate_genes2, prop_genes2)


def derive(params):
    d = Derived()
    lib().orc_derive(C.byref(params), C.byref(d))
    return d


def core_plan(lam_mut, lam_hr, L):
    p = CorePlan()
    lib().orc_core_plan_make(float(lam_mut), float(lam_hr), int(L), C.byref(p))
    return p


def fmt_f64(v):
    buf = C.create_string_buffer(512)
    lib().orc_fmt_f64(float(v), buf, 512)
    return buf.value.decode()


def hamming(x, y):
    x = np.ascontiguousarray(x, np.uint8)
    y = np.ascontiguousarray(y, np.uint8)
    return int(lib().orc_hamming_bitwise_fast(x, y, x.size))


def jaccard(x, y):
    x = np.ascontiguousarray(x, np.uint8)
    y = np.ascontiguousarray(y, np.uint8)
    a, b = C.c_uint32(), C.c_uint32()
    lib().orc_jaccard_distance_fast(x, y, x.size, C.byref(a), C.byref(b))
    return a.value, b.value


def pairwise_distances(pop, core, core_genes, r1, r2):
    pop = np.ascontiguousarray(pop, np.uint8)
    r1 = np.ascontiguousarray(r1, np.uint32)
    r2 = np.ascontiguousarray(r2, np.uint32)
    out = np.zeros(r1.size, np.float64)
    lib().orc_pairwise_distances(pop, pop.shape[0], pop.shape[1], int(core), int(core_genes),
                                 r1.size, r1, r2, out)
    return out


def pairwise_hamming_counts(pop, col_begin, col_end, r1, r2):
    pop = np.ascontiguousarray(pop, np.uint8)
    r1 = np.ascontiguousarray(r1, np.uint32)
    r2 = np.ascontiguousarray(r2, np.uint32)
    out = np.zeros(r1.size, np.uint32)
    lib().orc_pairwise_hamming_counts(pop, pop.shape[0], pop.shape[1], col_begin, col_end,
                                      r1.size, r1, r2, out)
    return out


def average_distance(pop, core, core_genes):
    pop = np.ascontiguousarray(pop, np.uint8)
    out = np.zeros(pop.shape[0], np.float64)
    lib().orc_average_distance(pop, pop.shape[0], pop.shape[1], int(core), int(core_genes), out)
    return out


def gene_frequencies(pop, core_genes):
    pop = np.ascontiguousarray(pop, np.uint8)
    out = np.zeros(pop.shape[1] + core_genes, np.float64)
    lib().orc_gene_frequencies(pop, pop.shape[0], pop.shape[1], int(core_genes), out)
    return out


def next_generation(pop, sample):
    pop = np.ascontiguousarray(pop, np.uint8)
    sample = np.ascontiguousarray(sample, np.uint32)
    nxt = np.zeros((sample.size, pop.shape[1]), np.uint8)
    lib().orc_next_generation(pop, sample.size, pop.shape[1], sample, nxt)
    return nxt


def fitness_terms(pop, sel):
    pop = np.ascontiguousarray(pop, np.uint8)
    sel = np.ascontiguousarray(sel, np.float64)
    ng = np.zeros(pop.shape[0], np.int32)
    lw = np.zeros(pop.shape[0], np.float64)
    lib().orc_fitness_terms(pop, pop.shape[0], pop.shape[1], sel, ng, lw)
    return ng, lw


def sample_weights(num_genes, logw, G, avg_gene_num, avg_dists, no_control, penalty, competition):
    num_genes = np.ascontiguousarray(num_genes, np.int32)
    logw = np.ascontiguousarray(logw, np.float64)
    avg_dists = np.ascontiguousarray(avg_dists, np.float64)
    w = np.zeros(num_genes.size, np.float64)
    rc = lib().orc_sample_weights(num_genes, logw, num_genes.size, int(G), int(avg_gene_num),
                                  avg_dists, int(no_control), float(penalty), float(competition), w)
    return rc, w


def draw_parents(weights, seed, gen):
    weights = np.ascontiguousarray(weights, np.float64)
    idx = np.zeros(weights.size, np.uint32)
    rc = lib().orc_draw_parents(weights, weights.size, int(seed), int(gen), idx)
    return rc, idx


def sample_indices(pop, seed, gen, avg_gene_num, avg_dists, sel, no_control=False, penalty=0.99,
                   competition=0.0):
    pop = np.ascontiguousarray(pop, np.uint8)
    idx = np.zeros(pop.shape[0], np.uint32)
    rc = lib().orc_sample_indices(pop, pop.shape[0], pop.shape[1], int(seed), int(gen),
                                  int(avg_gene_num), np.ascontiguousarray(avg_dists, np.float64),
                                  np.ascontiguousarray(sel, np.float64), int(no_control),
                                  float(penalty), float(competition), idx)
    return rc, idx


def init_core_vec(seed, L):
    v = np.zeros(L, np.uint8)
    lib().orc_init_core_vec(int(seed), int(L), v)
    return v


def init_acc_vec(seed, G, agf):
    v = np.zeros(G, np.uint8)
    lib().orc_init_acc_vec(int(seed), int(G), float(agf), v)
    return v


def selection_coefficients(seed, G, prop_positive, pos_lambda, neg_lambda):
    out = np.zeros(G, np.float64)
    lib().orc_selection_coefficients(int(seed), int(G), float(prop_positive), float(pos_lambda),
                                     float(neg_lambda), out)
    return out


def sample_pairs(seed, N, P):
    r1 = np.zeros(P, np.uint32)
    r2 = np.zeros(P, np.uint32)
    lib().orc_sample_pairs(int(seed), int(N), int(P), r1, r2)
    return r1, r2


def mutate_core(pop, site_offset, seed, gen, plan):
    assert pop.dtype == np.uint8 and pop.flags.c_contiguous
    lib().orc_mutate_core(pop, pop.shape[0], pop.shape[1], int(site_offset), int(seed), int(gen),
                          C.byref(plan))
    return pop


def recombine_core(pop, site_offset, seed, gen, plan):
    assert pop.dtype == np.uint8 and pop.flags.c_contiguous
    lib().orc_recombine_core(pop, pop.shape[0], pop.shape[1], int(site_offset), int(seed), int(gen),
                             C.byref(plan))
    return pop


def _comps(comp_begin, comp_end, lambdas):
    return (np.ascontiguousarray(comp_begin, np.uint64), np.ascontiguousarray(comp_end, np.uint64),
            np.ascontiguousarray(lambdas, np.float64))


def mutate_acc(pop, seed, gen, comp_begin, comp_end, lambdas):
    assert pop.dtype == np.uint8 and pop.flags.c_contiguous
    b, e, l = _comps(comp_begin, comp_end, lambdas)
    lib().orc_mutate_acc(pop, pop.shape[0], pop.shape[1], int(seed), int(gen), len(l), b, e, l)
    return pop


def poisson_table(lam):
    """(kmin, thresholds) of the integer Poisson inversion table of HGT event counts."""
    cap = int(24.0 * np.sqrt(lam) + 64.0)
    thr = np.zeros(cap, np.uint32)
    kmin = np.zeros(1, np.uint32)
    n = lib().orc_poisson_table(float(lam), kmin, thr, cap)
    return int(kmin[0]), thr[:n].copy()


def poisson_from_table(u, kmin, thr):
    return int(lib().orc_poisson_from_table(int(u), int(kmin), np.ascontiguousarray(thr, np.uint32), len(thr)))


def recombine_acc(pop, seed, gen, comp_begin, comp_end, lambdas):
    assert pop.dtype == np.uint8 and pop.flags.c_contiguous
    b, e, l = _comps(comp_begin, comp_end, lambdas)
    return int(lib().orc_recombine_acc(pop, pop.shape[0], pop.shape[1], int(seed), int(gen),
                                       len(l), b, e, l))


class RefSim:
    """Reference-algorithm (event-driven) CPU mode: the 'port' CPU baseline."""

    def __init__(self, params, seed=0, threads=1, competition_strength=0.0):
        self.params = params
        self.d = derive(params)
        self.h = lib().orc_ref_create(C.byref(params), int(seed), int(threads))
        if competition_strength:
            lib().orc_ref_set_competition(self.h, float(competition_strength))
        self.N, self.L, self.G = params.pop_size, params.core_size, self.d.pan_size

    def generation(self, gen):
        rc = lib().orc_ref_generation(self.h, int(gen))
        if rc:
            raise RuntimeError("orc_ref_generation failed: %d" % rc)

    def core(self):
        p = lib().orc_ref_core(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (self.N, self.L))

    def acc(self):
        p = lib().orc_ref_acc(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (self.N, self.G))

    def pairwise(self, core, r1, r2):
        out = np.zeros(len(r1), np.float64)
        lib().orc_ref_pairwise(self.h, int(core), len(r1), np.ascontiguousarray(r1, np.uint32),
                               np.ascontiguousarray(r2, np.uint32), out)
        return out

    def close(self):
        if self.h:
            lib().orc_ref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
