"""ctypes loader for libpansim_hip.so (the C ABI of include/pansim_hip.h).

The library is built in-tree by `make -C pansim_amd/csrc` (see __graft_entry__.build).
There is no fallback: if the shared object is missing, loading fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PANSIM_HIP_LIBRARY") or os.path.join(_HERE, "libpansim_hip.so")

PS_OK = 0
PS_ERR_INVALID, PS_ERR_NO_DEVICE, PS_ERR_OOM, PS_ERR_WEIGHTS, PS_ERR_IO, PS_ERR_STATE = -1, -2, -3, -4, -5, -6


class PansimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpansim_hip error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("pop_size", C.c_uint64), ("ncols", C.c_uint64), ("global_cols", C.c_uint64),
                ("col_offset", C.c_uint64), ("core_genes", C.c_uint64), ("seed", C.c_uint64),
                ("core", C.c_int32), ("device", C.c_int32)]


class SimParams(C.Structure):
    _fields_ = [("pop_size", C.c_uint64), ("core_size", C.c_uint64), ("pan_genes", C.c_uint64),
                ("core_genes", C.c_uint64), ("avg_gene_freq", C.c_double), ("HR_rate", C.c_double),
                ("HGT_rate", C.c_double), ("n_gen", C.c_int32), ("max_distances", C.c_uint64),
                ("core_mu", C.c_double), ("rate_genes1", C.c_double), ("rate_genes2", C.c_double),
                ("prop_genes2", C.c_double), ("prop_positive", C.c_double),
                ("pos_lambda", C.c_double), ("neg_lambda", C.c_double), ("seed", C.c_uint64),
                ("print_dist", C.c_int32), ("print_matrices", C.c_int32),
                ("print_selection", C.c_int32), ("verbose", C.c_int32),
                ("no_control_genome_size", C.c_int32), ("genome_size_penalty", C.c_double),
                ("competition_strength", C.c_double), ("shard_rank", C.c_int32),
                ("shard_count", C.c_int32), ("device", C.c_int32), ("reference_seed_stream", C.c_int32)]


class Derived(C.Structure):
    _fields_ = [("pan_size", C.c_uint64), ("avg_gene_freq_adj", C.c_double),
                ("avg_gene_num", C.c_int32), ("n_core_mutations", C.c_double),
                ("n_recombinations_core", C.c_double), ("n_recombinations_pan_total", C.c_double),
                ("n_comp", C.c_int32), ("comp_begin", C.c_uint64 * 2), ("comp_end", C.c_uint64 * 2),
                ("n_pan_mutations", C.c_double * 2), ("n_recombinations_pan", C.c_double * 2)]


# every symbol include/pansim_hip.h declares (tests/test_host_logic.py::test_library_exports_every_declared_symbol checks the header against this)
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_vp, _u64, _u32, _i32, _f64, _int = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_double, C.c_int
# ps_exchange_fn: int (*)(void *ctx, void *d_words, uint64_t n_words, void *hip_stream)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)

SIGNATURES = {
    "ps_last_error": (C.c_char_p, []),
    "ps_abi_version": (_int, []),
    "ps_device_count": (_int, []),
    "ps_population_create": (_int, [C.POINTER(Config), _vp, C.POINTER(_vp)]),
    "ps_population_destroy": (None, [_vp]),
    "ps_init_vector": (_int, [_u64, _int, _u64, _u64, _f64, _u8p]),
    "ps_load_matrix": (_int, [_vp, _u8p]),
    "ps_read_matrix": (_int, [_vp, _u8p]),
    "ps_set_rates": (_int, [_vp, _int, _f64p, _f64p, _u64p, _u64p]),
    "ps_next_generation": (_int, [_vp, _u32p]),
    "ps_mutate_alleles": (_int, [_vp, _u32]),
    "ps_recombine": (_int, [_vp, _u32]),
    "ps_step": (_int, [_vp, _u32, _u32p, _int]),
    "ps_set_donor_shard": (_int, [_vp, _u32, _u32, _vp, _vp]),
    "ps_sample_indices": (_int, [_vp, _u32, _i32, _f64p, _f64p, _int, _int, _f64, _f64, _u32p]),
    "ps_fitness_terms": (_int, [_vp, _f64p, _i32p, _f64p]),
    "ps_sample_weights": (_int, [_i32p, _f64p, _u64, _u64, _i32, _f64p, _int, _f64, _f64, _f64p]),
    "ps_draw_parents": (_int, [_f64p, _u64, _u64, _u32, _u32p]),
    "ps_average_distance": (_int, [_vp, _f64p]),
    "ps_average_distance_rows": (_int, [_vp, _u64, _u64, _f64p]),
    "ps_pairwise_distances": (_int, [_vp, _u64, _u32p, _u32p, _f64p]),
    "ps_pairwise_counts": (_int, [_vp, _u64, _u32p, _u32p, _vp, _vp, _int]),
    "ps_last_pair_form": (_int, [_vp]),
    "ps_last_sweep_form": (_int, [_vp]),
    "ps_gene_frequencies": (_int, [_vp, _f64p]),
    "ps_calc_gene_freq": (_int, [_vp, C.POINTER(_f64)]),
    "ps_write": (_int, [_vp, C.c_char_p]),
    "ps_sync": (_int, [_vp]),
    "ps_set_tuning": (_int, [_vp, C.c_char_p, C.c_int64]),
    "ps_poisson_table": (_u32, [_f64, _u32p, _u32p, _u32]),
    "ps_hamming_bitwise_fast": (_int, [_u8p, _u8p, C.c_size_t, C.POINTER(_u32)]),
    "ps_jaccard_distance_fast": (_int, [_u8p, _u8p, C.c_size_t, C.POINTER(_u32), C.POINTER(_u32)]),
    "ps_standard_deviation": (_int, [_f64p, _u64, C.POINTER(_f64), C.POINTER(_f64)]),
    "ps_int_to_base": (C.c_char, [C.c_uint8]),
    "ps_fmt_f64": (_int, [_f64, C.c_char_p, C.c_size_t]),
    "ps_sim_default_params": (None, [C.POINTER(SimParams)]),
    "ps_sim_validate": (_int, [C.POINTER(SimParams), C.c_char_p, C.c_size_t]),
    "ps_sim_derive": (_int, [C.POINTER(SimParams), C.POINTER(Derived)]),
    "ps_selection_coefficients": (_int, [_u64, _u64, _f64, _f64, _f64, _f64p]),
    "ps_sample_pairs": (_int, [_u64, _u64, _u64, _u32p, _u32p]),
    "ps_reference_selection_coefficients": (_int, [_u64, _u64, _f64, _f64, _f64, _f64p]),
    "ps_chacha_block": (None, [_u32p, _u64, _u64, _int, _u32p]),
    "ps_sim_create": (_int, [C.POINTER(SimParams), C.POINTER(_vp)]),
    "ps_sim_destroy": (None, [_vp]),
    "ps_sim_run": (_int, [_vp, _u32, _u32]),
    "ps_sim_sync": (_int, [_vp]),
    "ps_sim_set_exchange": (_int, [_vp, _vp, _vp]),
    "ps_sim_emulate_exchange": (_int, [_vp, _int]),
    "ps_sim_exchange_stats": (_int, [_vp, _int, C.POINTER(_u64), C.POINTER(_u64)]),
    "ps_sim_emulated_link_time": (_int, [_vp, _int, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64)]),
    "ps_rccl_available": (_int, []),
    "ps_rccl_unique_id": (_int, [_u8p]),
    "ps_rccl_exchange_create": (_int, [_u8p, _int, _int, _int, C.POINTER(_vp)]),
    "ps_rccl_exchange_destroy": (None, [_vp]),
    "ps_exchange_rccl": (_int, [_vp, _vp, _u64, _vp]),
    "ps_rccl_exchange_stats": (_int, [_vp, _int, C.POINTER(_u64), C.POINTER(_u64)]),
    "ps_sim_core": (_vp, [_vp]),
    "ps_sim_acc": (_vp, [_vp]),
    "ps_sim_selection": (C.POINTER(_f64), [_vp]),
    "ps_sim_range1": (C.POINTER(_u32), [_vp]),
    "ps_sim_range2": (C.POINTER(_u32), [_vp]),
    "ps_sim_last_parents": (_int, [_vp, _u32p]),
    "ps_sim_sweep_timing": (_int, [_vp, _int, C.POINTER(_u64), C.POINTER(_f64), C.POINTER(_f64)]),
    "ps_sim_enable_timing": (_int, [_vp, _int]),
    "ps_sim_pairwise_distances": (_int, [_vp, _f64p, _f64p]),
    "ps_sim_distance_timing": (_int, [_vp, C.POINTER(_f64), C.POINTER(_f64)]),
    "ps_sim_host_timing": (_int, [_vp, _int, C.POINTER(_u64), C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64)]),
    "ps_multi_create": (_int, [C.POINTER(SimParams), _int, C.POINTER(_int), C.POINTER(_vp)]),
    "ps_multi_destroy": (None, [_vp]),
    "ps_multi_shards": (_int, [_vp]),
    "ps_multi_shard": (_vp, [_vp, _int]),
    "ps_multi_run": (_int, [_vp, _u32, _u32]),
    "ps_multi_sync": (_int, [_vp]),
    "ps_multi_pairwise_counts": (_int, [_vp, _u32p]),
    "ps_multi_pairwise_distances": (_int, [_vp, _f64p, _f64p]),
    "ps_multi_write": (_int, [_vp, C.c_char_p]),
}

_lib = None


def load():
    """Load libpansim_hip.so; raises if it has not been built (no fallback).

    A process that also uses torch (pansim_amd.distributed, bench.py) must `import torch` BEFORE this call: torch wheels
    bundle their own HIP / HSA / RCCL libraries under the same sonames, and only when they are loaded first does the
    process end up with a single HIP runtime (tests/conftest.py has the details)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libpansim_hip.so is missing at %s: build it with `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (or `make -C pansim_amd/csrc`).  pansim_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != PS_OK:
        raise PansimError(rc, load().ps_last_error().decode(errors="replace"))
    return rc
