"""The reference's main() as a library (pansim/src/main.rs:155-553) over the C ABI:
parameter validation/derivation, the seeded host draws and the generation loop.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Derived, SimParams, check
from .population import Population, fmt_f64, standard_deviation

# flag names and defaults of main.rs:21-151
DEFAULTS = dict(pop_size=1000, core_size=1200000, pan_genes=6000, core_genes=2000, avg_gene_freq=0.5,
                n_gen=100, max_distances=100000, core_mu=0.05, HR_rate=0.05, HGT_rate=0.05,
                rate_genes1=1.0, rate_genes2=1000.0, prop_genes2=0.1, prop_positive=-0.1,
                pos_lambda=10.0, neg_lambda=10.0, seed=0, genome_size_penalty=0.99,
                competition_strength=0.0, print_dist=0, print_matrices=0, print_selection=0,
                verbose=0, no_control_genome_size=0, shard_rank=0, shard_count=1, device=-1,
                reference_seed_stream=0)


def make_params(**kw):
    p = SimParams()
    _lib.load().ps_sim_default_params(C.byref(p))
    for k, v in kw.items():
        if k not in DEFAULTS:
            raise TypeError("unknown parameter %r" % k)
        setattr(p, k, v)
    return p


def validate(params):
    """main.rs:195-247 -> (ok, stdout text the reference prints before `return Ok(())`)"""
    buf = C.create_string_buffer(2048)
    rc = _lib.load().ps_sim_validate(C.byref(params), buf, 2048)
    return rc == 0, buf.value.decode()


def derive(params):
    """main.rs:259-367"""
    d = Derived()
    check(_lib.load().ps_sim_derive(C.byref(params), C.byref(d)))
    return d


def selection_coefficients(seed, n_genes, prop_positive, pos_lambda, neg_lambda):
    out = np.zeros(n_genes, np.float64)
    check(_lib.load().ps_selection_coefficients(int(seed), int(n_genes), float(prop_positive),
                                                float(pos_lambda), float(neg_lambda), out))
    return out


def sample_pairs(seed, pop_size, max_distances):
    r1 = np.zeros(max_distances, np.uint32)
    r2 = np.zeros(max_distances, np.uint32)
    check(_lib.load().ps_sample_pairs(int(seed), int(pop_size), int(max_distances), r1, r2))
    return r1, r2


class Simulation:
    """State of main() between main.rs:259 and :553 for one process (one GPU)."""

    def __init__(self, params=None, _handle=None, **kw):
        self._lib = _lib.load()
        self.params = params if params is not None else make_params(**kw)
        self.derived = derive(self.params)
        self._owned = _handle is None
        self._h = C.c_void_p(_handle) if _handle is not None else C.c_void_p()
        if self._owned:
            check(self._lib.ps_sim_create(C.byref(self.params), C.byref(self._h)))
        p, d = self.params, self.derived
        sb = p.core_size * p.shard_rank // p.shard_count
        se = p.core_size * (p.shard_rank + 1) // p.shard_count
        self.core_genome = Population(p.pop_size, se - sb, 4, True, 0.0, p.seed, p.core_genes,
                                      global_cols=p.core_size, _handle=self._lib.ps_sim_core(self._h),
                                      _owned=False)
        self.pan_genome = Population(p.pop_size, d.pan_size, 2, False, d.avg_gene_freq_adj, p.seed,
                                     p.core_genes, _handle=self._lib.ps_sim_acc(self._h), _owned=False)
        P = p.max_distances
        self.range1 = np.ctypeslib.as_array(self._lib.ps_sim_range1(self._h), (P,)).copy()
        self.range2 = np.ctypeslib.as_array(self._lib.ps_sim_range2(self._h), (P,)).copy()
        G = d.pan_size
        self.selection_weights = (np.ctypeslib.as_array(self._lib.ps_sim_selection(self._h), (G,)).copy()
                                  if G else np.zeros(0))
        self.generation = 0

    def close(self):
        # core_genome / pan_genome borrow handles owned by the ps_sim: they die with it
        for name in ("core_genome", "pan_genome"):
            pop = getattr(self, name, None)
            if pop is not None:
                pop._h = C.c_void_p()
        if getattr(self, "_h", None) and self._h.value and self._owned:
            self._lib.ps_sim_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, count, first_generation=None):
        """main.rs:429-464 for `count` generations (asynchronous; call sync())"""
        g0 = self.generation if first_generation is None else int(first_generation)
        check(self._lib.ps_sim_run(self._h, g0, int(count)))
        self.generation = g0 + int(count)

    def sync(self):
        check(self._lib.ps_sim_sync(self._h))

    def last_parents(self):
        out = np.zeros(self.params.pop_size, np.uint32)
        check(self._lib.ps_sim_last_parents(self._h, out))
        return out

    def set_exchange(self, fn, ctx=None):
        """shard the HGT donors over the site shards of this run; `fn` (an _lib.EXCHANGE_FN object, or the address of a
        native ps_exchange_fn such as ps_exchange_rccl with its handle as `ctx`) ORs the shards' delta buffers once per
        generation (ps_sim_set_exchange).  Every shard of the run must install one."""
        self._exchange_fn = fn          # keep the ctypes thunk alive
        check(self._lib.ps_sim_set_exchange(self._h, C.cast(fn, C.c_void_p), ctx))

    def emulate_exchange(self, n_shards):
        """bench.py --emulate-shard: shard 0 of n_shards, exchange stood in for by device-local copies (timing only)"""
        check(self._lib.ps_sim_emulate_exchange(self._h, int(n_shards)))

    def emulated_link_time(self, reset=True):
        """(modelled microseconds of link time charged by the emulated exchange since the last reset, GB/s per link, latency us)"""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(self._lib.ps_sim_emulated_link_time(self._h, int(reset), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def exchange_stats(self, reset=True):
        """(exchange calls, bytes sent + received by this shard in them) since the last reset -- library providers only"""
        n, b = C.c_uint64(), C.c_uint64()
        check(self._lib.ps_sim_exchange_stats(self._h, int(reset), C.byref(n), C.byref(b)))
        return n.value, b.value

    def enable_timing(self, on=True):
        check(self._lib.ps_sim_enable_timing(self._h, int(on)))

    def sweep_timing(self, reset=True):
        n, ms, b = C.c_uint64(), C.c_double(), C.c_double()
        check(self._lib.ps_sim_sweep_timing(self._h, int(reset), C.byref(n), C.byref(ms), C.byref(b)))
        return n.value, ms.value, b.value

    def host_timing(self, reset=True):
        """(generations, ms waiting for the device half, ms in the softmaxes, ms in the parent draw) since the last reset"""
        n, a, b, c = C.c_uint64(), C.c_double(), C.c_double(), C.c_double()
        check(self._lib.ps_sim_host_timing(self._h, int(reset), C.byref(n), C.byref(a), C.byref(b), C.byref(c)))
        return n.value, a.value, b.value, c.value

    # -- outputs of main.rs:467-499 ---------------------------------------------------
    def final_distances(self):
        """main.rs:467-470 (ps_sim_pairwise_distances: both matrices' kernels enqueued together)"""
        P = self.params.max_distances
        core, acc = np.zeros(P), np.zeros(P)
        check(self._lib.ps_sim_pairwise_distances(self._h, core, acc))
        return core, acc

    def distance_timing(self):
        """device ms of the core / accessory distance kernels of the last final_distances() call"""
        a, b = C.c_double(), C.c_double()
        check(self._lib.ps_sim_distance_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def write_outputs(self, outpref):
        core, acc = self.final_distances()
        with open(outpref + ".tsv", "w") as f:                       # main.rs:474-482
            f.writelines("%s\t%s\n" % (fmt_f64(c), fmt_f64(a)) for c, a in zip(core, acc))
        with open(outpref + "_freqs.txt", "w") as f:                 # main.rs:487-497
            f.writelines("%s\n" % fmt_f64(x) for x in self.pan_genome.gene_frequencies())
        if self.params.print_matrices:                               # main.rs:550-553
            self.core_genome.write(outpref)
            self.pan_genome.write(outpref)


class MultiSimulation:
    """The run sharded by core site inside ONE process (ps_multi): shard k on HIP device devices[k]
    (ordinals may repeat), one host thread per shard inside each call; results equal the unsharded run."""

    def __init__(self, params, n_shards, devices=None):
        self._lib = _lib.load()
        self.params = params
        self._h = C.c_void_p()
        dev = None
        if devices is not None:
            dev = (C.c_int * int(n_shards))(*[int(x) for x in devices])
        check(self._lib.ps_multi_create(C.byref(params), int(n_shards), dev, C.byref(self._h)))
        self.n_shards = self._lib.ps_multi_shards(self._h)
        self.shards = []
        for k in range(self.n_shards):
            q = make_params(**{f: getattr(params, f) for f, _ in params._fields_})
            q.shard_rank, q.shard_count = k, self.n_shards
            self.shards.append(Simulation(q, _handle=self._lib.ps_multi_shard(self._h, k)))
        self.range1, self.range2 = self.shards[0].range1, self.shards[0].range2
        self.pan_genome = self.shards[0].pan_genome
        self.generation = 0

    def close(self):
        # the shard wrappers borrow handles owned by the ps_multi: null them first, so that a later use raises
        # a PansimError (null handle) instead of touching freed memory
        for s in getattr(self, "shards", []):
            for pop in (s.core_genome, s.pan_genome):
                pop._h = C.c_void_p()
            s._h = C.c_void_p()
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ps_multi_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, count, first_generation=None):
        g0 = self.generation if first_generation is None else int(first_generation)
        check(self._lib.ps_multi_run(self._h, g0, int(count)))
        self.generation = g0 + int(count)

    def sync(self):
        check(self._lib.ps_multi_sync(self._h))

    def pairwise_counts(self):
        out = np.zeros(self.params.max_distances, np.uint32)
        check(self._lib.ps_multi_pairwise_counts(self._h, out))
        return out

    def final_distances(self):
        P = self.params.max_distances
        core, acc = np.zeros(P), np.zeros(P)
        check(self._lib.ps_multi_pairwise_distances(self._h, core, acc))
        return core, acc

    def write(self, outpref):
        check(self._lib.ps_multi_write(self._h, str(outpref).encode()))


__all__ = ["Simulation", "MultiSimulation", "make_params", "validate", "derive", "selection_coefficients", "sample_pairs",
           "DEFAULTS", "standard_deviation"]
