#!/usr/bin/env python3
"""bench.py -- generations/s of the per-generation hot path on MI355X.

A "step" is one generation of main.rs:429-464 (fitness-weighted parent draw, parent gather of
both matrices, core mutation, accessory gain/loss, HR, HGT) on BASELINE.json configs[1]:
--pop_size 1000 --core_size 1200000 --pan_genes 6000 (defaults otherwise), state resident in HBM.

The default line (N = 1) also carries `other_configs`: short runs of the other BASELINE configurations on the same
GPU -- cfg3 (HR = HGT = 0.5), the cfg5 population (N = 8192: sweep + the all-pairs distance phase at P = 2^25),
cfg4 (N = 65536) as one rank of 8 (`--emulate-shard 8`) and whole on one GPU -- each with generations/s, the
sweep's roofline fraction and the distance phase.  `--config NAME` makes one of them the main workload (that is
how the per-config rocprofv3 summaries under profiles/ are collected).

Multi-GPU (one process per GPU, launched by torch.distributed.run): the core genome is sharded
BY SITE.  The driver's contract line is WEAK scaling of cfg2: every rank holds 1.2 M sites of a core
genome of n_gpus x 1.2 M sites; the accessory matrix is replicated and every rank draws the same parents
from the same seeded stream.  `value` is the rate at which the (n_gpus x larger) simulation itself advances, at every
world size -- flat under perfect weak scaling; the whole-job aggregates that grow with the ranks are
`shard_generations_per_s` (= n_gpus x value, 1.2 M-site shard-generations per second summed over the ranks) and
`cell_updates_per_s`.  The same line carries `north_star_scaling`: the
north-star's scaling workload, --pop_size 65536 with the 1.2 M core sites split over the ranks (STRONG scaling),
in whole-simulation generations/s.  `--scaling strong` makes that split the main workload (`value` = the
whole simulation's generations/s).
`--emulate-shard K` (one GPU): this process runs shard 0 of K -- 1/K of the sites, its share of the accessory
chain and the host half of the parent draw -- i.e. what one rank of a K-GPU strong-scaling run does per generation.

Before anything is timed the sweep is run in untimed batches until its per-launch time is stable (clock ramp and
first touch on a cold box), independently of --warmup; with several ranks the decision to stop is shared, so that
every rank runs the same number of generations.

The JSON line also carries `roofline` (fused sweep: algorithmic bytes / HIP-event launch time against
8 TB/s, PMC traffic from profiles/), `distance_roofline` (the distance phase priced against the roofline of the
kernel form that ran), `cpu_baseline` (the reference algorithm restated in C on the host cores: all cores, one
thread, and its distance phase), `mpairs_per_s` / `distance_ms` / `pair_sites_per_s` for the whole distance phase.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
MFMA_I8_PEAK_TOPS = 5000.0   # MI355X_MICROARCH.md, Matrix cores: I8 = 2x BF16 per clock, BF16 ~2.5 PFLOP/s dense
MFMA_FP4_PEAK_TOPS = 10000.0  # the same table: FP4 (block-scaled) = 4x BF16 per clock, ~10 PFLOP/s dense

CONFIGS = {
    # name: (simulation parameters, P, shard emulation, default steps, default warmup, BASELINE label)
    "cfg2": (dict(pop_size=1000, core_size=1200000, pan_genes=6000, HR_rate=0.05, HGT_rate=0.05), 100000, 0, None, None,
             "BASELINE configs[1]"),
    "cfg3": (dict(pop_size=1000, core_size=1200000, pan_genes=6000, HR_rate=0.5, HGT_rate=0.5), 100000, 0, 40, 5,
             "BASELINE configs[2]"),
    "cfg5pop": (dict(pop_size=8192, core_size=1200000, pan_genes=6000, HR_rate=0.05, HGT_rate=0.05), 1 << 25, 0, 10, 2,
                "BASELINE configs[4] (population, P = 2^25 all-pairs distance phase; the matrix writers are not timed)"),
    "cfg4_shard8": (dict(pop_size=65536, core_size=1200000, pan_genes=6000, HR_rate=0.05, HGT_rate=0.05), 100000, 8, 10, 2,
                    "BASELINE configs[3], shard 0 of 8 emulated on one GPU"),
    "cfg4": (dict(pop_size=65536, core_size=1200000, pan_genes=6000, HR_rate=0.05, HGT_rate=0.05), 100000, 0, 5, 1,
             "BASELINE configs[3], all 1.2 M sites on one GPU (78.6 GB of core state)"),
    # not a BASELINE configuration: the one workload the reference's AUTHORS script themselves
    # (scripts/run_pansim_benchmark.sh:2-12, :235-278, the "weak competition" run: one accessory compartment, no
    # recombination, D-avg + the competition softmax every generation, --threads 4)
    "authors": (dict(pop_size=1000, core_size=1342000, pan_genes=4400, core_genes=1342, avg_gene_freq=0.45, core_mu=0.019,
                     HR_rate=0.0, HGT_rate=0.0, rate_genes1=1.0, rate_genes2=1000.0, prop_genes2=0.0, pos_lambda=100.0,
                     neg_lambda=100.0, competition_strength=100.0), 100000, 0, 40, 5,
                "the reference authors' scripted run (scripts/run_pansim_benchmark.sh: --competition_strength 100)"),
}
ORACLE_PARAM_KEYS = ("pop_size", "core_size", "pan_genes", "core_genes", "avg_gene_freq", "HR_rate", "HGT_rate", "core_mu",
                     "rate_genes1", "rate_genes2", "prop_genes2")


def pmc_traffic(kernel, algorithmic_bytes):
    """HBM bytes per launch of the fused sweep from the committed rocprofv3 --pmc passes (profiles/r0N_pmc_*.json, written
    by scripts/collect_pmc.py; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  The passes ran on one
    workload per kernel (the file names it); for another workload of the same kernel the measured traffic / algorithmic
    ratio is applied to this workload's algorithmic bytes, and the source string says so.  None if no file is present."""
    names = {"wave": ("r06_pmc_sweep.json", "r05_pmc_sweep.json", "r04_pmc_sweep.json", "r03_pmc_sweep.json", "r02_pmc_sweep.json", "r01_pmc_sweep.json"),
             "window": ("r06_pmc_window_sweep.json", "r05_pmc_window_sweep.json", "r04_pmc_window_sweep.json", "r03_pmc_window_sweep.json"),
             "block": ("r02_pmc_block_sweep.json", "r01_pmc_block_sweep.json")}.get(kernel, ())
    try:
        for name in names:
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                d = json.load(open(path))
                src = d["source"] + " (profiles/%s, workload %s)" % (name, d.get("workload", "?"))
                alg = d.get("algorithmic_bytes_per_launch")
                if alg and abs(alg - algorithmic_bytes) > 1e-6 * alg:
                    return d["hbm_bytes_per_launch"] / alg * algorithmic_bytes, src + "; scaled by algorithmic bytes to this workload"
                return d["hbm_bytes_per_launch"], src
        return None, None
    except (OSError, KeyError, ValueError):
        return None, None


def host_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(kw, seed, pairs, budget_s=20.0, also_threads=None):
    """Reference algorithm (event-driven, rows in parallel like rayon) timed on the host cores:
    all cores (the figure `value` reports), one thread (the reference's --threads default), and the
    sampled-pair distance phase on a bounded number of pairs.  `also_threads`: one more generation at that thread count
    (the authors' script runs --threads 4) instead of the one-thread run."""
    from oracle import oracle as o
    threads = host_cores()
    comp = float(kw.get("competition_strength", 0.0))
    okw = {k: v for k, v in kw.items() if k in ORACLE_PARAM_KEYS}
    if also_threads:
        sim = o.RefSim(o.make_params(**okw), seed=seed, threads=threads, competition_strength=comp)
        t0 = time.perf_counter()
        sim.generation(0)
        first = time.perf_counter() - t0
        n = max(1, min(8, int((budget_s - first) / max(first, 1e-3))))
        t0 = time.perf_counter()
        for g in range(n):
            sim.generation(1 + g)
        dt = time.perf_counter() - t0
        sim.close()
        simk = o.RefSim(o.make_params(**okw), seed=seed, threads=int(also_threads), competition_strength=comp)
        simk.generation(0)
        t0 = time.perf_counter()
        simk.generation(1)
        dtk = time.perf_counter() - t0
        simk.close()
        what = "pop=%d core=%d pan=%d core_genes=%d, competition_strength %g" % (kw["pop_size"], kw["core_size"], kw["pan_genes"],
                                                                                 kw.get("core_genes", 2000), comp)
        return {"value": n / dt, "unit": "generations/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
                "sample": "%d generations of %s after 1 warm-up generation; event-driven reference algorithm incl. the row-parallel "
                          "average_distance (oracle/pansim_oracle.c orc_ref_*), %d threads" % (n, what, threads),
                "threads%d" % int(also_threads): {"value": 1.0 / dtk, "unit": "generations/s", "cores": int(also_threads),
                                                  "sample": "1 generation after 1 warm-up generation with %d threads, the script's "
                                                            "--threads value" % int(also_threads)}}
    sim = o.RefSim(o.make_params(**okw), seed=seed, threads=threads, competition_strength=comp)
    t0 = time.perf_counter()
    sim.generation(0)                      # warm-up generation (page faults, first touch)
    first = time.perf_counter() - t0
    n = max(1, min(8, int((budget_s - first) / max(first, 1e-3))))
    t0 = time.perf_counter()
    for g in range(n):
        sim.generation(1 + g)
    dt = time.perf_counter() - t0
    # distance phase of the reference (2 rows streamed per pair), all cores, bounded sample
    r1, r2 = pairs
    n_pairs = min(len(r1), 20000)
    t0 = time.perf_counter()
    sim.pairwise(True, r1[:n_pairs], r2[:n_pairs])
    sim.pairwise(False, r1[:n_pairs], r2[:n_pairs])
    dist_dt = time.perf_counter() - t0
    sim.close()
    # --threads 1 is the reference's default (main.rs:127-131): one generation, state already warm
    sim1 = o.RefSim(o.make_params(**okw), seed=seed, threads=1, competition_strength=comp)
    t0 = time.perf_counter()
    sim1.generation(0)
    dt1 = time.perf_counter() - t0
    sim1.close()
    return {"value": n / dt, "unit": "generations/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model(),
            "sample": "%d generations of pop=%d core=%d pan=%d after 1 warm-up generation; "
                      "event-driven reference algorithm (oracle/pansim_oracle.c orc_ref_*), %d threads"
                      % (n, kw["pop_size"], kw["core_size"], kw["pan_genes"], threads),
            "threads1": {"value": 1.0 / dt1, "unit": "generations/s", "cores": 1,
                         "sample": "1 generation (the first, no warm-up) with one thread, the reference's --threads default"},
            "distances": {"value": n_pairs / dist_dt / 1e6, "unit": "Mpairs/s", "cores": threads,
                          "sample": "%d of the run's sampled pairs, core Hamming + accessory Jaccard, 2 rows streamed per pair"
                                    % n_pairs}}


class Ctx:
    """Process-group plumbing shared by every workload of one bench.py invocation.

    Two planes.  CONTROL (barriers, the max / min of a timing over the ranks, the 128-byte RCCL id): always the default
    gloo group on CPU tensors -- a few bytes per call, and it works whatever state the GPU fabric is in.  DATA (the sum of
    the u32 distance numerators, the OR exchange of TorchExchange): an nccl (= RCCL) group when the backend is nccl AND a
    probe collective went through on every rank; otherwise the same collectives run over gloo on host copies and the JSON
    line says so (`data_plane`)."""

    def __init__(self, torch, dist, world, rank, local_rank, backend):
        self.torch, self.dist, self.world, self.rank, self.local_rank, self.backend = torch, dist, world, rank, local_rank, backend
        self.data_group = None                  # None: the default (gloo) group
        self.data_plane = "none (one rank)" if world == 1 else "gloo (host copies)"
        self.rccl_ranks_seen = 0                # ranks that answered the probe collective of the nccl (= RCCL) group
        self.notes = []

    def probe_data_plane(self):
        """create the nccl group and push one tiny all-reduce + one all-to-all through it; every rank must succeed"""
        torch, dist = self.torch, self.dist
        if self.world == 1 or self.backend != "nccl":
            return
        err = None
        try:
            g = dist.new_group(backend="nccl", timeout=_timeout())
            t = torch.full((self.world,), float(self.rank + 1), device="cuda")
            dist.all_reduce(t, group=g)
            o = torch.empty_like(t)
            dist.all_to_all_single(o, t, group=g)
            ones = torch.ones(1, device="cuda")
            dist.all_reduce(ones, group=g)
            torch.cuda.synchronize()
            self.rccl_ranks_seen = int(round(float(ones.item())))
            want = self.world * (self.world + 1) / 2.0
            if abs(float(o[0].item()) - want) > 1e-6:
                raise RuntimeError("probe all-reduce returned %r, expected %r" % (float(o[0].item()), want))
        except Exception as e:          # (RCCL refusing two ranks on one device, a missing IPC mode, ...)
            err = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
        bad = self.reduce(1.0 if err else 0.0, "max")
        if bad == 0.0:
            self.data_group, self.data_plane = g, "nccl (RCCL)"
        else:
            self.data_plane = "gloo (host copies; fallback: the nccl probe failed on %s)" % (
                "this rank: " + err if err else "another rank")
            self.notes.append(self.data_plane)

    def barrier(self):
        if self.world > 1:
            self.dist.all_reduce(self.torch.zeros(1))          # control plane: gloo, CPU
        self.torch.cuda.synchronize()

    def reduce(self, x, op="max"):
        """all-reduce of one float over the ranks (max or min), control plane"""
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.MIN)
        return float(t.item())

    def sum_counts(self, cnt):
        """sum of an int32 device tensor over the ranks (the distance phase's one collective), data plane"""
        if self.world == 1:
            return cnt
        if self.data_group is not None:
            self.dist.all_reduce(cnt, group=self.data_group)
            return cnt
        c = cnt.cpu()
        self.dist.all_reduce(c)
        return c


def _timeout():
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("PANSIM_BENCH_PG_TIMEOUT", "120")))


def _probe_buffer(torch, rank, world, n):
    """a delta-like buffer whose OR over the ranks is known: word w carries bit (rank) and, on its own residue, bit 32 + rank"""
    w = torch.arange(n, dtype=torch.int64)
    mine = (torch.ones(n, dtype=torch.int64) << rank) | torch.where(w % world == rank, torch.ones(n, dtype=torch.int64) << (32 + rank),
                                                                  torch.zeros(n, dtype=torch.int64))
    want = torch.zeros(n, dtype=torch.int64)
    for r in range(world):
        want |= (torch.ones(n, dtype=torch.int64) << r) | torch.where(w % world == r, torch.ones(n, dtype=torch.int64) << (32 + r),
                                                                   torch.zeros(n, dtype=torch.int64))
    return mine, want


def pick_exchange(ctx, shard_rank, shard_count, wanted=None):
    """The provider of the per-generation OR exchange, chosen by a FALLBACK CHAIN: the library's own RCCL provider
    (ps_exchange_rccl) -> torch.distributed (RCCL group if the data plane is nccl, else gloo) -> none (the HGT chain stays
    replicated on every rank).  Each candidate is built and pushed through one small exchange whose result is known
    (lengths the world does not divide); it is taken only if EVERY rank got the right words.  Returns (mode, provider or
    None, chain) where chain records what was tried and why it was dropped -- the JSON line carries it."""
    torch = ctx.torch
    order = [wanted] if wanted and wanted not in ("auto", "") else ["rccl", "torch"]
    chain = []

    def make_rccl():
        if ctx.data_group is None:
            raise RuntimeError("the nccl data plane is not up (%s): a second RCCL communicator would fail the same way" % ctx.data_plane)
        from pansim_amd.distributed import RcclExchange
        return RcclExchange(shard_rank, shard_count, ctx.local_rank)

    def make_torch():
        from pansim_amd.distributed import TorchExchange
        return TorchExchange(group=ctx.data_group, device=ctx.local_rank)

    factories, dev = {"rccl": make_rccl, "torch": make_torch}, "cuda"
    if os.environ.get("PANSIM_BENCH_STUB"):          # CPU test double of the providers (tests/bench_stub.py)
        import importlib
        factories, dev = importlib.import_module(os.environ["PANSIM_BENCH_STUB"]).providers(ctx, shard_rank, shard_count), "cpu"
    for mode in order:
        if mode == "none":
            break
        x, err = None, None
        try:
            if mode not in factories:
                raise ValueError("unknown exchange provider %r" % mode)
            x = factories[mode]()
            n = 1000 + 37                                   # (not a multiple of 2, 3, 4, 8)
            mine, want = _probe_buffer(torch, shard_rank, shard_count, n)
            d = mine.to(dev)
            ctx.torch.cuda.synchronize()
            x(d.data_ptr(), n, 0)
            ctx.torch.cuda.synchronize()
            if not torch.equal(d.cpu(), want):
                raise RuntimeError("the probe exchange returned wrong words")
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
        bad = ctx.reduce(1.0 if err else 0.0, "max")
        if bad == 0.0:
            chain.append({"provider": mode, "ok": True})
            return mode, x, chain
        chain.append({"provider": mode, "ok": False, "error": err or "failed on another rank"})
        if x is not None and hasattr(x, "close"):
            try:
                x.close()
            except Exception:
                pass
    return "none", None, chain


def distance_roofline(form, N, L_local, G_acc, P, core_ms, acc_ms):
    """Prices the core distance kernels' HIP-event time against the roofline of the kernel form that ran
    (include/pansim_hip.h, PS_PAIR_FORM_*); byte and operation counts are the KERNELS' own algorithmic ones."""
    kms = max(core_ms, 1e-6) * 1e-3
    bpair_ref = 2.0 * (L_local + (G_acc + 7) // 8)      # the reference streams two byte rows per pair
    out = {"kernel_ms": {"core": core_ms, "accessory": acc_ms},
           "reference_equivalent_GBps": P * bpair_ref / kms / 1e9,
           "note_reference_equivalent": "P x 2 (L + G/8) bytes of the reference's formulation / kernel time: a rate, not a "
                                        "roofline fraction (the kernels read packed strings)"}
    if form in (1, 3):
        # regime (ii): pairs compared from LDS tiles.  Per 16 sites of a pair (one dword of 2-bit codes) the compare
        # issues v_xor, v_lshrrev, v_bitop3 and v_bcnt: 4 VALU wave-instructions.  Peak at the guide's 2-cycle issue
        # (MI355X_MICROARCH.md, wave scheduling): 64 lanes x 16 sites / 8 cycles per SIMD, 1024 SIMDs, 2.4 GHz;
        # at the issue costs measured by scripts/ubench/valu_issue.hip (2.5 / 2.5 / 2.5 / 4.4 cycles) the same count
        # gives 11.9 cycles.  Nibble form (3): 8 sites per dword, 2 instructions (v_xor, v_bcnt).
        if form == 1:
            peak_guide, peak_meas = 64 * 16 / 8.0 * 1024 * 2.4e9, 64 * 16 / 11.9 * 1024 * 2.4e9
        else:
            peak_guide, peak_meas = 64 * 8 / 4.0 * 1024 * 2.4e9, 64 * 8 / 6.9 * 1024 * 2.4e9
        ach = P * float(L_local) / kms
        out.update({"regime": "ii (sampled pairs from LDS tiles of the packed matrix, with reuse)", "bound": "valu",
                    "achieved": ach, "peak": peak_guide, "unit": "pair-sites/s", "frac": ach / peak_guide,
                    "peak_at_measured_issue_costs": peak_meas, "frac_at_measured_issue_costs": ach / peak_meas})
    elif form == 4:
        # regime (i): one transposition of the matrix to 2-bit strings (read N L bytes, write N L / 4), then two
        # strings of L / 4 bytes streamed per pair
        alg = float(N) * L_local * 1.25 + float(P) * 2.0 * L_local / 4.0
        ach = alg / kms / 1e9
        out.update({"regime": "i (matrix transposed once to 2-bit strings, two strings streamed per pair)", "bound": "hbm",
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "algorithmic_bytes": alg})
    elif form in (6, 7, 8):
        # all pairs on the matrix cores: N (N - 1) / 2 pairs x L sites x 4 one-hot products, 2 operations each.  Form 8 (round 6)
        # gets the same counts from 3 products per site (+-1 features): it is priced on the SAME 4-product operation count, so
        # that `frac` stays comparable across the forms -- `products_per_site_issued` says what the matrix cores really ran
        ops = float(N) * (N - 1) / 2.0 * L_local * 4.0 * 2.0
        ach = ops / kms / 1e12
        peak = MFMA_I8_PEAK_TOPS if form == 6 else MFMA_FP4_PEAK_TOPS
        out.update({"regime": ("all pairs, one-hot X X^T on v_mfma_i32_32x32x32_i8 (exact i32 counts), then lookup" if form == 6 else
                               "all pairs, one-hot X X^T on v_mfma_scale_f32_32x32x64_f8f6f4 (E2M1 {0, 1}, scales 2^0; exact f32 counts), then lookup" if form == 7 else
                               "all pairs, three +-1 features per site on v_mfma_scale_f32_32x32x64_f8f6f4 (E2M1 +-1, scales 2^0; S = 4 matches - sites, exact f32 sums), then lookup"),
                    "products_per_site_issued": 3 if form == 8 else 4,
                    "bound": "mfma-i8" if form == 6 else "mfma-fp4", "achieved": ach, "peak": peak, "unit": "TOP/s", "frac": ach / peak,
                    "pair_sites_per_s": float(N) * (N - 1) / 2.0 * L_local / kms,
                    # the same time priced three ways: `frac` counts the N (N - 1) / 2 distinct pairs; the kernel computes whole
                    # 256 x 256 tiles of the upper triangle (diagonal tiles in both orders, the last tile padded); the run asked
                    # for P sampled pairs
                    "frac_on_computed_tile_pairs": (lambda nt: nt * (nt + 1) / 2.0 * 65536.0 * L_local * 8.0 / kms / 1e12 / peak)((N + 255) // 256),
                    "frac_on_requested_pairs": float(P) * L_local * 8.0 / kms / 1e12 / peak})
    elif form == 2:
        # all pairs, xor + popcount on nibble strings: 2 VALU wave-instructions per 8 sites of a pair
        ach = float(N) * (N - 1) / 2.0 * L_local / kms
        peak_guide, peak_meas = 64 * 8 / 4.0 * 1024 * 2.4e9, 64 * 8 / 6.9 * 1024 * 2.4e9
        out.update({"regime": "all pairs, register-tiled xor + popcount on nibble strings, then lookup", "bound": "valu",
                    "achieved": ach, "peak": peak_guide, "unit": "pair-sites/s (all N (N-1) / 2 pairs)", "frac": ach / peak_guide,
                    "peak_at_measured_issue_costs": peak_meas, "frac_at_measured_issue_costs": ach / peak_meas})
    else:
        out.update({"regime": "one thread per pair on the byte matrix", "bound": "hbm", "achieved": None, "peak": None,
                    "unit": None, "frac": None})
    return out


def _with_chain(r, exchange, chain):
    """records what pick_exchange tried; a run that ended without a provider says why in `mode`"""
    if chain is not None:
        r["exchange"]["provider_chain"] = chain
        failed = [c for c in chain if not c["ok"]]
        if exchange is None and failed:
            r["exchange"]["mode"] = "none (fallback: %s)" % "; ".join("%s: %s" % (c["provider"], c["error"]) for c in failed)
    return r


def measure(ctx, kw, P, steps, warmup, shard_rank, shard_count, seed=0, want_pairs=False, exchange=None, xchg=None, chain=None):
    """One workload: create the simulation (this rank's shard), settle, warm up, time exactly `steps` generations
    between barriers (max over ranks), then the distance phase.  Returns a dict of raw measurements.
    `exchange` names the per-generation exchange step of a strong-scaling run (HGT donors sharded over the ranks, deltas
    ORed across them): "rccl" / "torch" = the provider object `xchg` that pick_exchange built and probed (the caller owns
    it); "emulate" = this process plays shard 0 of shard_count and device-local copies of the same volume stand in for
    the collectives; None / "none" = the accessory chain replicated on every rank."""
    if os.environ.get("PANSIM_BENCH_STUB"):
        import importlib
        if xchg is None and exchange != "emulate":
            exchange = None
        r = importlib.import_module(os.environ["PANSIM_BENCH_STUB"]).measure(ctx, kw, P, steps, warmup, shard_rank, shard_count,
                                                                            exchange=exchange, want_pairs=want_pairs)
        return _with_chain(r, exchange, chain)
    import pansim_amd as pa
    torch, dist, world = ctx.torch, ctx.dist, ctx.world
    sim = pa.Simulation(pa.make_params(seed=seed, n_gen=steps + warmup, max_distances=P, shard_rank=shard_rank,
                                       shard_count=shard_count, device=ctx.local_rank, **kw))
    N = kw["pop_size"]
    if exchange == "emulate":
        sim.emulate_exchange(shard_count)
    elif exchange in ("torch", "rccl") and shard_count > 1 and xchg is not None:
        if exchange == "rccl":
            sim.set_exchange(xchg.fn, xchg.ctx)
        else:
            sim.set_exchange(xchg.fn)
    else:
        exchange, xchg = None, None
    sim.enable_timing(True)
    est_gen_ms = 2.0 * N * sim.core_genome.ncols / 4e9            # sweep at ~4 TB/s
    settle, prev, batch = [], None, int(max(5, min(50, 30.0 / max(est_gen_ms, 1e-3))))
    for _ in range(40):
        sim.sweep_timing(reset=True)
        sim.run(batch)
        sim.sync()
        n, ms, _b = sim.sweep_timing(reset=True)
        cur = ms / max(n, 1)
        settle.append(round(cur, 4))
        stable = prev is not None and abs(cur - prev) <= 0.01 * prev and len(settle) >= 3
        # every rank must leave the loop after the same number of generations (the parents and the accessory
        # mutations are keyed on the generation number): stop only when every rank is stable
        if ctx.reduce(0.0 if stable else 1.0, "max") == 0.0:
            break
        prev = cur
    sim.run(warmup)
    sim.sync()
    sim.sweep_timing(reset=True)
    sim.host_timing(reset=True)
    sim.exchange_stats(reset=True)
    x0 = (xchg.calls, xchg.bytes) if xchg else (0, 0)
    sim.emulated_link_time(reset=True)
    ctx.barrier()
    t0 = time.perf_counter()
    try:
        sim.run(steps)
        sim.sync()
    except Exception as e:          # the library only knows "the exchange failed (-1)": surface the provider's own error
        if xchg is not None:
            xchg.reraise(e)
        raise
    ctx.barrier()
    dt = ctx.reduce(time.perf_counter() - t0, "max")
    if xchg is not None:
        xchg.reraise()
    x_calls, x_bytes = (xchg.calls - x0[0], xchg.bytes - x0[1]) if xchg else sim.exchange_stats(reset=True)
    link_us, link_gbps, link_lat_us = sim.emulated_link_time(reset=True)
    launches, sweep_ms, bytes_per_launch = sim.sweep_timing(reset=True)
    host_n, host_wait, host_weights, host_draw = sim.host_timing(reset=True)
    sim.enable_timing(False)
    sweep_form = sim.core_genome.last_sweep_form()

    # distance phase (main.rs:467-482): P sampled pairs, core Hamming + accessory Jaccard.
    # Site-sharded: integer partial counts are summed over ranks (RCCL all-reduce).
    dist_kernel_ms = None
    if world == 1:
        # one process holds every site it will ever hold: the library's own distance phase (both matrices'
        # kernels enqueued together, pinned numerators, main.rs:467-470).  A first call builds the scratch
        # buffers (2-bit copy of the matrix, partial counts); the timed call is the steady state of --print_dist.
        sim.final_distances()
        ctx.barrier()
        t1 = time.perf_counter()
        core_d, acc_d = sim.final_distances()
        ctx.barrier()
        dist_dt = time.perf_counter() - t1
        dist_kernel_ms = sim.distance_timing()
    else:
        cnt = torch.zeros(P, dtype=torch.int32, device="cuda")
        sim.core_genome.pairwise_counts_device(sim.range1, sim.range2, cnt.data_ptr())      # builds the scratch buffers
        ctx.barrier()
        t1 = time.perf_counter()
        sim.core_genome.pairwise_counts_device(sim.range1, sim.range2, cnt.data_ptr())
        cnt = ctx.sum_counts(cnt)
        acc_d = sim.pan_genome.pairwise_distances(P, sim.range1, sim.range2)
        core_d = (cnt.cpu().numpy().astype("uint32") // 2) / float(kw["core_size"])
        ctx.barrier()
        dist_dt = ctx.reduce(time.perf_counter() - t1, "max")
    assert core_d.shape == acc_d.shape
    avg_ms = sweep_ms / max(launches, 1)
    r = {"dt": dt, "steps": steps, "warmup": warmup, "launches": launches, "sweep_avg_ms": avg_ms,
         "sweep_avg_ms_max_over_ranks": ctx.reduce(avg_ms, "max"), "sweep_avg_ms_min_over_ranks": ctx.reduce(avg_ms, "min"),
         "bytes_per_launch": bytes_per_launch,
         "host": (host_n, host_wait, host_weights, host_draw), "settle": settle, "dist_dt": dist_dt,
         "dist_kernel_ms": dist_kernel_ms, "pair_form": sim.core_genome.last_pair_form(), "L_local": sim.core_genome.ncols,
         "sweep_form": sweep_form,
         "G_acc": sim.pan_genome.ncols, "P": P, "N": N, "kw": kw,
         "exchange": {"mode": exchange or "none (accessory chain replicated on every rank)", "calls": x_calls,
                      "bytes_sent_plus_received_per_generation": x_bytes / max(steps, 1)}}
    _with_chain(r, exchange, chain)
    if exchange == "emulate":
        r["exchange"]["modelled_link_ms_per_generation"] = link_us / 1e3 / max(steps, 1)
        ring = os.environ.get("PANSIM_EMU_RING", "0") not in ("", "0")
        r["exchange"]["link_model"] = (
            "two collectives per generation, both DIRECT all-to-alls (slice k of every rank to rank k; the merged slice to every "
            "peer): one slice of buffer / K on each of the K - 1 point-to-point xGMI links at once, so each is charged %.0f us + "
            "(buffer / K) / %.0f GB/s%s, as a kernel of RCCL's own footprint (256 threads, 264 registers, 19.7 KB LDS) that "
            "holds the accessory stream, beside device-local copies of the same volume"
            % (link_lat_us, link_gbps, "; PANSIM_EMU_RING=1: the second one priced as a ring all-gather, (K - 1) / K x buffer through "
                                       "ONE link (round 4's figure)" if ring else ""))
    if want_pairs:
        r["pairs"] = (sim.range1, sim.range2)
    sim.close()
    return r


SWEEP_FORMS = {
    1: ("wave", "core_sweep_wave_kernel<gather,mutate,HR> (one wave per site row, 4 rows per trip, out of place)"),
    3: ("window", "core_sweep_window_kernel<gather,mutate,HR> (children in ascending parent order, out of place)"),
    4: ("block", "core_sweep_block_kernel<gather,mutate,HR> (whole rows in workgroup-shared LDS, in place)"),
    5: ("inline", "core_sweep_inline_kernel<gather,mutate,HR> (queue-free form)"),
}


def sweep_roofline(r, with_traffic):
    """the sweep's roofline object, labelled and priced from the kernel form the library reports (ps_last_sweep_form)"""
    achieved = r["bytes_per_launch"] / (r["sweep_avg_ms"] * 1e-3) / 1e9 if r["launches"] else 0.0
    kern, label = SWEEP_FORMS.get(r.get("sweep_form", 0), ("unknown", "no core sweep was launched"))
    traffic, traffic_src = pmc_traffic(kern, r["bytes_per_launch"]) if with_traffic else (None, None)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "traffic_measured_in_this_run": False,
            "kernel": label, "avg_launch_ms": r["sweep_avg_ms"],
            "algorithmic_bytes_per_launch": r["bytes_per_launch"]}


def measure_print_dist(ctx, kw, P, gens, warm=10):
    """--print_dist as a workload (main.rs:502-519): after EVERY generation the run's P sampled distances of
    both matrices and their mean / population sigma -- what turns the distance phase into a per-generation cost.  One
    generation at a time, as the CLI does it; returns generations/s and the distance phase's share."""
    import numpy as np
    import pansim_amd as pa
    sim = pa.Simulation(pa.make_params(seed=0, n_gen=gens + warm, max_distances=P, device=ctx.local_rank, print_dist=1, **kw))
    for _ in range(warm):
        sim.run(1)
        sim.final_distances()
    ctx.barrier()
    t0 = time.perf_counter()
    t_dist, per_gen = 0.0, []
    for _ in range(gens):
        sim.run(1)
        t1 = time.perf_counter()
        cd, ad = sim.final_distances()
        per_gen.append((float(cd.mean()), float(cd.std()), float(ad.mean()), float(ad.std())))
        t_dist += time.perf_counter() - t1
    ctx.barrier()
    dt = time.perf_counter() - t0
    core_ms, acc_ms = sim.distance_timing()
    form = sim.core_genome.last_pair_form()
    sim.close()
    return {"workload": "BASELINE configs[1] + --print_dist: --pop_size %d --core_size %d --pan_genes %d, P = %d sampled distances and their "
                        "mean / sigma after every generation (main.rs:502-519)" % (kw["pop_size"], kw["core_size"], kw["pan_genes"], P),
            "generations_per_s": gens / dt, "ms_per_generation": 1e3 * dt / gens, "steps": gens, "warmup": warm,
            "distance_phase_ms_per_generation_host_clock": 1e3 * t_dist / gens,
            "distance_kernel_ms_last_generation": {"core": core_ms, "accessory": acc_ms}, "pair_form": form,
            "last_generation_mean_core_distance": per_gen[-1][0]}


def host_half(r):
    host_n, host_wait, host_weights, host_draw = r["host"]
    return {"generations": host_n, "wait_for_device": host_wait / max(host_n, 1),
            "softmaxes": host_weights / max(host_n, 1), "parent_draw": host_draw / max(host_n, 1),
            "note": "per generation, inside ps_sim_run; hidden behind the previous sweep when shorter than it"}


def summary(r, label, emu=0):
    """compact record of one workload (the entries of `other_configs` / `north_star_scaling`)"""
    roof = sweep_roofline(r, False)
    period = 1e3 * r["dt"] / r["steps"]
    out = {"workload": label, "generations_per_s": r["steps"] / r["dt"], "ms_per_generation": period, "steps": r["steps"],
           "warmup": r["warmup"], "core_sites_this_process": r["L_local"],
           "sweep": {"kernel": roof["kernel"], "avg_launch_ms": roof["avg_launch_ms"], "achieved_GBps": roof["achieved"],
                     "frac": roof["frac"], "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"]},
           "exposed_non_sweep_ms": period - roof["avg_launch_ms"],
           "distance_ms": 1e3 * r["dist_dt"], "pairs": r["P"], "mpairs_per_s": r["P"] / r["dist_dt"] / 1e6,
           "host_half_ms": host_half(r), "settle_sweep_ms": r["settle"], "exchange": r["exchange"]}
    if r["dist_kernel_ms"] is not None:
        out["distance_roofline"] = distance_roofline(r["pair_form"], r["N"], r["L_local"], r["G_acc"], r["P"], *r["dist_kernel_ms"])
    if emu:
        out["emulated_shards"] = emu
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="make this BASELINE configuration the main workload (default: cfg2, the metric's own)")
    ap.add_argument("--pop_size", type=int, default=None)
    ap.add_argument("--core_size", type=int, default=None, help="core sites PER GPU (weak) / in all (strong)")
    ap.add_argument("--pan_genes", type=int, default=None)
    ap.add_argument("--HR_rate", type=float, default=None)
    ap.add_argument("--HGT_rate", type=float, default=None)
    ap.add_argument("--max_distances", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other BASELINE configurations")
    ap.add_argument("--emulate-shard", type=int, default=None, metavar="K",
                    help="one GPU only: run shard 0 of K of a strong-scaling run (1/K of --core_size, its share of the accessory chain)")
    ap.add_argument("--competition_strength", type=float, default=0.0)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the driver's contract): --core_size sites PER GPU; strong: --core_size is the "
                         "whole genome, split by site over the ranks (BASELINE configs[3]: --pop_size 65536 --scaling strong)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the bare form `python3 bench.py --gpus N`: this process starts the N ranks itself -- BEFORE anything here has
        # imported torch or touched HIP (a process that has initialised the GPU must never exec or fork GPU children)
        raise SystemExit(launch(args, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start it as `python3 bench.py --gpus %d` (it launches its own ranks) or "
                         "under torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus, args.gpus))
    rank_main(args)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args, argv):
    """Parent of the bare multi-GPU form: starts `torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
    process group (never an exec), relays the ranks' stderr, and prints exactly ONE JSON line: the last complete line
    rank 0 produced (the final one; or, if the run died or hung in its second workload, the contract line rank 0 had
    already printed).  A run that fails before any line exists is retried once with the gloo data plane (host copies):
    a slower exchange, but a measured curve instead of a launch error.  Returns the exit code."""
    import signal
    import subprocess
    import threading
    # budgets that fit the driver's 1800-s limit for one bench.py call whatever happens: first attempt <= 700 s, the gloo
    # retry <= 500 s, and inside a run each of the two north-star workloads under a 300-s watchdog (rank_main)
    budgets = [float(os.environ.get("PANSIM_BENCH_LAUNCH_TIMEOUT", "700")), float(os.environ.get("PANSIM_BENCH_RETRY_TIMEOUT", "500"))]
    attempts = []
    backends = [os.environ.get("PANSIM_BENCH_BACKEND", "nccl")]
    if backends[0] != "gloo":
        backends.append("gloo")
    for attempt_no, backend in enumerate(backends):
        budget = budgets[min(attempt_no, 1)]
        env = dict(os.environ)
        env["PANSIM_BENCH_BACKEND"] = backend
        env["PANSIM_BENCH_LAUNCHED"] = "1"
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
        t0 = time.perf_counter()
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True, start_new_session=True)
        lines = []

        def pump():
            for line in proc.stdout:
                line = line.strip()
                if line.startswith("{") and line.endswith("}"):
                    try:
                        lines.append(json.loads(line))
                    except ValueError:
                        pass
                elif line:
                    print(line, file=sys.stderr, flush=True)

        th = threading.Thread(target=pump, daemon=True)
        th.start()
        timed_out = False
        try:
            rc = proc.wait(timeout=budget)
        except subprocess.TimeoutExpired:
            timed_out = True
            try:
                os.killpg(proc.pid, signal.SIGTERM)         # (the exact process group this function started)
                rc = proc.wait(timeout=20)
            except (subprocess.TimeoutExpired, ProcessLookupError):
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                rc = proc.wait()
        th.join(timeout=10)
        attempts.append({"backend": backend, "rc": rc, "timed_out": timed_out, "wall_s": round(time.perf_counter() - t0, 1),
                         "json_lines": len(lines)})
        if lines:
            out = lines[-1]
            out["launcher"] = {"self_launched": True, "attempts": attempts,
                               "note": "python3 bench.py --gpus N started its own ranks (torch.distributed.run as a child process)"}
            if rc != 0 or timed_out:
                out["launcher"]["note"] += "; the ranks ended abnormally after this line was printed"
            print(json.dumps(out), flush=True)
            # the line stands (a parser of stdout is served), the exit code tells CI that something behind it went wrong
            return 0 if rc == 0 and not timed_out else 4
        print("bench.py launcher: %d ranks over %s produced no line (rc %s%s)" % (args.gpus, backend, rc, ", timed out" if timed_out else ""),
              file=sys.stderr, flush=True)
    return 1


class Watchdog:
    """A rank that sits in a collective its peers never enter cannot be interrupted from Python.  Armed around the
    optional second workload: when it fires, rank 0 prints the line it was given (the contract line, already complete)
    with the reason, and every rank leaves with os._exit(3) -- so a hang costs the extras, never the measurement, and the
    non-zero exit code says that a workload was cut short (the launcher passes it on as its own 4)."""

    def __init__(self, seconds, rank, line_fn):
        import threading
        self.t = threading.Timer(seconds, self.fire)
        self.t.daemon = True
        self.rank, self.line_fn, self.seconds = rank, line_fn, seconds

    def fire(self):
        try:
            if self.rank == 0:
                print(json.dumps(self.line_fn("timed out after %.0f s (watchdog)" % self.seconds)), flush=True)
        finally:
            os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def rank_main(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = bool(os.environ.get("PANSIM_BENCH_STUB"))
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: pansim_amd has no CPU path")
    # PANSIM_BENCH_BACKEND=gloo runs every collective over gloo on host copies (several ranks sharing one GPU on a
    # 1-GPU box; the launcher's retry).  The driver's multi-GPU runs use nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("PANSIM_BENCH_BACKEND", "nccl")
    ndev = 1 if stub else torch.cuda.device_count()
    notes = []
    if world > ndev:
        notes.append("%d ranks on %d GPU(s): ranks share devices (rank %% %d), RCCL cannot be used" % (world, ndev, ndev))
        backend = "gloo"
    local_rank = local_rank % ndev
    if not stub:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=_timeout())          # the control plane (see Ctx)
    ctx = Ctx(torch, dist, world, rank, local_rank, backend)
    ctx.notes = notes
    if stub:
        torch.cuda.synchronize = lambda *a, **k: None
    else:
        ctx.probe_data_plane()

    # ---- the main workload
    base_kw, base_P, base_emu, d_steps, d_warmup, label = CONFIGS[args.config or "cfg2"]
    kw = dict(base_kw)
    for k in ("pop_size", "core_size", "pan_genes", "HR_rate", "HGT_rate"):
        if getattr(args, k) is not None:
            kw[k] = getattr(args, k)
    if args.competition_strength == 0.0 and "competition_strength" in kw:
        args.competition_strength = kw["competition_strength"]
    P = args.max_distances if args.max_distances is not None else base_P
    emu = args.emulate_shard if args.emulate_shard is not None else base_emu
    steps = args.steps if args.steps is not None else (d_steps or 100)
    warmup = args.warmup if args.warmup is not None else (d_warmup if d_warmup is not None else 10)
    if emu and world != 1:
        raise SystemExit("--emulate-shard runs on one GPU")
    strong = args.scaling == "strong" or emu > 0
    default_wl = (args.config in (None, "cfg2") and kw == CONFIGS["cfg2"][0] and P == CONFIGS["cfg2"][1] and not strong
                  and args.competition_strength == 0.0)
    if not default_wl and kw == base_kw and args.config:
        wl_label = label
    elif default_wl:
        wl_label = CONFIGS["cfg2"][5]
    else:
        wl_label = "BASELINE configs[3]" if kw["pop_size"] == 65536 else "custom"
    core_per_gpu = kw["core_size"]
    kw["core_size"] = core_per_gpu * (1 if strong else world)
    if args.competition_strength:
        kw["competition_strength"] = args.competition_strength
    # HGT donors are sharded (with the per-generation exchange) in strong-scaling runs of wide populations, where the
    # replicated accessory chain is the Amdahl term; the weak cfg2 contract line keeps it replicated (it hides behind the sweep)
    xmode, xchg, chain = None, None, None
    if strong and kw["pop_size"] >= 4096:
        if emu:
            xmode = "emulate"
        elif world > 1:
            xmode, xchg, chain = pick_exchange(ctx, rank, world, os.environ.get("PANSIM_BENCH_EXCHANGE"))
    r = measure(ctx, kw, P, steps, warmup, 0 if emu else rank, emu if emu else world, want_pairs=True, exchange=xmode, xchg=xchg,
                chain=chain)

    out = None
    if rank == 0:
        L_local, G_acc, N = r["L_local"], r["G_acc"], kw["pop_size"]
        droof = None
        if r["dist_kernel_ms"] is not None:
            droof = distance_roofline(r["pair_form"], N, L_local, G_acc, P, *r["dist_kernel_ms"])
        roof = sweep_roofline(r, default_wl or args.config in ("cfg3", "cfg4", "cfg4_shard8", "cfg5pop", "authors"))
        rate = steps / r["dt"]
        out = {
            # `value` is the rate at which the simulation itself advances, at every world size (a sum of
            # shard-generations read as "generations/s" of a simulation that advances N times slower).  Under weak scaling
            # the simulated genome grows with the ranks, so a flat `value` IS perfect scaling; the whole-job aggregates that
            # grow with the ranks are `shard_generations_per_s` (= n_gpus x value) and `cell_updates_per_s`
            "metric": "generations/sec", "value": rate,
            # schema 2 (round 4 on): with n_gpus > 1 `value` is the whole simulation's rate; the lines of rounds 1-3 summed
            # shard-generations over the ranks there (= today's shard_generations_per_s), so compare like with like
            "schema_version": 2,
            "value_semantics": "generations/s of the WHOLE simulation (all ranks advance one generation together)",
            "shard_generations_per_s": world * rate if not strong else None,
            "cell_updates_per_s": rate * float(kw["pop_size"]) * float(kw["core_size"]),
            "whole_genome_generations_per_s": rate, "core_sites_total": kw["core_size"],
            "unit": ("generations/s of the whole simulation (pop=%d, %d core sites in all, split over %d GPU(s), pan=%d)"
                     % (kw["pop_size"], kw["core_size"], world, kw["pan_genes"])) if strong else
                    ("generations/s (pop=%d, %d core sites, pan=%d)" % (kw["pop_size"], core_per_gpu, kw["pan_genes"])) if world == 1 else
                    ("generations/s of the whole simulation (weak scaling: pop=%d, %d core sites PER GPU = %d in all over %d ranks, "
                     "pan=%d); the ranks together complete shard_generations_per_s = n_gpus x value 1.2 M-site shard-generations/s"
                     % (kw["pop_size"], core_per_gpu, kw["core_size"], world, kw["pan_genes"])),
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * r["dt"] / steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: --pop_size %d --core_size %d --pan_genes %d, seed 0, HR_rate %g HGT_rate %g; core sites "
                                   "sharded %d-way" % (wl_label, kw["pop_size"], kw["core_size"], kw["pan_genes"], kw["HR_rate"],
                                                       kw["HGT_rate"], world),
                       "parallelism": "site-shard x%d" % world},
            "mpairs_per_s": P / r["dist_dt"] / 1e6,
            "distance_ms": 1e3 * r["dist_dt"],
            "pair_sites_per_s": P * float(L_local if emu else kw["core_size"]) / r["dist_dt"],
            "distance_roofline": droof,
            "host_half_ms": host_half(r),
            "settle_sweep_ms": r["settle"],
            "exchange": r["exchange"],
            "roofline": roof,
            "data_plane": ctx.data_plane, "control_plane": "gloo" if world > 1 else "none (one rank)",
            # ranks that answered the probe collective of the RCCL group (0: no such group -- one rank, or the gloo data plane)
            "rccl_ranks_seen": ctx.rccl_ranks_seen,
        }
        if ctx.notes:
            out["notes"] = list(ctx.notes)
        if world > 1:
            out["sweep_avg_ms_over_ranks"] = {"min": r["sweep_avg_ms_min_over_ranks"], "max": r["sweep_avg_ms_max_over_ranks"]}
        if world == 1 and not args.no_cpu_baseline and not emu:
            out["cpu_baseline"] = cpu_baseline(kw, 0, r["pairs"], budget_s=20.0 if kw["pop_size"] <= 1000 else 30.0,
                                               also_threads=4 if args.config == "authors" else None)
        if emu:
            out["emulated_shards"] = emu
            out["config"]["workload"] += "; THIS LINE: shard 0 of %d emulated on one GPU (%d sites)" % (emu, L_local)

    # ---- the other BASELINE configurations, short runs on the same GPU (N = 1, default workload only)
    if default_wl and world == 1 and not args.no_other_configs:
        others = {}
        for name in ("cfg3", "authors", "cfg5pop", "cfg4_shard8", "cfg4"):
            okw, oP, oemu, osteps, owarm, olabel = CONFIGS[name]
            try:
                t0 = time.perf_counter()
                ro = measure(ctx, dict(okw), oP, osteps, owarm, 0, oemu if oemu else 1, exchange="emulate" if oemu else None)
                others[name] = summary(ro, olabel + ": --pop_size %d --core_size %d --pan_genes %d --HR_rate %g --HGT_rate %g, P = %d"
                                       % (okw["pop_size"], okw["core_size"], okw["pan_genes"], okw["HR_rate"], okw["HGT_rate"], oP),
                                       emu=oemu)
                if name == "authors":
                    others[name]["workload"] += ("; --core_genes 1342 --avg_gene_freq 0.45 --core_mu 0.019 --prop_genes2 0.0 "
                                                 "--pos_lambda 100 --neg_lambda 100 --competition_strength 100")
                    if not args.no_cpu_baseline:
                        others[name]["cpu_baseline"] = cpu_baseline(dict(okw), 0, None, budget_s=8.0, also_threads=4)
                others[name]["wall_s_incl_setup"] = time.perf_counter() - t0
            except Exception as e:      # (a configuration that does not fit this GPU must not void the contract line)
                others[name] = {"error": str(e)[:300]}
        try:
            others["cfg2_print_dist"] = measure_print_dist(ctx, dict(CONFIGS["cfg2"][0]), CONFIGS["cfg2"][1], 100)
        except Exception as e:
            others["cfg2_print_dist"] = {"error": str(e)[:300]}
        out["other_configs"] = others

    # ---- the north-star's scaling workload at this world size: --pop_size 65536, 1.2 M core sites split over the ranks.
    # The contract line is complete at this point: rank 0 prints it NOW, so that whatever happens in the second workload
    # (a hang inside a collective, a kill from outside) a parseable line exists; the same line, extended, is printed again
    # at the end (`line`: "final").  The bare form's launcher relays only the last one.
    if default_wl and world > 1:
        okw, oP, _e, osteps, owarm, olabel = CONFIGS["cfg4"]
        if os.environ.get("PANSIM_BENCH_NS_CORE_SIZE"):      # (smoke runs of the launch path: a shorter genome, labelled as such)
            okw = dict(okw, core_size=int(os.environ["PANSIM_BENCH_NS_CORE_SIZE"]))
        if rank == 0:
            print(json.dumps(dict(out, line="contract (north_star_scaling still running)")), flush=True)

        def on_timeout(why):
            return dict(out, line="final", north_star_scaling={"error": why}, north_star_generations_per_s=None)

        ro, ns_chain = None, None
        with Watchdog(float(os.environ.get("PANSIM_BENCH_NS_TIMEOUT", "300")), rank, on_timeout):
            err = None
            try:
                if xchg is None:
                    xmode, xchg, ns_chain = pick_exchange(ctx, rank, world, os.environ.get("PANSIM_BENCH_EXCHANGE"))
                ro = measure(ctx, dict(okw), oP, 10, 2, rank, world, exchange=xmode, xchg=xchg, chain=ns_chain)
            except Exception as e:       # (the contract line above must survive a failure of the second workload)
                err = "%s: %s" % (type(e).__name__, str(e)[:300])
            if err is not None and rank == 0:
                out["north_star_scaling"] = {"error": err, "provider_chain": ns_chain}
                out["north_star_generations_per_s"] = None
        if rank == 0 and ro is not None:
            ns = summary(ro, "BASELINE configs[3] / north_star scaling: --pop_size 65536, %d core sites split over %d ranks (strong "
                             "scaling), P = %d" % (okw["core_size"], world, oP))
            ns["n_gpus"] = world
            ns["scaling"] = "strong"
            ns["unit"] = "generations/s of the whole simulation"
            ns["sweep"]["avg_launch_ms_over_ranks"] = {"min": ro["sweep_avg_ms_min_over_ranks"], "max": ro["sweep_avg_ms_max_over_ranks"]}
            ns["collective_bytes_per_generation"] = ro["exchange"]["bytes_sent_plus_received_per_generation"]
            # the same figures where a parser of the top level sees them
            out["north_star_generations_per_s"] = ns["generations_per_s"]
            out["north_star_sweep_frac_per_rank"] = {
                "min": ro["bytes_per_launch"] / (ro["sweep_avg_ms_max_over_ranks"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "max": ro["bytes_per_launch"] / (ro["sweep_avg_ms_min_over_ranks"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            out["north_star_exposed_non_sweep_ms"] = ns["exposed_non_sweep_ms"]
            out["north_star_collective_bytes_per_generation"] = ns["collective_bytes_per_generation"]
            out["north_star_exchange"] = ro["exchange"]["mode"]
            out["north_star_exchange_world_size"] = getattr(xchg, "world", world) if xchg is not None else 0
            ns["collectives"] = ("per generation: HGT donors sharded over the ranks, the delta bit matrices (N x G bits) ORed with one "
                                 "all-to-all + one all-gather (bytes above: sent + received per rank); every rank draws the same "
                                 "parents; distance phase: one all-reduce of %d u32 numerators" % oP)
            out["north_star_scaling"] = ns

        # ---- BASELINE configs[4] at this world size: --pop_size 8192 --max_distances 33554432, all 1.2 M core sites split over
        # the ranks.  Distance phase as SURVEY 8(e) puts it: every rank counts the mismatches of ALL requested pairs over ITS
        # sites (the all-pairs matrix-core form + lookup), ONE all-reduce (sum, u32) of the P = 2^25 partial counts -- 134 MB --
        # over the data plane, the accessory Jaccard from the replicated matrix (population.rs:787-837).  Under its own
        # watchdog; the line printed so far stands whatever happens here.
        if rank == 0:
            print(json.dumps(dict(out, line="contract + north_star_scaling (north_star_distance still running)")), flush=True)
        dkw, dP, _e, _s, _w, dlabel = CONFIGS["cfg5pop"]
        if os.environ.get("PANSIM_BENCH_NS_CORE_SIZE"):
            dkw = dict(dkw, core_size=int(os.environ["PANSIM_BENCH_NS_CORE_SIZE"]))
        if os.environ.get("PANSIM_BENCH_NS_PAIRS"):          # (smoke runs of the launch path)
            dP = int(os.environ["PANSIM_BENCH_NS_PAIRS"])

        def on_timeout_d(why):
            return dict(out, line="final", north_star_distance={"error": why})

        rd = None
        with Watchdog(float(os.environ.get("PANSIM_BENCH_NS_TIMEOUT", "300")), rank, on_timeout_d):
            err = None
            try:
                rd = measure(ctx, dict(dkw), dP, 5, 1, rank, world)
            except Exception as e:
                err = "%s: %s" % (type(e).__name__, str(e)[:300])
            if err is not None and rank == 0:
                out["north_star_distance"] = {"error": err}
        if rank == 0 and rd is not None:
            nd = summary(rd, "BASELINE configs[4] / north_star distance phase: --pop_size 8192 --max_distances %d, %d core sites split "
                             "over %d ranks" % (dP, dkw["core_size"], world))
            nd["n_gpus"] = world
            nd["collective"] = "one all-reduce (sum, u32) of the %d partial Hamming numerators per distance phase over the data plane (%s)" % (dP, ctx.data_plane)
            nd["collective_bytes_per_rank"] = 4 * dP
            nd["pair_form"] = rd["pair_form"]
            out["north_star_distance"] = nd
            out["north_star_distance_mpairs_per_s"] = nd["mpairs_per_s"]

    if rank == 0:
        out["line"] = "final"
        print(json.dumps(out), flush=True)
    if xchg is not None and hasattr(xchg, "close"):
        try:
            xchg.close()
        except Exception:
            pass
    if world > 1:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
