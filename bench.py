#!/usr/bin/env python3
"""bench.py -- generations/s of the per-generation hot path on MI355X.

A "step" is one generation of main.rs:429-464 (fitness-weighted parent draw, parent gather of
both matrices, core mutation, accessory gain/loss, HR, HGT) on BASELINE.json configs[1]:
--pop_size 1000 --core_size 1200000 --pan_genes 6000 (defaults otherwise), state resident in HBM.

Multi-GPU (one process per GPU, launched by torch.distributed.run): the core genome is sharded
BY SITE.  Weak scaling: every rank holds 1.2 M sites of a core genome of
n_gpus x 1.2 M sites; the accessory matrix is replicated and every rank draws the same parents
from the same seeded stream, so a generation needs no data-path collective.  `value` counts
1.2 M-site shard-generations per second summed over ranks (at n_gpus = 1: plain generations/s);
`whole_genome_generations_per_s` is the rate of the (n_gpus x larger) simulation itself.
`--scaling strong` (not the driver's contract) keeps --core_size as the whole genome and splits it
over the ranks: BASELINE configs[3] is `--pop_size 65536 --scaling strong`.
`--emulate-shard K` (one GPU): this process runs shard 0 of K -- 1/K of the sites, the whole replicated
accessory chain and the host half of the parent draw -- i.e. what one rank of a K-GPU strong-scaling run does
per generation; `host_half_ms` splits the host side.

Before anything is timed the sweep is run in untimed batches until its per-launch time is stable (clock ramp and
first touch on a cold box), independently of --warmup.

The JSON line also carries `roofline` (fused sweep: algorithmic bytes / HIP-event launch time against
8 TB/s, PMC traffic from profiles/), `distance_roofline` (the sampled-pair distance phase: kernel time by HIP
events), `cpu_baseline` (the reference algorithm restated in C on the host cores: all cores, one thread, and its
distance phase), `mpairs_per_s` / `distance_ms` / `pair_sites_per_s` for the whole distance phase.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def pmc_traffic():
    """HBM bytes per launch of the fused sweep from the committed rocprofv3 --pmc passes
    (profiles/r02_pmc_sweep.json, written by scripts/collect_pmc.py; FETCH_SIZE doubled per the
    gfx950 correction of MI355X_MICROARCH.md).  None if that file is absent."""
    try:
        for name in ("r02_pmc_sweep.json", "r01_pmc_sweep.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                d = json.load(open(path))
                return d["hbm_bytes_per_launch"], d["source"] + " (profiles/%s)" % name
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


def cpu_baseline(kw, seed, pairs, budget_s=20.0):
    """Reference algorithm (event-driven, rows in parallel like rayon) timed on the host cores:
    all cores (the figure `value` reports), one thread (the reference's --threads default), and the
    sampled-pair distance phase on a bounded number of pairs."""
    import numpy as np
    from oracle import oracle as o
    threads = host_cores()
    sim = o.RefSim(o.make_params(**kw), seed=seed, threads=threads)
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
    sim1 = o.RefSim(o.make_params(**kw), seed=seed, threads=1)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pop_size", type=int, default=1000)
    ap.add_argument("--core_size", type=int, default=1200000, help="core sites PER GPU")
    ap.add_argument("--pan_genes", type=int, default=6000)
    ap.add_argument("--HR_rate", type=float, default=0.05)
    ap.add_argument("--HGT_rate", type=float, default=0.05)
    ap.add_argument("--max_distances", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-shard", type=int, default=0, metavar="K",
                    help="one GPU only: run shard 0 of K of a strong-scaling run (1/K of --core_size, full accessory chain)")
    ap.add_argument("--competition_strength", type=float, default=0.0)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the driver's contract): --core_size sites PER GPU; strong: --core_size is the "
                         "whole genome, split by site over the ranks (BASELINE configs[3]: --pop_size 65536 --scaling strong)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import pansim_amd as pa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: pansim_amd has no CPU path")
    # PANSIM_BENCH_BACKEND=gloo lets several ranks share one GPU (debugging the N>1 path on a
    # 1-GPU box); the driver's multi-GPU runs use nccl (= RCCL over xGMI), one rank per GPU
    backend = os.environ.get("PANSIM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    emu = args.emulate_shard
    if emu and world != 1:
        raise SystemExit("--emulate-shard runs on one GPU")
    strong = args.scaling == "strong" or emu > 0
    kw = dict(pop_size=args.pop_size, core_size=args.core_size * (1 if strong else world), pan_genes=args.pan_genes,
              HR_rate=args.HR_rate, HGT_rate=args.HGT_rate)
    if args.competition_strength:
        kw["competition_strength"] = args.competition_strength
    seed = 0
    sim = pa.Simulation(pa.make_params(seed=seed, n_gen=args.steps + args.warmup,
                                       max_distances=args.max_distances, shard_rank=0 if emu else rank,
                                       shard_count=emu if emu else world, device=local_rank, **kw))
    # untimed warm-up until the sweep's per-launch time is stable: a fresh box ramps its clocks and touches
    # its pages during the first few hundred launches (the driver's --warmup 5 is 3 ms of GPU time at cfg2)
    sim.enable_timing(True)
    est_gen_ms = 2.0 * args.pop_size * sim.core_genome.ncols / 4e9            # sweep at ~4 TB/s
    settle, prev, batch = [], None, int(max(5, min(50, 30.0 / max(est_gen_ms, 1e-3))))
    for _ in range(40):
        sim.sweep_timing(reset=True)
        sim.run(batch)
        sim.sync()
        n, ms, _b = sim.sweep_timing(reset=True)
        cur = ms / max(n, 1)
        settle.append(round(cur, 4))
        if prev is not None and abs(cur - prev) <= 0.01 * prev and len(settle) >= 3:
            break
        prev = cur
    sim.run(args.warmup)
    sim.sync()
    sim.sweep_timing(reset=True)
    sim.host_timing(reset=True)
    barrier()
    t0 = time.perf_counter()
    sim.run(args.steps)
    sim.sync()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launches, sweep_ms, bytes_per_launch = sim.sweep_timing(reset=True)
    host_n, host_wait, host_weights, host_draw = sim.host_timing(reset=True)
    sim.enable_timing(False)

    # distance phase (main.rs:467-482): P sampled pairs, core Hamming + accessory Jaccard.
    # Site-sharded: integer partial counts are summed over ranks (RCCL all-reduce).
    P = args.max_distances
    dist_kernel_ms = None
    if world == 1:
        # one process holds every site it will ever hold: the library's own distance phase (both matrices'
        # kernels enqueued together, pinned numerators, main.rs:467-470).  A first call builds the scratch
        # buffers (2-bit copy of the matrix, partial counts); the timed call is the steady state of --print_dist.
        sim.final_distances()
        barrier()
        t1 = time.perf_counter()
        core_d, acc_d = sim.final_distances()
        barrier()
        dist_dt = time.perf_counter() - t1
        dist_kernel_ms = sim.distance_timing()
    else:
        cnt = torch.zeros(P, dtype=torch.int32, device="cuda")
        barrier()
        t1 = time.perf_counter()
        sim.core_genome.pairwise_counts_device(sim.range1, sim.range2, cnt.data_ptr())
        if backend == "nccl":
            dist.all_reduce(cnt)
        else:
            c = cnt.cpu()
            dist.all_reduce(c)
            cnt = c
        acc_d = sim.pan_genome.pairwise_distances(P, sim.range1, sim.range2)
        core_d = (cnt.cpu().numpy().astype("uint32") // 2) / float(kw["core_size"])
        barrier()
        dist_dt = time.perf_counter() - t1
    assert core_d.shape == acc_d.shape

    if rank == 0:
        avg_ms = sweep_ms / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        default_wl = (args.pop_size, args.core_size, args.pan_genes, args.HR_rate, args.HGT_rate, strong, args.competition_strength) == (1000, 1200000, 6000, 0.05, 0.05, False, 0.0)
        L_local = sim.core_genome.ncols
        G_acc = sim.pan_genome.ncols
        # regime of SURVEY 8(d) the core distance kernels ran in: N <= ~5000 -> pairs compared from LDS tiles of the
        # 2-bit packed matrix (ii, integer-VALU bound); wider populations -> the matrix is transposed to 2-bit strings
        # once and two strings are streamed per pair (i, HBM bound), or all-pairs tiles when 2 P > N^2 / 2
        tiled = args.pop_size * 8 * 4 <= 160 * 1024
        # VALU peak of regime (ii): per 16 sites of a pair (one dword of 2-bit codes) the compare issues
        # v_xor, v_lshrrev, v_bitop3 (2.5 cycles each per wave-instruction on a SIMD at full occupancy) and
        # v_bcnt_u32_b32 (4.4) -- scripts/ubench/valu_issue.hip -- i.e. 64 lanes x 16 sites per 11.9 cycles per
        # SIMD, 1024 SIMDs at 2.4 GHz
        valu_peak = 64 * 16 / 11.9 * 1024 * 2.4e9
        bpair = 2.0 * (L_local + (G_acc + 7) // 8)
        droof = None
        if dist_kernel_ms is not None:
            core_ms, acc_ms = dist_kernel_ms
            kms = max(core_ms, 1e-6)
            droof = {"regime": "ii (LDS-tiled with reuse)" if tiled else "i (two bit strings streamed per pair after one transposition)",
                     "kernel_ms": {"core": core_ms, "accessory": acc_ms},
                     "bound": "valu" if tiled else "hbm",
                     "achieved": (P * float(L_local) / (kms * 1e-3)) if tiled else (P * bpair / (kms * 1e-3) / 1e9),
                     "peak": valu_peak if tiled else HBM_PEAK_GBS,
                     "unit": "pair-sites/s" if tiled else "GB/s of the reference's 2 (L + G/8) bytes per pair",
                     "regime_i_equivalent_GBps": P * bpair / (kms * 1e-3) / 1e9}
            droof["frac"] = droof["achieved"] / droof["peak"]
        traffic, traffic_src = pmc_traffic() if default_wl else (None, None)
        out = {
            "metric": "generations/sec", "value": (1 if strong else world) * args.steps / dt,
            "whole_genome_generations_per_s": args.steps / dt, "core_sites_total": kw["core_size"],
            "unit": ("generations/s (pop=%d, %d core sites in all, pan=%d)" if strong else
                     "generations/s (pop=%d, %d core sites per GPU, pan=%d)") % (args.pop_size, args.core_size, args.pan_genes),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: " if default_wl else "BASELINE configs[3]: " if args.pop_size == 65536
                                    else "") + "--pop_size %d --core_size %d --pan_genes %d, seed 0, "
                                   "HR_rate %g HGT_rate %g; core sites sharded %d-way"
                                   % (args.pop_size, kw["core_size"], args.pan_genes, args.HR_rate,
                                      args.HGT_rate, world),
                       "parallelism": "site-shard x%d" % world},
            "mpairs_per_s": P / dist_dt / 1e6,
            "distance_ms": 1e3 * dist_dt,
            # regime (ii), pairs compared from LDS tiles (HBM traffic ~ N*L per 32768 pairs,
            # integer-VALU bound); secondary figure pair-sites/s over the whole distance phase
            "distance_regime": "ii (LDS-tiled with reuse)" if tiled else "i (transposed 2-bit strings streamed per pair) / all-pairs tiles",
            "pair_sites_per_s": P * float(kw["core_size"]) / dist_dt,
            "distance_roofline": droof,
            "host_half_ms": {"generations": host_n, "wait_for_device": host_wait / max(host_n, 1),
                             "softmaxes": host_weights / max(host_n, 1), "parent_draw": host_draw / max(host_n, 1),
                             "note": "per generation, inside ps_sim_run; hidden behind the previous sweep when shorter than it"},
            "settle_sweep_ms": settle,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ("core_sweep_wave_kernel<gather,mutate,HR>" if args.pop_size <= 1024 else
                                    "core_sweep_block_kernel<gather,mutate,HR>"), "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if world == 1 and not args.no_cpu_baseline and not emu:
            out["cpu_baseline"] = cpu_baseline(kw, seed, (sim.range1, sim.range2))
        if emu:
            out["emulated_shards"] = emu
            out["config"]["workload"] += "; THIS LINE: shard 0 of %d emulated on one GPU (%d sites)" % (emu, L_local)
        print(json.dumps(out), flush=True)
    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
