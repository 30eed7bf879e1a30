"""CPU double of bench.py's engine (test infrastructure; selected with PANSIM_BENCH_STUB=bench_stub).

bench.py's launcher, process-group plumbing, provider fallback chain, two-line printing and watchdog are host logic
that must work the first time an 8-GPU node runs `python3 bench.py --gpus 8`; this container has no GPU, so the
per-workload measurement is replaced by a canned record (with real barriers and reductions over gloo in it) and the
exchange providers by CPU doubles over gloo.  PANSIM_BENCH_STUB_MODE selects a failure to inject:
  hang_ns   the second (north-star) workload never returns on rank 1      -> the watchdog must print the final line
  raise_ns  the second workload raises on every rank                      -> the final line carries the error
  die_ns    rank 1 dies (os._exit(9)) inside the second workload          -> the launcher relays the contract line
  no_torch  the "torch" provider fails its probe                          -> the chain ends in "none (fallback: ...)"
  hang_nd   the third (configs[4] distance) workload never returns on rank 1 -> the watchdog prints the line with both
            earlier workloads in it
  hang_all  rank 1 hangs before any line exists                           -> the launcher's first-attempt budget ends it
"""
import ctypes
import os
import time

MODE = os.environ.get("PANSIM_BENCH_STUB_MODE", "")


class _GlooOr:
    """ps_exchange_fn double: OR over the default gloo group of the int64 words at a host address"""

    def __init__(self, ctx, broken=False):
        self.ctx, self.broken, self.calls, self.bytes, self.fn, self.error = ctx, broken, 0, 0, None, None

    def __call__(self, ptr, n, stream=0):
        import torch
        if self.broken:
            raise RuntimeError("stub: this provider is broken")
        buf = torch.frombuffer((ctypes.c_int64 * n).from_address(ptr), dtype=torch.int64)
        self.ctx.dist.all_reduce(buf, op=self.ctx.dist.ReduceOp.BOR)
        self.calls += 1
        self.bytes += 16 * n

    def reraise(self, cause=None):
        pass

    def close(self):
        pass


def providers(ctx, rank, world):
    def rccl():
        raise RuntimeError("stub: no RCCL on a CPU box")
    return {"rccl": rccl, "torch": lambda: _GlooOr(ctx, broken=(MODE == "no_torch"))}


def measure(ctx, kw, P, steps, warmup, shard_rank, shard_count, exchange=None, want_pairs=False):
    second = kw["pop_size"] == 65536            # (bench.py's north-star workload)
    third = kw["pop_size"] == 8192              # (... and its configs[4] distance workload)
    if MODE == "hang_all" and ctx.rank == 1 and os.environ.get("PANSIM_BENCH_BACKEND") != "gloo":
        time.sleep(3600)
    if (second and MODE == "hang_ns" or third and MODE == "hang_nd") and ctx.rank == 1:
        time.sleep(3600)
    if second and MODE == "raise_ns":
        raise RuntimeError("stub: the second workload failed")
    if second and MODE == "die_ns" and ctx.rank == 1:
        os._exit(9)
    ctx.barrier()
    t0 = time.perf_counter()
    time.sleep(0.001 * steps)
    ctx.barrier()
    dt = ctx.reduce(time.perf_counter() - t0, "max")
    L_local = kw["core_size"] // shard_count
    ms = 0.5 + 0.01 * ctx.rank
    r = {"dt": dt, "steps": steps, "warmup": warmup, "launches": steps, "sweep_avg_ms": ms,
         "sweep_avg_ms_max_over_ranks": ctx.reduce(ms, "max"), "sweep_avg_ms_min_over_ranks": ctx.reduce(ms, "min"),
         "bytes_per_launch": 2.0 * kw["pop_size"] * L_local, "host": (steps, 0.0, 0.0, 0.0), "settle": [ms], "dist_dt": 0.002,
         "dist_kernel_ms": None, "pair_form": 0, "L_local": L_local, "sweep_form": 1 if kw["pop_size"] <= 1024 else 3,
         "G_acc": kw["pan_genes"] - 2000, "P": P, "N": kw["pop_size"], "kw": kw,
         "exchange": {"mode": exchange or "none (accessory chain replicated on every rank)", "calls": 0,
                      "bytes_sent_plus_received_per_generation": 0.0}}
    if want_pairs:
        r["pairs"] = (None, None)
    return r
