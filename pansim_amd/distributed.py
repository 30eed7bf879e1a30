"""Site-sharded multi-GPU execution (one process per GPU, torch.distributed over RCCL).

SURVEY 8(e): every per-generation core operator is site-local in the site-major layout, so
each rank owns a contiguous range of core sites for ALL individuals.  The accessory matrix is
replicated and every rank draws the same parents from the same seeded stream.  Two exchange steps:
  * per generation (when HGT is on): the HGT donors are sharded over the ranks -- rank r generates the events of
    donors [N r / K, N (r + 1) / K) into a delta bit matrix (33 MB at N = 65536) -- and the deltas are ORed across
    the ranks: an all-to-all of the K row slices, a local OR, and an all-gather of the merged slices over RCCL
    (`TorchExchange`; the north-star's "all-gather of donor tiles").  Events are keyed per donor and ORed into the
    recipient, so the result equals the replicated form bit for bit;
  * the distance phase: the integer Hamming numerators of the sampled pairs are summed over ranks (all-reduce,
    u32), after which every rank applies the reference's `/2` and `/ncols` (population.rs:817-822).
"""
import ctypes as C

import numpy as np


class _DevWords:
    """a device buffer of int64 words as a __cuda_array_interface__ object (zero copy into torch)"""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (int(n_words),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def or_all_reduce(buf, group=None):
    """in-place bitwise OR of the int64 tensor `buf` over the ranks of `group`: all-to-all of the K slices, local OR,
    all-gather of the merged slices (NCCL / RCCL has no OR reduction; gloo has, and takes it on CPU tensors).
    Returns the bytes this rank sent + received."""
    import torch
    import torch.distributed as dist
    K = dist.get_world_size(group)
    if K == 1:
        return 0
    n = buf.numel()
    if dist.get_backend(group) != "nccl":
        host = buf if buf.device.type == "cpu" else buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.BOR, group=group)
        if host is not buf:
            buf.copy_(host)
        return 2 * n * 8 * (K - 1) // K
    part = (n + K - 1) // K
    send = torch.zeros(K * part, dtype=buf.dtype, device=buf.device)
    send[:n] = buf
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)             # slice k of every rank's delta arrives at rank k
    mine = recv[:part].clone()
    for k in range(1, K):
        mine |= recv[k * part:(k + 1) * part]
    dist.all_gather_into_tensor(send, mine, group=group)        # the merged slices, everywhere
    buf.copy_(send[:n])
    return 2 * part * 8 * (K - 1)


class TorchExchange:
    """ps_exchange_fn over torch.distributed: `fn` is the ctypes thunk to hand to Simulation.set_exchange."""

    def __init__(self, group=None):
        from ._lib import EXCHANGE_FN
        self.group = group
        self.calls = 0
        self.bytes = 0
        self.error = None
        self.fn = EXCHANGE_FN(self._call)

    def _call(self, ctx, d_words, n_words, hip_stream):
        try:
            import torch
            stream = torch.cuda.ExternalStream(int(hip_stream or 0)) if hip_stream else torch.cuda.current_stream()
            with torch.cuda.stream(stream):
                buf = torch.as_tensor(_DevWords(d_words, n_words), device="cuda")
                import torch.distributed as dist
                if dist.get_backend(self.group) != "nccl":
                    stream.synchronize()                # the delta is complete before it leaves the device
                self.bytes += or_all_reduce(buf, self.group)
                if dist.get_backend(self.group) != "nccl":
                    stream.synchronize()
            self.calls += 1
            return 0
        except Exception as e:          # never let an exception cross the C boundary
            self.error = e
            return -1


def shard_bounds(core_size, rank, world):
    """[begin, end) of the core sites held by `rank` (same formula as ps_sim_create)."""
    return core_size * rank // world, core_size * (rank + 1) // world


def core_distances_from_counts(counts, core_size):
    """population.rs:817 (`hamming / 2`, integer) and :822 (`as f64 / ncols as f64`)."""
    c = np.asarray(counts).astype(np.uint32)
    return (c // 2).astype(np.float64) / float(core_size)


class ShardedSimulation:
    """One rank of a site-sharded run.  `engine` builds the per-rank simulation; it defaults to
    the HIP-backed pansim_amd.Simulation (tests inject a CPU double to exercise the exchange
    logic over gloo)."""

    def __init__(self, rank, world, engine=None, group=None, shard_hgt_donors=None, **params):
        if engine is None:
            from .simulation import Simulation, make_params

            def engine(**kw):
                return Simulation(make_params(**kw))
        self.rank, self.world, self.group = rank, world, group
        self.core_size = params["core_size"]
        self.bounds = shard_bounds(self.core_size, rank, world)
        self.sim = engine(shard_rank=rank, shard_count=world, **params)
        # the per-generation exchange step: HGT donors sharded over the ranks, deltas ORed across them.  By default only
        # where the replicated HGT is worth more than two small collectives per generation (wide populations: it is the
        # Amdahl term of the N = 65536 run; at N = 1000 the whole HGT hides behind the sweep)
        self.exchange = None
        if shard_hgt_donors is None:
            shard_hgt_donors = params.get("pop_size", 0) >= 4096
        if shard_hgt_donors and world > 1 and hasattr(self.sim, "set_exchange") and world <= params.get("pop_size", world):
            self.exchange = TorchExchange(group)
            self.sim.set_exchange(self.exchange.fn)

    def run(self, count):
        self.sim.run(count)
        if self.exchange is not None and self.exchange.error is not None:
            raise self.exchange.error

    def sync(self):
        self.sim.sync()

    def _all_reduce(self, tensor):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor

    def core_pair_counts(self):
        """global Hamming numerators of the sampled pairs: local kernel, then all-reduce"""
        import torch
        self.sim.sync()
        P = len(self.sim.range1)
        use_cuda = self.world > 1 and torch.distributed.get_backend(self.group) == "nccl"
        if use_cuda or (self.world == 1 and torch.cuda.is_available() and hasattr(self.sim.core_genome, "pairwise_counts_device")):
            t = torch.zeros(P, dtype=torch.int32, device="cuda")
            self.sim.core_genome.pairwise_counts_device(self.sim.range1, self.sim.range2, t.data_ptr())
            torch.cuda.synchronize()
        else:
            (c,) = self.sim.core_genome.pairwise_counts(self.sim.range1, self.sim.range2)
            t = torch.from_numpy(c.astype(np.int32))
        return self._all_reduce(t).cpu().numpy().astype(np.uint32)

    def final_distances(self):
        """main.rs:471-472 for the whole (unsharded) core genome"""
        core = core_distances_from_counts(self.core_pair_counts(), self.core_size)
        P = len(self.sim.range1)
        acc = self.sim.pan_genome.pairwise_distances(P, self.sim.range1, self.sim.range2)
        return core, acc

    def parents_agree(self):
        """every rank must have drawn the same parents (they share the seeded stream)"""
        import torch
        import torch.distributed as dist
        idx = torch.from_numpy(self.sim.last_parents().astype(np.int64))
        if self.world == 1:
            return True
        lo, hi = idx.clone(), idx.clone()
        if dist.get_backend(self.group) == "nccl":
            lo, hi = lo.cuda(), hi.cuda()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        return bool((lo == hi).all().item())

    def close(self):
        self.sim.close()
