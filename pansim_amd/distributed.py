"""Site-sharded multi-GPU execution (one process per GPU, torch.distributed over RCCL).

SURVEY 8(e): every per-generation core operator is site-local in the site-major layout, so
each rank owns a contiguous range of core sites for ALL individuals.  The accessory matrix is
replicated and every rank draws the same parents from the same seeded stream, so a generation
needs NO data-path collective.  The only exchange step is the distance phase: the integer
Hamming numerators of the sampled pairs are summed over ranks (all-reduce, u32), after which
every rank applies the reference's `/2` and `/ncols` (population.rs:817-822).
"""
import numpy as np


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

    def __init__(self, rank, world, engine=None, group=None, **params):
        if engine is None:
            from .simulation import Simulation, make_params

            def engine(**kw):
                return Simulation(make_params(**kw))
        self.rank, self.world, self.group = rank, world, group
        self.core_size = params["core_size"]
        self.bounds = shard_bounds(self.core_size, rank, world)
        self.sim = engine(shard_rank=rank, shard_count=world, **params)

    def run(self, count):
        self.sim.run(count)

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
