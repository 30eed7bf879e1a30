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


class OrScratch:
    """the three buffers of the all-to-all form of or_all_reduce, allocated once per (length, world, device): `send`
    (K slices of `part` words: the padded delta, later the gathered result), `recv` (slice `rank` of every rank's delta)
    and `mine` (their OR).  The pad words behind the delta are zeroed here and stay zero: the all-gather writes the OR of
    zeros over them."""

    def __init__(self, n, K, like):
        import torch
        self.n, self.K, self.part = int(n), int(K), (int(n) + int(K) - 1) // int(K)
        self.send = torch.zeros(self.K * self.part, dtype=like.dtype, device=like.device)
        self.recv = torch.empty_like(self.send)
        self.mine = torch.empty(self.part, dtype=like.dtype, device=like.device)

    def fits(self, buf, K):
        return self.n == buf.numel() and self.K == K and self.send.device == buf.device and self.send.dtype == buf.dtype


def or_all_reduce(buf, group=None, force_a2a=False, scratch=None, ring_gather=False):
    """in-place bitwise OR of the int64 tensor `buf` over the ranks of `group`.  NCCL / RCCL has no OR reduction: an
    all-to-all of the K slices (slice k of every rank's buffer arrives at rank k), a local OR, and the merged slices
    sent to everyone (a second direct all-to-all; `ring_gather`: all_gather_into_tensor instead).  gloo has ReduceOp.BOR and takes it on CPU tensors unless `force_a2a` -- which runs the very same
    slice / pad / OR / all-gather logic over gloo, so that the RCCL branch is covered by CPU tests.  `scratch`
    (an OrScratch) keeps the three work buffers between calls.  Returns the bytes this rank sent + received."""
    import torch.distributed as dist
    K = dist.get_world_size(group)
    if K == 1:
        return 0
    n = buf.numel()
    backend = dist.get_backend(group)
    if backend != "nccl" and not force_a2a:
        host = buf if buf.device.type == "cpu" else buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.BOR, group=group)
        if host is not buf:
            buf.copy_(host)
        return 2 * n * 8 * (K - 1) // K
    if scratch is None or not scratch.fits(buf, K):
        scratch = OrScratch(n, K, buf)
    part, send, recv, mine = scratch.part, scratch.send, scratch.recv, scratch.mine
    send[:n].copy_(buf)                                         # (words n .. K * part stay zero)
    dist.all_to_all_single(recv, send, group=group)             # slice k of every rank's delta arrives at rank k
    mine.copy_(recv[:part])
    for k in range(1, K):
        mine |= recv[k * part:(k + 1) * part]
    if ring_gather:
        dist.all_gather_into_tensor(send, mine, group=group)    # the merged slices, everywhere (RCCL: a ring, one link's rate)
    else:
        # the same as a second DIRECT all-to-all: this rank's merged slice to every peer over that peer's own xGMI link,
        # all links at once (recv is free after the OR: it carries K copies of `mine`)
        recv.view(K, part).copy_(mine)
        dist.all_to_all_single(send, recv, group=group)
    buf.copy_(send[:n])
    return 2 * part * 8 * (K - 1)


class TorchExchange:
    """ps_exchange_fn over torch.distributed: `fn` is the ctypes thunk to hand to Simulation.set_exchange.  `device` is
    the HIP ordinal of the simulation whose delta buffer the calls will carry (None: torch's current device).  A Python
    exception inside a call is kept in `error` (and the call returns -1, which fails ps_sim_run); `reraise()` surfaces it."""

    def __init__(self, group=None, device=None, force_a2a=False):
        from ._lib import EXCHANGE_FN
        self.group = group
        self.device = device
        self.force_a2a = force_a2a
        self.scratch = {}           # one OrScratch per buffer length (the HGT deltas; the N average distances of D-avg)
        self.calls = 0
        self.bytes = 0
        self.error = None
        self.fn = EXCHANGE_FN(self._call)

    def _call(self, ctx, d_words, n_words, hip_stream):
        try:
            import torch
            import torch.distributed as dist
            dev = torch.device("cuda", torch.cuda.current_device() if self.device is None or self.device < 0 else int(self.device))
            with torch.cuda.device(dev):
                stream = torch.cuda.ExternalStream(int(hip_stream), device=dev) if hip_stream else torch.cuda.current_stream(dev)
                with torch.cuda.stream(stream):
                    buf = torch.as_tensor(_DevWords(d_words, n_words), device=dev)
                    K = dist.get_world_size(self.group)
                    on_device = dist.get_backend(self.group) == "nccl"
                    if not on_device:
                        stream.synchronize()                # the delta is complete before it leaves the device
                        host = buf.cpu()
                        sc = self.scratch.get(int(n_words))
                        if self.force_a2a and (sc is None or not sc.fits(host, K)):
                            sc = self.scratch[int(n_words)] = OrScratch(n_words, K, host)
                        self.bytes += or_all_reduce(host, self.group, self.force_a2a, sc)
                        buf.copy_(host)
                        stream.synchronize()
                    else:
                        sc = self.scratch.get(int(n_words))
                        if sc is None or not sc.fits(buf, K):
                            sc = self.scratch[int(n_words)] = OrScratch(n_words, K, buf)   # (allocated once: 3 x 33 MB at N = 65536)
                        self.bytes += or_all_reduce(buf, self.group, False, sc)
            self.calls += 1
            return 0
        except Exception as e:          # never let an exception cross the C boundary
            self.error = e
            return -1

    def reraise(self, cause=None):
        """raise the Python exception a failed call kept (chained to the library's error), if any"""
        if self.error is not None:
            err, self.error = self.error, None
            raise err from cause

    def __call__(self, d_words, n_words, hip_stream=0):
        """one exchange on a raw device buffer (tests, bench.py's provider probe); raises what the call kept"""
        if self._call(None, int(d_words), int(n_words), int(hip_stream)) != 0:
            self.reraise()

    def close(self):
        self.scratch = {}


class RcclExchange:
    """The library's own ps_exchange_fn over RCCL (ps_rccl_*: no Python and no torch inside the per-generation call).
    torch.distributed is used ONCE, to carry rank 0's 128-byte communicator id to the other ranks; a host without torch
    carries it any other way.  `fn` / `ctx` are what Simulation.set_exchange takes."""

    def __init__(self, rank, world, device, group=None):
        import torch
        import torch.distributed as dist
        from . import _lib
        self._lib = _lib.load()
        # rank 0's id travels with a status byte: if rank 0 cannot produce one (no librccl), EVERY rank raises instead of
        # the others waiting in the broadcast for a rank that has already left
        msg = np.zeros(129, np.uint8)
        first_error = None
        if rank == 0:
            try:
                ident = np.zeros(128, np.uint8)
                _lib.check(self._lib.ps_rccl_unique_id(ident))
                msg[0], msg[1:] = 1, ident
            except Exception as e:
                first_error = e
        if world > 1:
            t = torch.from_numpy(msg)
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0, group=group)
            msg = t.cpu().numpy().copy()
        if msg[0] != 1:
            raise RuntimeError("rank 0 could not create an RCCL communicator id") from first_error
        ident = np.ascontiguousarray(msg[1:])
        self._h = C.c_void_p()
        _lib.check(self._lib.ps_rccl_exchange_create(ident, int(rank), int(world), int(device), C.byref(self._h)))
        self.fn = C.cast(self._lib.ps_exchange_rccl, C.c_void_p)
        self.ctx = self._h
        self.error = None

    def stats(self, reset=False):
        from . import _lib
        n, b = C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.ps_rccl_exchange_stats(self._h, int(reset), C.byref(n), C.byref(b)))
        return n.value, b.value

    calls = property(lambda self: self.stats()[0])
    bytes = property(lambda self: self.stats()[1])

    def __call__(self, d_words, n_words, hip_stream=0):
        """one exchange on a raw device buffer (tests)"""
        from . import _lib
        _lib.check(self._lib.ps_exchange_rccl(self._h, C.c_void_p(int(d_words)), int(n_words), C.c_void_p(int(hip_stream))))

    def reraise(self, cause=None):
        pass            # (native: failures arrive as the library's own error text)

    def close(self):
        if self._h:
            self._lib.ps_rccl_exchange_destroy(self._h)
            self._h = C.c_void_p()


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

    def __init__(self, rank, world, engine=None, group=None, shard_hgt_donors=None, exchange="torch", **params):
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
            # "torch": torch.distributed collectives (RCCL under nccl, BOR under gloo); "torch_a2a": the RCCL branch's
            # all-to-all / OR / all-gather logic whatever the backend; "rccl": the library's own provider (ps_rccl_*)
            if exchange == "rccl":
                self.exchange = RcclExchange(rank, world, params.get("device", -1), group)
                self.sim.set_exchange(self.exchange.fn, self.exchange.ctx)
            else:
                self.exchange = TorchExchange(group, device=params.get("device", None), force_a2a=(exchange == "torch_a2a"))
                self.sim.set_exchange(self.exchange.fn)

    def run(self, count):
        try:
            self.sim.run(count)
        except Exception as e:          # the library only knows "the exchange failed (-1)": show why
            if self.exchange is not None:
                self.exchange.reraise(e)
            raise
        if self.exchange is not None:
            self.exchange.reraise()

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
        if isinstance(self.exchange, RcclExchange):
            self.exchange.close()
