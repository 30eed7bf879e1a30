"""world_size-2 gloo test of the site-sharded path on CPU.

The exchange logic (shard bounds, rank-redundant parent draws, all-reduce of the integer
Hamming numerators) is the product's pansim_amd.distributed; the per-rank compute engine is a
CPU double built on the oracle (this container has no GPU).  The sharded result must equal the
unsharded oracle run bit for bit.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(pop_size=60, core_size=2001, pan_genes=300, core_genes=100, HR_rate=0.3, HGT_rate=0.3)
SEED, GENS, P = 4, 4, 700


class _Pop:
    def __init__(self, owner, core):
        self.o, self.core = owner, core

    def pairwise_counts(self, r1, r2):
        from oracle import oracle as orc
        m = self.o.ref.core if self.core else self.o.ref.acc
        return (orc.pairwise_hamming_counts(m, 0, m.shape[1], r1, r2),)

    def pairwise_distances(self, P, r1, r2):
        from oracle import oracle as orc
        m = self.o.ref.core if self.core else self.o.ref.acc
        return orc.pairwise_distances(m, self.core, KW["core_genes"], r1[:P], r2[:P])


class OracleEngine:
    """CPU double of pansim_amd.Simulation for one site shard"""

    def __init__(self, shard_rank, shard_count, seed, max_distances, **kw):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from orc_sim import OracleSim
        from oracle import oracle as orc
        from pansim_amd.distributed import shard_bounds
        b, e = shard_bounds(kw["core_size"], shard_rank, shard_count)
        self.ref = OracleSim(seed=seed, site_begin=b, site_end=e, **kw)
        self.range1, self.range2 = orc.sample_pairs(seed, kw["pop_size"], max_distances)
        self.core_genome, self.pan_genome = _Pop(self, True), _Pop(self, False)
        self.gen = 0

    def run(self, count):
        for _ in range(count):
            self.ref.generation(self.gen)
            self.gen += 1

    def sync(self):
        pass

    def last_parents(self):
        return self.ref.last_idx

    def close(self):
        pass


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pansim_amd.distributed import ShardedSimulation
    s = ShardedSimulation(rank, world, engine=OracleEngine, seed=SEED, max_distances=P, **KW)
    s.run(GENS)
    agree = s.parents_agree()
    core, acc = s.final_distances()
    np.save(os.path.join(out, "core_%d.npy" % rank), core)
    np.save(os.path.join(out, "acc_%d.npy" % rank), acc)
    np.save(os.path.join(out, "shard_%d.npy" % rank), s.sim.ref.core)
    open(os.path.join(out, "agree_%d" % rank), "w").write(str(int(agree)))
    dist.destroy_process_group()


def test_two_rank_site_sharding_matches_unsharded(tmp_path, orc):
    world = 2
    port = 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from orc_sim import OracleSim
    full = OracleSim(seed=SEED, **KW)
    for g in range(GENS):
        full.generation(g)
    r1, r2 = orc.sample_pairs(SEED, KW["pop_size"], P)
    want_core = orc.pairwise_distances(full.core, True, KW["core_genes"], r1, r2)
    want_acc = orc.pairwise_distances(full.acc, False, KW["core_genes"], r1, r2)
    shards = [np.load(tmp_path / ("shard_%d.npy" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate(shards, axis=1), full.core)
    for r in range(world):
        assert (tmp_path / ("agree_%d" % r)).read_text() == "1"
        assert np.array_equal(np.load(tmp_path / ("core_%d.npy" % r)), want_core)
        assert np.array_equal(np.load(tmp_path / ("acc_%d.npy" % r)), want_acc)


def test_shard_bounds_partition():
    from pansim_amd.distributed import core_distances_from_counts, shard_bounds
    for L in (1, 7, 1200000, 1200001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(L, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == L
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
    # integer /2 before the float divide (population.rs:817-822)
    assert core_distances_from_counts([2, 3, 10], 9).tolist() == [1 / 9, 1 / 9, 5 / 9]


OR_SIZES = (1, 7, 4096, 10001)


def _or_worker(rank, world, port, out, force_a2a):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pansim_amd.distributed import OrScratch, or_all_reduce
    rng = np.random.default_rng(100 + rank)
    scratch = None
    for rep, n in enumerate(OR_SIZES + OR_SIZES[-1:]):          # (the last size twice: the scratch buffers are reused)
        mine = torch.from_numpy(rng.integers(-2**62, 2**62, n, dtype=np.int64) & rng.integers(-2**62, 2**62, n, dtype=np.int64))
        buf = mine.clone()
        if force_a2a and (scratch is None or not scratch.fits(buf, world)):
            scratch = OrScratch(n, world, buf)
        # (both forms of the last step: the direct all-to-all of the merged slices, and the ring all-gather)
        sent = or_all_reduce(buf, force_a2a=force_a2a, scratch=scratch, ring_gather=(rep % 2 == 1))
        np.save(os.path.join(out, "or_%d_%d.npy" % (rep, rank)), buf.numpy())
        np.save(os.path.join(out, "in_%d_%d.npy" % (rep, rank)), mine.numpy())
        part = (n + world - 1) // world
        assert sent == (2 * part * 8 * (world - 1) if force_a2a else 2 * n * 8 * (world - 1) // world)
        if force_a2a:                                           # the pad words behind the delta must still be zero
            assert not scratch.send[n:].any() or world * part == n
    dist.destroy_process_group()


@pytest.mark.parametrize("world,force_a2a", [(3, False), (2, True), (3, True), (8, True)])
def test_or_all_reduce_of_hgt_deltas(tmp_path, world, force_a2a):
    # the per-generation exchange step of the donor-sharded HGT: every rank ends with the OR of all ranks' buffers.
    # force_a2a runs the RCCL branch's logic (pad to K slices, all-to-all, local OR, all-gather) over gloo; the sizes
    # include lengths that K does not divide and lengths below K (empty slices)
    port = 27000 + (os.getpid() * 7 + world * 2 + int(force_a2a)) % 2000
    mp.spawn(_or_worker, args=(world, port, str(tmp_path), force_a2a), nprocs=world, join=True)
    for rep, n in enumerate(OR_SIZES + OR_SIZES[-1:]):
        want = np.zeros(n, np.int64)
        for r in range(world):
            want |= np.load(tmp_path / ("in_%d_%d.npy" % (rep, r)))
        for r in range(world):
            assert np.array_equal(np.load(tmp_path / ("or_%d_%d.npy" % (rep, r))), want)
