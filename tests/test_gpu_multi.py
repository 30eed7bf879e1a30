"""Several core-site shards inside one process behind the C ABI (ps_multi, `pansim --gpus N`): main() of the
reference is one process (main.rs:429-553).  On a one-GPU box every shard sits on device 0; results must
equal the unsharded run bit for bit (a generation needs no exchange between the shards, the distance phase
sums integer numerators)."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_shards", [2, 3])
def test_multi_shards_equal_unsharded(pa, orc, n_shards):
    from orc_sim import OracleSim
    kw = dict(pop_size=300, core_size=5003, pan_genes=600, core_genes=200, HR_rate=0.3)
    P = 4000
    ref = OracleSim(seed=2, **kw)
    multi = pa.MultiSimulation(pa.make_params(seed=2, n_gen=3, max_distances=P, **kw), n_shards, devices=[0] * n_shards)
    assert multi.n_shards == n_shards
    multi.run(2)
    multi.run(1)
    multi.sync()
    for g in range(3):
        ref.generation(g)
    got = np.concatenate([s.core_genome.read_matrix() for s in multi.shards], axis=1)
    assert np.array_equal(got, ref.core)
    for s in multi.shards:
        assert np.array_equal(s.pan_genome.read_matrix(), ref.acc)
        assert np.array_equal(s.last_parents(), ref.last_idx)
    r1, r2 = orc.sample_pairs(2, 300, P)
    assert np.array_equal(multi.pairwise_counts(), orc.pairwise_hamming_counts(ref.core, 0, 5003, r1, r2))
    core_d, acc_d = multi.final_distances()
    assert np.array_equal(core_d, orc.pairwise_distances(ref.core, True, 200, r1, r2))
    assert np.array_equal(acc_d, orc.pairwise_distances(ref.acc, False, 200, r1, r2))
    multi.close()


def test_borrowed_shard_cannot_be_driven_alone(pa):
    # the shards' generations meet at barriers (shared parent weights, HGT delta exchange): a lone caller gets an error,
    # not a hang
    kw = dict(pop_size=300, core_size=900, pan_genes=700, core_genes=100, HGT_rate=0.6)
    multi = pa.MultiSimulation(pa.make_params(seed=5, n_gen=2, max_distances=20, **kw), 2, devices=[0, 0])
    multi.run(1)
    multi.sync()
    with pytest.raises(pa.PansimError) as e:
        multi.shards[1].run(1)
    assert e.value.code == -6
    with pytest.raises(pa.PansimError) as e:
        multi.shards[0].pan_genome.recombine(1)
    assert e.value.code == -6
    multi.run(1)             # the run itself is unharmed
    multi.sync()
    multi.close()


def test_multi_rejects_bad_arguments(pa):
    p = pa.make_params(seed=0, n_gen=1, max_distances=10, pop_size=10, core_size=5, pan_genes=20, core_genes=10)
    with pytest.raises(pa.PansimError):
        pa.MultiSimulation(p, 6)                       # more shards than core sites
    with pytest.raises(pa.PansimError):
        pa.MultiSimulation(p, 2, devices=[0, 99])      # no such device


def test_cli_gpus_2_equals_gpus_1(pa, tmp_path):
    # `pansim --gpus 2` (both shards on the one GPU of this box) against `--gpus 1`: all six files byte for byte
    exe = os.path.join(ROOT, "pansim_amd", "pansim")
    base = ["--pop_size", "1100", "--core_size", "2001", "--pan_genes", "500", "--core_genes", "100", "--n_gen", "3",
            "--seed", "6", "--max_distances", "5000", "--print_matrices", "--print_dist", "--print_selection",
            "--prop_positive", "0.3", "--verbose"]
    outs = []
    for n in (1, 2, 5):
        pref = tmp_path / ("g%d" % n)
        r = subprocess.run([exe, *base, "--gpus", str(n), "--outpref", str(pref)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append((pref, r.stdout))
    for pref, out in outs[1:]:
        assert out == outs[0][1]
        for suffix in (".tsv", "_freqs.txt", "_per_gen.tsv", "_selection.tsv", "_core_genome.csv", "_pangenome.csv"):
            assert filecmp.cmp(str(outs[0][0]) + suffix, str(pref) + suffix, shallow=False), suffix


def _hip_worker(rank, world, port, out, exchange="torch", comp=0.0):
    # one rank of a two-process site-sharded run with the REAL HIP engine (both ranks share the box's one GPU);
    # the exchange is pansim_amd.distributed over gloo (the driver's multi-GPU runs use nccl = RCCL)
    import sys
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pansim_amd.distributed import ShardedSimulation
    s = ShardedSimulation(rank, world, seed=21, n_gen=3, max_distances=900, device=0, shard_hgt_donors=True, exchange=exchange,
                          competition_strength=comp, **_MP_KW)
    s.run(3)
    # one exchange per generation (the HGT deltas), two with competition (the row-sharded average distances first)
    assert s.exchange is not None and s.exchange.calls == (6 if comp > 0.0 else 3) and s.exchange.error is None
    agree = s.parents_agree()
    core, acc = s.final_distances()
    np.save(os.path.join(out, "core_%d.npy" % rank), core)
    np.save(os.path.join(out, "acc_%d.npy" % rank), acc)
    np.save(os.path.join(out, "shard_%d.npy" % rank), s.sim.core_genome.read_matrix())
    np.save(os.path.join(out, "accm_%d.npy" % rank), s.sim.pan_genome.read_matrix())
    open(os.path.join(out, "agree_%d" % rank), "w").write(str(int(agree)))
    s.close()
    dist.destroy_process_group()


_MP_KW = dict(pop_size=260, core_size=3001, pan_genes=400, core_genes=100, HR_rate=0.3, HGT_rate=0.3)


@pytest.mark.parametrize("exchange", ["torch", "torch_a2a"])
def test_two_processes_with_the_hip_engine(pa, orc, tmp_path, exchange):
    # SURVEY 8(e) with one PROCESS per shard: every rank runs libpansim_hip with shard_count = 2, draws the same
    # parents, and the all-reduced integer numerators give the unsharded distances.  "torch_a2a": the HGT deltas take
    # the RCCL branch's all-to-all / OR / all-gather logic (over gloo: RCCL cannot put two ranks on one GPU)
    import torch.multiprocessing as mp
    from orc_sim import OracleSim
    world = 2
    port = 31000 + (os.getpid() + 7 * len(exchange)) % 2000
    comp = 3.0 if exchange == "torch_a2a" else 0.0          # (the second form also shards D-avg by rows over the ranks)
    mp.spawn(_hip_worker, args=(world, port, str(tmp_path), exchange, comp), nprocs=world, join=True)
    full = OracleSim(seed=21, competition_strength=comp, **_MP_KW)
    for g in range(3):
        full.generation(g)
    r1, r2 = orc.sample_pairs(21, _MP_KW["pop_size"], 900)
    want_core = orc.pairwise_distances(full.core, True, _MP_KW["core_genes"], r1, r2)
    want_acc = orc.pairwise_distances(full.acc, False, _MP_KW["core_genes"], r1, r2)
    shards = [np.load(tmp_path / ("shard_%d.npy" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate(shards, axis=1), full.core)
    for r in range(world):
        assert (tmp_path / ("agree_%d" % r)).read_text() == "1"
        assert np.array_equal(np.load(tmp_path / ("core_%d.npy" % r)), want_core)
        assert np.array_equal(np.load(tmp_path / ("acc_%d.npy" % r)), want_acc)
        assert np.array_equal(np.load(tmp_path / ("accm_%d.npy" % r)), full.acc)      # donor-sharded HGT + OR exchange over gloo


@pytest.mark.parametrize("kw,n_shards", [
    (dict(pop_size=64, core_size=5, pan_genes=120, core_genes=20), 5),                     # one site per shard
    (dict(pop_size=130, core_size=333, pan_genes=50, core_genes=50), 2),                   # no accessory genes at all
    (dict(pop_size=2, core_size=40, pan_genes=30, core_genes=10, HR_rate=0.9), 3),         # the smallest population
])
def test_multi_edge_shapes(pa, orc, kw, n_shards):
    from orc_sim import OracleSim
    P = 50
    ref = OracleSim(seed=8, **kw)
    multi = pa.MultiSimulation(pa.make_params(seed=8, n_gen=3, max_distances=P, **kw), n_shards, devices=[0] * n_shards)
    multi.run(3)
    multi.sync()
    for g in range(3):
        ref.generation(g)
    got = np.concatenate([s.core_genome.read_matrix() for s in multi.shards], axis=1)
    assert np.array_equal(got, ref.core)
    r1, r2 = orc.sample_pairs(8, kw["pop_size"], P)
    core_d, acc_d = multi.final_distances()
    assert np.array_equal(core_d, orc.pairwise_distances(ref.core, True, kw["core_genes"], r1, r2))
    want_acc = orc.pairwise_distances(ref.acc, False, kw["core_genes"], r1, r2)
    assert np.array_equal(acc_d, want_acc, equal_nan=True)
    multi.close()


@pytest.mark.parametrize("N,G,K,mode", [(50, 300, 2, 1), (130, 1000, 3, 2), (700, 4000, 8, 1), (2500, 4000, 5, 2), (2500, 300, 7, 0)])
def test_hgt_donor_shards_union_is_the_unsharded_result(pa, orc, N, G, K, mode):
    # HGT events are keyed per donor and ORed into the recipient (population.rs:599, :632): the union of the K donor
    # shards' results is the unsharded result, in both kernel forms.  (No exchange hook here: every handle applies its
    # own donors' events to its copy of the same matrix; ps_multi / distributed.py supply the exchange.)
    rng = np.random.default_rng(N + G + K)
    a = (rng.random((N, G)) < 0.3).astype(np.uint8)
    g1 = G * 9 // 10
    cb, ce, lr = [0, g1], [g1, G], [40.0, 6.5]
    seed, gen = 77, 12
    want = a.copy()
    orc.recombine_acc(want, seed, gen, cb, ce, lr)
    union = a.copy()
    for r in range(K):
        acc = pa.Population(N, G, 2, False, 0.3, seed, 0)
        acc.set_tuning("hgt_mode", mode)
        acc.set_rates([0.0, 0.0], lr, cb, ce)
        acc.load_matrix(a)
        acc.set_donor_shard(r, K)
        acc.recombine(gen)
        got = acc.read_matrix()
        assert (got >= a).all()                      # gain only
        union |= got
        if K > 1:
            assert not np.array_equal(got, want) or N < 10
        acc.close()
    assert np.array_equal(union, want)


@pytest.mark.parametrize("env", [{}, {"PANSIM_MULTI_REPLICATED_HGT": "1"}, {"PANSIM_HEAVY_HGT": "1", "PANSIM_HGT_MODE": "2"},
                                 {"PANSIM_HEAVY_HGT": "1", "PANSIM_HGT_MODE": "1"}, {"PANSIM_MULTI_EXCHANGE": "or"},
                                 {"PANSIM_MULTI_EXCHANGE": "or", "PANSIM_HEAVY_HGT": "1", "PANSIM_HGT_MODE": "2"}])
def test_multi_donor_sharded_hgt_forms(pa, orc, env):
    # ps_multi with the HGT donors sharded over the shards and the deltas ORed between them (default), against the
    # replicated form and in both HGT kernel forms / schedules: always the unsharded oracle run.  The exchange itself in both
    # forms: sliced peer copies on copy streams + one merge kernel (default, round 5) and the reading kernels ("or")
    from orc_sim import OracleSim
    kw = dict(pop_size=300, core_size=900, pan_genes=700, core_genes=100, HR_rate=0.2, HGT_rate=0.6)
    old = {k: os.environ.get(k) for k in ("PANSIM_MULTI_REPLICATED_HGT", "PANSIM_HEAVY_HGT", "PANSIM_HGT_MODE", "PANSIM_MULTI_EXCHANGE")}
    try:
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        ref = OracleSim(seed=5, **kw)
        multi = pa.MultiSimulation(pa.make_params(seed=5, n_gen=4, max_distances=200, **kw), 3, devices=[0, 0, 0])
        multi.run(4)
        multi.sync()
        for g in range(4):
            ref.generation(g)
        assert np.array_equal(np.concatenate([s.core_genome.read_matrix() for s in multi.shards], axis=1), ref.core)
        for s in multi.shards:
            assert np.array_equal(s.pan_genome.read_matrix(), ref.acc)
            assert np.array_equal(s.last_parents(), ref.last_idx)
        calls = [s.exchange_stats()[0] for s in multi.shards]
        assert calls == ([0, 0, 0] if "PANSIM_MULTI_REPLICATED_HGT" in env else [4, 4, 4])
        # the three softmaxes of a generation run once per process (shard 0) and are handed to the other shards
        for k, s in enumerate(multi.shards):
            gens, _wait, weights_ms, _draw = s.host_timing()
            assert gens == 4 and (weights_ms > 0.0) == (k == 0)
        multi.close()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_native_rccl_exchange_provider(pa, orc):
    # ps_rccl_* / ps_exchange_rccl, the library's own ps_exchange_fn (dlopen of librccl.so, ncclSend / ncclRecv group, OR
    # kernel, ncclAllGather).  One GPU allows a communicator of ONE rank only (RCCL refuses two ranks on a device), which
    # still runs every call of the provider: the OR over one rank leaves the buffer as it was.
    import torch
    from pansim_amd.distributed import RcclExchange
    assert pa.load().ps_rccl_available() == 1
    torch.cuda.set_device(0)
    x = RcclExchange(0, 1, 0)
    rng = np.random.default_rng(3)
    st = torch.cuda.Stream()
    for n in (1, 1000, 4099, 4099):
        host = rng.integers(-2**62, 2**62, n, dtype=np.int64)
        with torch.cuda.stream(st):
            buf = torch.from_numpy(host).cuda()
            x(buf.data_ptr(), n, st.cuda_stream)
        st.synchronize()
        assert np.array_equal(buf.cpu().numpy(), host)
    assert x.stats() == (4, 0)
    # inside the generation loop: shard 0 of 2 whose exchange is the one-rank communicator applies its own donors' events
    # only -- exactly what the hook-less form (fn == NULL) does
    kw = dict(pop_size=200, core_size=600, pan_genes=500, core_genes=100, HR_rate=0.2, HGT_rate=0.6)
    mats = []
    for native in (False, True):
        sim = pa.Simulation(pa.make_params(seed=9, n_gen=3, max_distances=50, shard_rank=0, shard_count=2, device=0, **kw))
        if native:
            sim.set_exchange(x.fn, x.ctx)
        else:
            sim.set_exchange(None)
        sim.run(3)
        sim.sync()
        mats.append((sim.pan_genome.read_matrix(), sim.core_genome.read_matrix(), sim.last_parents()))
        sim.close()
    assert x.stats()[0] == 4 + 3
    for a, b in zip(*mats):
        assert np.array_equal(a, b)
    x.close()


@pytest.mark.parametrize("kw,n_shards", [
    (dict(pop_size=300, core_size=900, pan_genes=700, core_genes=100, HR_rate=0.2, HGT_rate=0.6, competition_strength=4.0), 3),
    (dict(pop_size=1300, core_size=500, pan_genes=3000, core_genes=1000, HGT_rate=0.3, competition_strength=9.0), 4),
])
def test_multi_competition_average_distance_sharded_by_rows(pa, orc, kw, n_shards):
    # --competition_strength > 0 in a sharded run: D-avg (population.rs:753-784) is sharded by rows like the HGT donors
    # -- shard k folds the rows of its individuals on the matrix cores, the OR exchange of the N doubles is the
    # all-gather -- and the parents every shard draws from it equal the unsharded oracle run's
    from orc_sim import OracleSim
    ref = OracleSim(seed=3, **kw)
    multi = pa.MultiSimulation(pa.make_params(seed=3, n_gen=3, max_distances=100, **kw), n_shards, devices=[0] * n_shards)
    for g in range(3):
        multi.run(1)
        multi.sync()
        ref.generation(g)
        for s in multi.shards:
            assert np.array_equal(s.last_parents(), ref.last_idx), "parents differ at generation %d" % g
    assert np.array_equal(np.concatenate([s.core_genome.read_matrix() for s in multi.shards], axis=1), ref.core)
    for s in multi.shards:
        assert np.array_equal(s.pan_genome.read_matrix(), ref.acc)
    # two exchanges per generation and shard: the average distances, then the HGT deltas
    assert [s.exchange_stats()[0] for s in multi.shards] == [6] * n_shards
    multi.close()


def test_bench_bare_form_two_ranks_one_gpu(pa):
    # `python3 bench.py --gpus 2` with NO launcher around it starts its own two ranks (both on
    # this box's one GPU: the nccl data plane is refused, the run falls back to gloo host copies and says so), prints ONE
    # line with n_gpus 2 and the north-star figure filled.  The second workload runs on a shortened genome here
    # (PANSIM_BENCH_NS_CORE_SIZE); profiles/r05_a_bare_2ranks_one_gpu.json is the full-size run of the same command.
    import json
    import sys
    env = dict(os.environ, PANSIM_BENCH_NS_CORE_SIZE="120000")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PANSIM_BENCH_BACKEND", "PANSIM_BENCH_STUB"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["line"] == "final" and d["launcher"]["self_launched"]
    assert d["value"] > 100 and d["roofline"]["frac"] > 0.1
    assert d["north_star_generations_per_s"] > 0
    assert d["data_plane"].startswith("gloo") and d["control_plane"] == "gloo"
    ns = d["north_star_scaling"]
    assert ns["exchange"]["mode"] == "torch" and ns["exchange"]["calls"] == 10
    chain = ns["exchange"]["provider_chain"]
    assert [c["provider"] for c in chain] == ["rccl", "torch"] and not chain[0]["ok"] and chain[1]["ok"]


@pytest.mark.parametrize("exchange", ["sliced", "or"])
def test_multi_eight_shards_with_competition(pa, orc, monkeypatch, exchange):
    # ps_multi at the north-star's shard count: 8 shards of one process (all on this box's one GPU), HGT donors and D-avg rows
    # sharded 8 ways, buffer lengths 8 does not divide (N = 301 doubles; N x GW = 301 x 7 words padded to 4096), both forms of
    # the in-process exchange (peer copies on copy streams + one merge kernel; the reading kernels) -- always the unsharded
    # oracle run
    from orc_sim import OracleSim
    if exchange == "or":
        monkeypatch.setenv("PANSIM_MULTI_EXCHANGE", "or")
    kw = dict(pop_size=301, core_size=1203, pan_genes=520, core_genes=100, HR_rate=0.2, HGT_rate=0.5, competition_strength=3.0)
    ref = OracleSim(seed=14, **kw)
    multi = pa.MultiSimulation(pa.make_params(seed=14, n_gen=4, max_distances=150, **kw), 8, devices=[0] * 8)
    for g in range(4):
        multi.run(1)
        multi.sync()
        ref.generation(g)
        for s in multi.shards:
            assert np.array_equal(s.last_parents(), ref.last_idx), "parents differ at generation %d" % g
    assert np.array_equal(np.concatenate([s.core_genome.read_matrix() for s in multi.shards], axis=1), ref.core)
    for s in multi.shards:
        assert np.array_equal(s.pan_genome.read_matrix(), ref.acc)
    assert [s.exchange_stats()[0] for s in multi.shards] == [8] * 8
    multi.close()
