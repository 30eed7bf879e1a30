"""K ranks of the library's native RCCL exchange provider inside ONE process on ONE GPU, one host thread per rank, over
tests/libfake_rccl.so (PANSIM_RCCL_LIBRARY) -- test infrastructure, started as a subprocess by tests/test_gpu_rccl_fake.py
(the pytest process itself keeps the real librccl: a process resolves the RCCL symbols once).

    python tests/rccl_threads_worker.py raw K [ring]     ps_exchange_rccl on raw device buffers against the numpy OR
    python tests/rccl_threads_worker.py sim K COMP [heavy]  K site shards with sharded HGT donors inside ps_sim_run against
                                                         the UNSHARDED oracle run (COMP = --competition_strength)
    python tests/rccl_threads_worker.py fail WHAT        one injected RCCL failure (PANSIM_FAKE_RCCL_FAIL) inside the exchange
Prints one JSON line; exit code 0 iff everything matched."""
import ctypes as C
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
FAKE = os.path.join(ROOT, "tests", "libfake_rccl.so")
os.environ["PANSIM_RCCL_LIBRARY"] = FAKE

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the library: one HIP runtime per process, tests/conftest.py)

import pansim_amd as pa  # noqa: E402
from pansim_amd import _lib  # noqa: E402


def handles(K):
    """K communicator handles of one id, created by K threads (ncclCommInitRank returns when all ranks are inside)"""
    lib = pa.load()
    ident = np.zeros(128, np.uint8)
    _lib.check(lib.ps_rccl_unique_id(ident))
    assert bytes(ident[8:17]) == b"fake_rccl", "the test double was not the library that got loaded"
    hs, errs = [C.c_void_p() for _ in range(K)], [None] * K

    def make(r):
        try:
            torch.cuda.set_device(0)
            _lib.check(lib.ps_rccl_exchange_create(ident, r, K, 0, C.byref(hs[r])))
        except Exception as e:
            errs[r] = repr(e)

    run_threads(make, K)
    assert errs == [None] * K, errs
    return lib, hs


def run_threads(fn, K):
    ts = [threading.Thread(target=fn, args=(r,)) for r in range(K)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not any(t.is_alive() for t in ts), "a rank never returned"


def raw(K):
    lib, hs = handles(K)
    rng = np.random.default_rng(100 + K)
    # lengths below K, lengths K does not divide, a repeat (the kept scratch set), five distinct lengths (the four scratch
    # sets are recycled), a last repeat of the first (allocated again after recycling)
    lengths = [max(1, K - 1), 1, 1000 + 37, 4099, 4099, 65536 + 3, 8 * K, max(1, K - 1)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    checked, want_bytes = 0, 0
    for n in lengths:
        host = [rng.integers(-2**62, 2**62, n, dtype=np.int64) for _ in range(K)]
        dev = [torch.from_numpy(h).cuda() for h in host]
        torch.cuda.synchronize()
        errs = [None] * K

        def go(r):
            try:
                torch.cuda.set_device(0)
                _lib.check(lib.ps_exchange_rccl(hs[r], C.c_void_p(dev[r].data_ptr()), n, C.c_void_p(streams[r].cuda_stream)))
                streams[r].synchronize()
            except Exception as e:
                errs[r] = repr(e)

        run_threads(go, K)
        assert errs == [None] * K, (n, errs)
        want = host[0].copy()
        for h in host[1:]:
            want |= h
        for r in range(K):
            assert np.array_equal(dev[r].cpu().numpy(), want), "rank %d of %d, %d words" % (r, K, n)
        part = (n + K - 1) // K
        want_bytes += 2 * part * 8 * (K - 1)
        checked += 1
    for r in range(K):
        calls, nbytes = C.c_uint64(), C.c_uint64()
        _lib.check(lib.ps_rccl_exchange_stats(hs[r], 0, C.byref(calls), C.byref(nbytes)))
        assert (calls.value, nbytes.value) == (len(lengths), want_bytes), (calls.value, nbytes.value, want_bytes)
    for h in hs:
        lib.ps_rccl_exchange_destroy(h)
    return {"mode": "raw", "K": K, "lengths": lengths, "exchanges_checked": checked}


def sim(K, comp):
    from orc_sim import OracleSim
    lib, hs = handles(K)
    kw = dict(pop_size=260, core_size=3001, pan_genes=400, core_genes=100, HR_rate=0.3, HGT_rate=0.6)
    if K == 8:
        kw.update(pop_size=1100, core_size=4003, pan_genes=700)
    if comp > 0.0:
        kw["competition_strength"] = comp
    gens, seed = 3, 21
    sims = [pa.Simulation(pa.make_params(seed=seed, n_gen=gens, max_distances=300, shard_rank=r, shard_count=K, device=0, **kw))
            for r in range(K)]
    fn = C.cast(lib.ps_exchange_rccl, C.c_void_p)
    for r, s in enumerate(sims):
        s.set_exchange(fn, hs[r])
    errs = [None] * K

    def go(r):
        try:
            torch.cuda.set_device(0)
            sims[r].run(gens)
            sims[r].sync()
        except Exception as e:
            errs[r] = repr(e)

    run_threads(go, K)
    assert errs == [None] * K, errs
    ref = OracleSim(seed=seed, **kw)
    for g in range(gens):
        ref.generation(g)
    assert np.array_equal(np.concatenate([s.core_genome.read_matrix() for s in sims], axis=1), ref.core), "core shards"
    for r, s in enumerate(sims):
        assert np.array_equal(s.pan_genome.read_matrix(), ref.acc), "accessory matrix of rank %d" % r
        assert np.array_equal(s.last_parents(), ref.last_idx), "parents of rank %d" % r
    per_gen = 2 if comp > 0.0 else 1
    for r in range(K):
        calls = C.c_uint64()
        _lib.check(lib.ps_rccl_exchange_stats(hs[r], 0, C.byref(calls), None))
        # (D-avg computed ahead of the sweep: generation 0's at its own start, then one per generation but the run's last)
        assert calls.value == per_gen * gens, calls.value
    for s in sims:
        s.close()
    for h in hs:
        lib.ps_rccl_exchange_destroy(h)
    return {"mode": "sim", "K": K, "competition_strength": comp, "generations": gens, "pop_size": kw["pop_size"]}


def fail(what):
    """an RCCL call fails once inside the exchange: the library returns PS_ERR_STATE with the call's name, the group is
    closed (the NEXT exchange on the same thread works), nothing hangs"""
    os.environ["PANSIM_FAKE_RCCL_FAIL"] = what
    if what == "allgather":
        os.environ["PANSIM_RCCL_GATHER"] = "ring"          # (the default gathers with direct sends: no ncclAllGather call)
    lib, hs = handles(1)
    torch.cuda.set_device(0)
    host = np.arange(1000, dtype=np.int64)
    buf = torch.from_numpy(host).cuda()
    rc = lib.ps_exchange_rccl(hs[0], C.c_void_p(buf.data_ptr()), 1000, None)
    msg = lib.ps_last_error().decode()
    assert rc == _lib.PS_ERR_STATE, (rc, msg)
    names = {"send": "ncclSend", "recv": "ncclRecv", "groupend": "ncclGroupEnd", "allgather": "AllGather"}
    assert names[what] in msg, msg
    buf.copy_(torch.from_numpy(host))
    _lib.check(lib.ps_exchange_rccl(hs[0], C.c_void_p(buf.data_ptr()), 1000, None))      # (the injection fires once)
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy(), host)
    lib.ps_rccl_exchange_destroy(hs[0])
    return {"mode": "fail", "what": what, "message": msg}


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "raw":
        if len(sys.argv) > 3:
            os.environ["PANSIM_RCCL_GATHER"] = sys.argv[3]      # "ring": ncclAllGather instead of the direct sends
        out = raw(int(sys.argv[2]))
    elif mode == "sim":
        if len(sys.argv) > 4 and sys.argv[4] == "heavy":
            # the binned HGT in its turn-taking schedule: with competition on, D-avg of generation g + 1 (row-sharded, its
            # all-gather through the exchange) is computed AHEAD of sweep(g) -- two exchanges per generation in another order
            os.environ["PANSIM_HEAVY_HGT"] = "1"
            os.environ["PANSIM_HGT_MODE"] = "2"
        out = sim(int(sys.argv[2]), float(sys.argv[3]))
    else:
        out = fail(sys.argv[2])
    print(json.dumps(out), flush=True)
