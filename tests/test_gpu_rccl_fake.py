"""K >= 2 of the library's native RCCL exchange provider (pansim_amd/csrc/exchange_rccl.h: ps_rccl_*, ps_exchange_rccl) on a
ONE-GPU box, through a test double of librccl (tests/fake_rccl.cpp, loaded via PANSIM_RCCL_LIBRARY) that plays several
ranks as threads of one process.  The slice offsets, the pad of lengths K does not divide, the OR over K received slices
(rccl_or_slices_kernel), the all-gather offsets, the byte accounting and the reuse of the four scratch sets all execute
here for worlds of 2, 3 and 8 -- on raw buffers against the numpy OR, and inside a donor-sharded ps_sim_run against the
UNSHARDED oracle run (HGT gains are ORed into the recipient, population.rs:632).  The product keeps opening the real
librccl by default; tests/test_gpu_multi.py::test_native_rccl_exchange_provider runs that one with its one-rank world."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "rccl_threads_worker.py")


def _worker(*args, timeout=600):
    if not os.path.exists(os.path.join(ROOT, "tests", "libfake_rccl.so")):      # (normally prebuilt by __graft_entry__.build())
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests")])
    env = dict(os.environ)
    env.pop("PANSIM_FAKE_RCCL_FAIL", None)
    p = subprocess.run([sys.executable, WORKER] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("K,gather", [(2, "direct"), (3, "direct"), (8, "direct"), (3, "ring"), (8, "ring")])
def test_exchange_rccl_raw_buffers(K, gather):
    # gather: the last step as direct sends to every peer (default) or as ncclAllGather (PANSIM_RCCL_GATHER=ring)
    out = _worker("raw", K, gather)
    assert out["K"] == K and out["exchanges_checked"] == len(out["lengths"]) == 8
    assert min(out["lengths"]) < K or K == 2


@pytest.mark.gpu
@pytest.mark.parametrize("K,comp,hgt", [(2, 0.0, ""), (3, 0.0, ""), (8, 0.0, ""), (3, 4.0, ""), (8, 2.5, ""), (3, 4.0, "heavy"), (2, 0.0, "heavy")])
def test_exchange_rccl_inside_the_generation_loop(K, comp, hgt):
    # K site shards, HGT donors sharded K ways, (comp > 0: D-avg sharded by rows too): core shards, accessory matrix and
    # parents of every rank equal the unsharded oracle run.  "heavy": the binned HGT taking turns with the sweep, and with
    # competition D-avg of the next generation computed ahead of the sweep
    out = _worker("sim", K, comp, *([hgt] if hgt else []))
    assert out["K"] == K and out["generations"] == 3


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["send", "recv", "groupend", "allgather"])
def test_exchange_rccl_error_paths(what):
    # a failing ncclSend / ncclRecv must not leave the group open; RCCL failures are PS_ERR_STATE
    out = _worker("fail", what)
    assert out["what"] == what and "failed" in out["message"]


def test_fake_rccl_exports_what_the_provider_resolves():
    # (CPU) the double implements exactly the entry points exchange_rccl.h looks up with dlsym
    import re
    src = open(os.path.join(ROOT, "pansim_amd", "csrc", "exchange_rccl.h")).read()
    wanted = set(re.findall(r'sym\("(nccl\w+)"\)', src))
    assert len(wanted) == 9
    so = os.path.join(ROOT, "tests", "libfake_rccl.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests")])
    syms = subprocess.run(["nm", "-D", "--defined-only", so], stdout=subprocess.PIPE, text=True, check=True).stdout
    have = set(re.findall(r" T (nccl\w+)", syms))
    assert wanted <= have, sorted(wanted - have)
    # and the product does not depend on the RCCL headers any more
    assert "#include <rccl" not in src
