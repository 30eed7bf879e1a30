"""`python3 bench.py --gpus N` must start its own ranks and always leave ONE parseable line.

The launcher, the gloo control plane, the provider fallback chain, the early contract line and the watchdog are host logic;
they run here on CPU with tests/bench_stub.py standing in for the HIP engine (the real engine takes the same path on the
GPU box: tests/test_gpu_multi.py::test_bench_bare_form_two_ranks_one_gpu)."""
import json
import os
import subprocess
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args, timeout=240):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update({"PANSIM_BENCH_STUB": "bench_stub", "PYTHONPATH": os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", ""),
                "PANSIM_BENCH_PG_TIMEOUT": "60"})
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_bare_form_launches_its_own_ranks():
    p, lines = _run({}, "--gpus", "2", "--steps", "20", "--warmup", "5")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, "the bare form prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["line"] == "final" and d["launcher"]["self_launched"]
    assert d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["schema_version"] == 2
    assert d["north_star_generations_per_s"] > 0 and d["north_star_scaling"]["scaling"] == "strong"
    # the chain: no RCCL here -> the torch provider, probed with a buffer 2 does not divide
    chain = d["north_star_scaling"]["exchange"]["provider_chain"]
    assert [c["provider"] for c in chain] == ["rccl", "torch"] and not chain[0]["ok"] and chain[1]["ok"]
    assert d["north_star_exchange"] == "torch" and d["control_plane"] == "gloo"
    assert d["roofline"]["traffic_measured_in_this_run"] is False
    # round 6: the third workload (BASELINE configs[4]: the distance phase of --pop_size 8192, P = 2^25, sites split over the
    # ranks, one all-reduce of the partial counts), what the probes saw at the top level
    nd = d["north_star_distance"]
    assert nd["n_gpus"] == 2 and nd["pairs"] == 1 << 25 and nd["collective_bytes_per_rank"] == 4 << 25 and nd["mpairs_per_s"] > 0
    assert d["north_star_distance_mpairs_per_s"] == nd["mpairs_per_s"]
    assert d["rccl_ranks_seen"] == 0 and d["north_star_exchange_world_size"] == 2       # (no RCCL on a CPU box)


def test_three_ranks_and_every_provider_failing():
    p, lines = _run({"PANSIM_BENCH_STUB_MODE": "no_torch"}, "--gpus", "3", "--steps", "5", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 3 and d["north_star_generations_per_s"] > 0, "the figure is filled even without a provider"
    mode = d["north_star_scaling"]["exchange"]["mode"]
    assert mode.startswith("none (fallback: ") and "rccl:" in mode and "torch:" in mode


def test_second_workload_raising_keeps_the_contract_line():
    p, lines = _run({"PANSIM_BENCH_STUB_MODE": "raise_ns"}, "--gpus", "2", "--steps", "5", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["value"] > 0 and d["line"] == "final" and "second workload failed" in d["north_star_scaling"]["error"]
    assert d["north_star_generations_per_s"] is None


def test_second_workload_hanging_is_cut_by_the_watchdog():
    p, lines = _run({"PANSIM_BENCH_STUB_MODE": "hang_ns", "PANSIM_BENCH_NS_TIMEOUT": "6"}, "--gpus", "2", "--steps", "5", "--warmup", "1")
    assert len(lines) == 1, (p.returncode, p.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 2
    assert d["line"].startswith("contract") or "watchdog" in d["north_star_scaling"]["error"]
    # a run cut short is not a success -- the line stands, the exit code says so
    assert p.returncode != 0


def test_third_workload_hanging_keeps_the_first_two():
    p, lines = _run({"PANSIM_BENCH_STUB_MODE": "hang_nd", "PANSIM_BENCH_NS_TIMEOUT": "6"}, "--gpus", "2", "--steps", "5", "--warmup", "1")
    assert len(lines) == 1, (p.returncode, p.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["north_star_generations_per_s"] > 0 and d["north_star_scaling"]["scaling"] == "strong"
    assert d["line"].startswith("contract + north_star_scaling") or "watchdog" in d["north_star_distance"]["error"]
    assert p.returncode != 0


def test_budgets_fit_the_drivers_limit():
    # first attempt <= 700 s, the gloo retry <= 500 s, each north-star workload <= 300 s: 1200 s + start-up < 1800 s even
    # when the first attempt hangs before a line exists (here with the budgets scaled down: 8 s, then a retry that works)
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'PANSIM_BENCH_LAUNCH_TIMEOUT", "700"', src) and re.search(r'PANSIM_BENCH_RETRY_TIMEOUT", "500"', src)
    assert len(re.findall(r'PANSIM_BENCH_NS_TIMEOUT", "300"', src)) == 2
    p, lines = _run({"PANSIM_BENCH_STUB_MODE": "hang_all", "PANSIM_BENCH_LAUNCH_TIMEOUT": "8", "PANSIM_BENCH_RETRY_TIMEOUT": "120"},
                    "--gpus", "2", "--steps", "5", "--warmup", "1")
    assert len(lines) == 1, (p.returncode, p.stderr[-2000:])
    d = json.loads(lines[0])
    att = d["launcher"]["attempts"]
    assert [a["backend"] for a in att] == ["nccl", "gloo"] and att[0]["timed_out"] and att[0]["json_lines"] == 0 and att[0]["wall_s"] < 40
    assert att[1]["rc"] == 0 and d["line"] == "final" and p.returncode == 0


def test_a_rank_dying_in_the_second_workload_keeps_the_contract_line():
    p, lines = _run({"PANSIM_BENCH_STUB_MODE": "die_ns", "PANSIM_BENCH_NS_TIMEOUT": "30"}, "--gpus", "2", "--steps", "5", "--warmup", "1")
    assert len(lines) == 1, (p.returncode, p.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 2 and d["launcher"]["attempts"][0]["json_lines"] >= 1


def test_under_torchrun_rank0_prints_the_contract_line_first():
    """the driver's own multi-GPU form: torch.distributed.run around bench.py; two lines, both complete"""
    env = dict(os.environ)
    env.update({"PANSIM_BENCH_STUB": "bench_stub", "PYTHONPATH": os.path.join(ROOT, "tests")})
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 3 and lines[0]["line"].startswith("contract") and lines[1]["line"].startswith("contract + north_star_scaling")
    assert lines[2]["line"] == "final" and "north_star_distance" in lines[2]
    lines = [lines[0], lines[2]]
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}
    for d in lines:
        assert need <= set(d)
    assert lines[0]["value"] == lines[1]["value"] and "north_star_scaling" in lines[1]


def test_world_size_mismatch_is_still_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", PANSIM_BENCH_STUB="bench_stub", PYTHONPATH=os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=60)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr
