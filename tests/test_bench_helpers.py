"""CPU checks of bench.py's bookkeeping (no GPU): the roofline object is labelled from the sweep form the library reports
and priced with the newest committed PMC pass of THAT kernel; the committed bench lines of the round keep the contract's
keys and the round-4 meaning of `value`."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    import importlib
    return importlib.import_module("bench")


def test_sweep_roofline_follows_the_reported_form():
    b = _bench()
    base = {"bytes_per_launch": 2.4e9, "sweep_avg_ms": 0.5, "launches": 20}
    wave = b.sweep_roofline(dict(base, sweep_form=1), True)
    assert "core_sweep_wave_kernel" in wave["kernel"]
    assert abs(wave["achieved"] - 4800.0) < 1e-6 and abs(wave["frac"] - 0.6) < 1e-9 and wave["peak"] == 8000.0
    assert "_pmc_sweep.json" in wave["traffic_source"] and 2.4e9 < wave["traffic"] < 2.6e9
    win = b.sweep_roofline(dict(base, sweep_form=3, bytes_per_launch=2 * 65536 * 150000.0, sweep_avg_ms=4.2), True)
    assert "core_sweep_window_kernel" in win["kernel"] and "_pmc_window_sweep.json" in win["traffic_source"]
    assert 1.0 < win["traffic"] / win["algorithmic_bytes_per_launch"] < 1.25
    # a workload of another size: the measured traffic / algorithmic ratio is applied, and the source string says so
    other = b.sweep_roofline(dict(base, sweep_form=3, bytes_per_launch=1.0e9, sweep_avg_ms=1.0), True)
    assert "scaled by algorithmic bytes" in other["traffic_source"]
    blk = b.sweep_roofline(dict(base, sweep_form=4), True)
    assert "core_sweep_block_kernel" in blk["kernel"] and "block_sweep" in blk["traffic_source"]
    none = b.sweep_roofline(dict(base, sweep_form=0), True)
    assert none["traffic"] is None


def test_distance_roofline_forms():
    b = _bench()
    r = b.distance_roofline(7, 8192, 1200000, 4000, 1 << 25, 52.0, 3.0)
    assert r["bound"] == "mfma-fp4" and r["peak"] == 10000.0 and 0.5 < r["frac"] < 0.7
    r = b.distance_roofline(4, 65536, 150000, 4000, 100000, 3.5, 0.1)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s"
    r = b.distance_roofline(1, 1000, 1200000, 4000, 100000, 1.9, 0.1)
    assert r["bound"] == "valu" and r["frac"] < r["frac_at_measured_issue_costs"]


def test_committed_round4_lines_keep_the_contract():
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_[gh]_bench_default.json")))
    assert files
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        assert need <= set(d), sorted(need - set(d))
        assert d["dtype"] == "u8" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
        assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
        assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
        assert abs(d["value"] - d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    # N > 1: `value` is the whole simulation's rate, the aggregate has its own key
    d = json.loads(open(os.path.join(ROOT, "profiles", "r04_b_gloo_2ranks_one_gpu.json")).read().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and abs(d["shard_generations_per_s"] - 2 * d["value"]) < 1e-6 * d["value"]
    assert {"north_star_generations_per_s", "north_star_sweep_frac_per_rank", "north_star_exposed_non_sweep_ms",
            "north_star_collective_bytes_per_generation"} <= set(d)
