#!/usr/bin/env python3
"""Same-box A/B of several (library build, environment) arms inside the generation loop: child processes alternate over the arms,
each reporting bench.py's in-loop HIP-event time per sweep launch and its period.
  python scripts/ab_matrix.py CONFIG ROUNDS name=LIB[,ENV=VALUE...] ...      (LIB 'default' = pansim_amd/libpansim_hip.so)"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg, rounds = sys.argv[1], int(sys.argv[2])
extra = []
if "+" in cfg:                   # e.g. cfg2+competition_strength=10: further bench.py flags
    cfg, rest = cfg.split("+", 1)
    for kv in rest.split("+"):
        k, v = kv.split("=", 1)
        extra += ["--" + k, v]
arms = []
for spec in sys.argv[3:]:
    name, rest = spec.split("=", 1)
    parts = rest.split(",")
    arms.append((name, parts[0], dict(kv.split("=", 1) for kv in parts[1:])))
res = {a[0]: {"sweep_ms": [], "period_ms": [], "value": []} for a in arms}
for r in range(rounds):
    for name, lib, env_add in (arms if r % 2 == 0 else arms[::-1]):
        env = dict(os.environ)
        env.update(env_add)
        if lib != "default":
            env["PANSIM_HIP_LIBRARY"] = os.path.join(ROOT, lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--no-cpu-baseline", "--no-other-configs",
                              "--steps", "200" if cfg in ("cfg2", "cfg3", "authors") else "20", "--warmup", "10", "--max_distances", "1000"] + extra,
                             capture_output=True, text=True, env=env).stdout.strip().splitlines()
        try:
            d = json.loads(out[-1])
            res[name]["sweep_ms"].append(round(d["roofline"]["avg_launch_ms"], 4))
            res[name]["period_ms"].append(round(d["ms_per_step"], 4))
            res[name]["value"].append(round(d["value"], 1))
        except (IndexError, ValueError, KeyError):
            res[name].setdefault("failed", 0)
            res[name]["failed"] += 1
for name in res:
    for k in ("sweep_ms", "period_ms", "value"):
        if res[name][k]:
            res[name][k + "_median"] = statistics.median(res[name][k])
print(json.dumps({"config": cfg, "extra": extra, "rounds": rounds, "arms": res}))
