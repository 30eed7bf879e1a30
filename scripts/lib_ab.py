#!/usr/bin/env python3
"""A/B of two builds of libpansim_hip.so inside the generation loop: alternates child processes (one per library and
round), each reporting the in-loop HIP-event time per sweep launch: python scripts/lib_ab.py LIB_A LIB_B [rounds] [config]"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
res = {lib: [] for lib in libs}
per = {lib: [] for lib in libs}
for r in range(rounds):
    for lib in (libs if r % 2 == 0 else libs[::-1]):
        env = dict(os.environ)
        if lib != "default":
            env["PANSIM_HIP_LIBRARY"] = os.path.join(ROOT, lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--no-cpu-baseline", "--no-other-configs",
                              "--steps", "200" if cfg in ("cfg2", "cfg3") else "20", "--warmup", "10", "--max_distances", "1000"],
                             capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        res[lib].append(d["roofline"]["avg_launch_ms"])
        per[lib].append(d["ms_per_step"])
print(json.dumps({lib: {"sweep_ms_median": round(statistics.median(v), 4), "sweep_ms_all": [round(x, 4) for x in v],
                        "period_ms_median": round(statistics.median(per[lib]), 4)} for lib, v in res.items()}))
