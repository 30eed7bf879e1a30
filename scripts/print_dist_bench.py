#!/usr/bin/env python3
"""--print_dist as a workload on its own (bench.py's other_configs.cfg2_print_dist): python scripts/print_dist_bench.py [generations]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

ctx = bench.Ctx(torch, None, 1, 0, 0, "nccl")
gens = int(sys.argv[1]) if len(sys.argv) > 1 else 100
print(json.dumps(bench.measure_print_dist(ctx, dict(bench.CONFIGS["cfg2"][0]), bench.CONFIGS["cfg2"][1], gens)))
