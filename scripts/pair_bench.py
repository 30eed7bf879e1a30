#!/usr/bin/env python3
"""Distance-phase kernels at cfg2 (N=1000, L=1.2M): sampled pairs and all pairs, device-resident output."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pansim_amd as pa  # noqa: E402

N, L = 1000, 1200000
core = pa.Population(N, L, 4, True, 0.0, 0, 2000)
core.set_rates([60000.0], [3000.0])
core.mutate_alleles(0)
for mode in (1, 2):
    core.set_tuning("pair_mode", mode)
    for P in (100000, 300000):
        r1, r2 = pa.sample_pairs(0, N, P)
        out = torch.zeros(P, dtype=torch.int32, device="cuda")
        for _ in range(2):
            core.pairwise_counts_device(r1, r2, out.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            core.pairwise_counts_device(r1, r2, out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(json.dumps({"lib": os.environ.get("PANSIM_HIP_LIBRARY", "default"), "mode": mode, "P": P,
                          "ms": round(dt * 1e3, 3), "checksum": int(out.sum().item())}), flush=True)
core.close()
