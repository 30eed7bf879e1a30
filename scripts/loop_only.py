#!/usr/bin/env python3
"""Runs only the generation loop (for rocprofv3 counter passes on the in-loop kernels, e.g. the window sweep, which needs
ps_sim's ascending parents): python scripts/loop_only.py [n] [unused] [N] [L] [HR_rate] [HGT_rate]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pansim_amd as pa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
L = int(sys.argv[4]) if len(sys.argv) > 4 else 150000
HR = float(sys.argv[5]) if len(sys.argv) > 5 else 0.05
HGT = float(sys.argv[6]) if len(sys.argv) > 6 else 0.05
sim = pa.Simulation(pa.make_params(seed=0, n_gen=n, max_distances=1000, pop_size=N, core_size=1200000, pan_genes=6000, HR_rate=HR, HGT_rate=HGT,
                                   shard_rank=0, shard_count=max(1, 1200000 // L)))
sim.run(n)
sim.sync()
print("done", n, sim.core_genome.ncols)
