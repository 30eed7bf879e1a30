#!/bin/bash
# round 6, run g: cfg2 + --competition_strength 10 (D-avg in the chain beside the wave sweep): round-5 library against this one over workgroups per CU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V1=pansim_amd/libpansim_hip_v1.so
python scripts/ab_matrix.py cfg2+competition_strength=10 3 v1=$V1 n4=default,PANSIM_SWEEP_BLOCKS_PER_CU=4 n5=default,PANSIM_SWEEP_BLOCKS_PER_CU=5 n6=default,PANSIM_SWEEP_BLOCKS_PER_CU=6 n3=default,PANSIM_SWEEP_BLOCKS_PER_CU=3 > gpurun_out/r06_g_ab_competition.json 2>gpurun_out/r06_g_err.txt; cat gpurun_out/r06_g_ab_competition.json
python scripts/ab_matrix.py authors 3 v1=$V1 n4=default,PANSIM_SWEEP_BLOCKS_PER_CU=4 n5=default,PANSIM_SWEEP_BLOCKS_PER_CU=5 > gpurun_out/r06_g_ab_authors.json 2>>gpurun_out/r06_g_err.txt; cat gpurun_out/r06_g_ab_authors.json
