#!/bin/bash
# round 6, run h: cfg2 at 4 against 5 workgroups per CU; the final default line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V1=pansim_amd/libpansim_hip_v1.so
python scripts/ab_matrix.py cfg2 3 v1=$V1 n4=default,PANSIM_SWEEP_BLOCKS_PER_CU=4 n5=default,PANSIM_SWEEP_BLOCKS_PER_CU=5 > gpurun_out/r06_h_ab_cfg2.json 2>gpurun_out/r06_h_err.txt; cat gpurun_out/r06_h_ab_cfg2.json
python scripts/ab_matrix.py cfg3 2 v1=$V1 n7=default n5=default,PANSIM_SWEEP_BLOCKS_PER_CU=5 > gpurun_out/r06_h_ab_cfg3.json 2>>gpurun_out/r06_h_err.txt; cat gpurun_out/r06_h_ab_cfg3.json
