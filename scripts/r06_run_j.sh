#!/bin/bash
# round 6, run j: the wave sweep with the next batch's rows requested behind the push (variant build) against the default, by workgroups per CU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
PF=pansim_amd/libpansim_hip_pf.so
PANSIM_HIP_LIBRARY=$(pwd)/$PF timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
python scripts/ab_matrix.py cfg2 3 n4=default pf3=$PF,PANSIM_SWEEP_BLOCKS_PER_CU=3 pf4=$PF,PANSIM_SWEEP_BLOCKS_PER_CU=4 pf5=$PF,PANSIM_SWEEP_BLOCKS_PER_CU=5 > gpurun_out/r06_j_ab_cfg2_prefetch.json 2>gpurun_out/r06_j_err.txt; cat gpurun_out/r06_j_ab_cfg2_prefetch.json
python scripts/ab_matrix.py cfg3 3 n4=default pf3=$PF,PANSIM_SWEEP_BLOCKS_PER_CU=3 pf4=$PF,PANSIM_SWEEP_BLOCKS_PER_CU=4 > gpurun_out/r06_j_ab_cfg3_prefetch.json 2>>gpurun_out/r06_j_err.txt; cat gpurun_out/r06_j_ab_cfg3_prefetch.json
