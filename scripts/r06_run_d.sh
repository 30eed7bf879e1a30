#!/bin/bash
# round 6, run d: symbol-decided mutations applied in registers (byte-lane plane layout), residual cells only in the queue, HR list
# per batch in the window sweep: parity, then same-box A/B matrices against the round-5 library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r06_d_parity.log 2>&1
tail -4 gpurun_out/r06_d_parity.log
V1=pansim_amd/libpansim_hip_v1.so
python scripts/ab_matrix.py cfg2 3 v1=$V1 n5=default,PANSIM_SWEEP_BLOCKS_PER_CU=5 n6=default,PANSIM_SWEEP_BLOCKS_PER_CU=6 n7=default,PANSIM_SWEEP_BLOCKS_PER_CU=7 > gpurun_out/r06_d_ab_cfg2.json 2>gpurun_out/r06_d_err.txt; cat gpurun_out/r06_d_ab_cfg2.json
python scripts/ab_matrix.py cfg3 3 v1=$V1 n=default n6=default,PANSIM_SWEEP_BLOCKS_PER_CU=6 > gpurun_out/r06_d_ab_cfg3.json 2>>gpurun_out/r06_d_err.txt; cat gpurun_out/r06_d_ab_cfg3.json
python scripts/ab_matrix.py authors 3 v1=$V1 n=default n6=default,PANSIM_SWEEP_BLOCKS_PER_CU=6 > gpurun_out/r06_d_ab_authors.json 2>>gpurun_out/r06_d_err.txt; cat gpurun_out/r06_d_ab_authors.json
python scripts/ab_matrix.py cfg4_shard8 3 v1=$V1 n6=default n5=default,PANSIM_WINDOW_BPC=5 w7=pansim_amd/libpansim_hip_w7.so > gpurun_out/r06_d_ab_cfg4_shard8.json 2>>gpurun_out/r06_d_err.txt; cat gpurun_out/r06_d_ab_cfg4_shard8.json
python scripts/ab_matrix.py cfg5pop 3 v1=$V1 n6=default w7=pansim_amd/libpansim_hip_w7.so > gpurun_out/r06_d_ab_cfg5pop.json 2>>gpurun_out/r06_d_err.txt; cat gpurun_out/r06_d_ab_cfg5pop.json
