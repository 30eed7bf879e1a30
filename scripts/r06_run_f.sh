#!/bin/bash
# round 6, run f: window sweep -- registers per wave x workgroups per CU (what fits beside it), and --print_dist with the pair list mapped on the device
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python scripts/print_dist_bench.py 200 > gpurun_out/r06_f_print_dist.json 2>gpurun_out/r06_f_err.txt; cat gpurun_out/r06_f_print_dist.json
W7=pansim_amd/libpansim_hip_w7.so
python scripts/ab_matrix.py cfg5pop 3 n6=default n5=default,PANSIM_WINDOW_BPC=5 r72b6=$W7,PANSIM_WINDOW_BPC=6 r72b5=$W7,PANSIM_WINDOW_BPC=5 > gpurun_out/r06_f_ab_cfg5pop.json 2>>gpurun_out/r06_f_err.txt; cat gpurun_out/r06_f_ab_cfg5pop.json
python scripts/ab_matrix.py cfg4_shard8 3 n5=default n6=default,PANSIM_WINDOW_BPC=6 r72b6=$W7,PANSIM_WINDOW_BPC=6 r72b5=$W7,PANSIM_WINDOW_BPC=5 > gpurun_out/r06_f_ab_cfg4_shard8.json 2>>gpurun_out/r06_f_err.txt; cat gpurun_out/r06_f_ab_cfg4_shard8.json
python scripts/ab_matrix.py cfg4 2 n6=default n5=default,PANSIM_WINDOW_BPC=5 r72b6=$W7,PANSIM_WINDOW_BPC=6 > gpurun_out/r06_f_ab_cfg4.json 2>>gpurun_out/r06_f_err.txt; cat gpurun_out/r06_f_ab_cfg4.json
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_multi.py -q -m gpu -x 2>&1 | tail -3
