#!/bin/bash
# round 6, run k: the fallback sweeps on the new level 1 (block sweep: two Philox blocks per slot) against round 5's library, and the window sweep at 4 workgroups per CU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for lib in pansim_amd/libpansim_hip_v1.so default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$(pwd)/$lib; fi
  echo "== $lib"; python scripts/block_phase_bench.py 8192x150000 2048x600000 65536x18000
done > gpurun_out/r06_k_block_sweep.txt 2>&1
unset PANSIM_HIP_LIBRARY
cat gpurun_out/r06_k_block_sweep.txt
python scripts/ab_matrix.py cfg5pop 3 n6=default n5=default,PANSIM_WINDOW_BPC=5 n4=default,PANSIM_WINDOW_BPC=4 > gpurun_out/r06_k_ab_cfg5pop.json 2>gpurun_out/r06_k_err.txt; cat gpurun_out/r06_k_ab_cfg5pop.json
python scripts/ab_matrix.py cfg4_shard8 3 n5=default n4=default,PANSIM_WINDOW_BPC=4 > gpurun_out/r06_k_ab_cfg4_shard8.json 2>>gpurun_out/r06_k_err.txt; cat gpurun_out/r06_k_ab_cfg4_shard8.json
