#!/bin/bash
# round 6, run l: the block sweep with the symbol-decided mutations in registers (residual-only queue): parity, then against round 5's library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/r06_l_parity.log 2>&1; tail -3 gpurun_out/r06_l_parity.log
python scripts/stress_parity.py 150 63 2>&1 | tail -2
for lib in pansim_amd/libpansim_hip_v1.so default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$(pwd)/$lib; fi
  echo "== $lib"; python scripts/block_phase_bench.py 8192x150000 2048x600000 65536x18000
done > gpurun_out/r06_l_block_sweep.txt 2>&1
cat gpurun_out/r06_l_block_sweep.txt
