#!/bin/bash
# round 6, run a: parity of the bit-sliced level-1 spec + A/B against the round-5 library (pansim_amd/libpansim_hip_v1.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r06_a_parity.log 2>&1
tail -5 gpurun_out/r06_a_parity.log
timeout 600 python scripts/lib_ab.py pansim_amd/libpansim_hip_v1.so default 4 cfg2 > gpurun_out/r06_a_ab_cfg2.json 2>gpurun_out/r06_a_ab_cfg2.err
cat gpurun_out/r06_a_ab_cfg2.json
