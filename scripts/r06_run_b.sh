#!/bin/bash
# round 6, run b: the whole GPU suite on the bit-sliced level-1 spec, then same-box A/B against the round-5 library
# (pansim_amd/libpansim_hip_v1.so) on every benched workload, and the access-pattern skeleton of the same box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06_b_gpu_tests.log 2>&1
tail -15 gpurun_out/r06_b_gpu_tests.log
for cfg in cfg2 cfg3 authors cfg5pop cfg4_shard8; do
  timeout 900 python scripts/lib_ab.py pansim_amd/libpansim_hip_v1.so default 3 $cfg > gpurun_out/r06_b_ab_$cfg.json 2>gpurun_out/r06_b_ab_$cfg.err
  echo $cfg; cat gpurun_out/r06_b_ab_$cfg.json
done
(cd scripts/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o sweep_skeleton sweep_skeleton.hip && for b in 5 6 7; do ./sweep_skeleton $b; done) > gpurun_out/r06_b_skeleton.txt 2>&1
tail -30 gpurun_out/r06_b_skeleton.txt
