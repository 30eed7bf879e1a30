#!/bin/bash
# round 6, run e: outputs in draw order (row k = the child of draw k), the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r06_e_gpu_tests.log 2>&1
tail -25 gpurun_out/r06_e_gpu_tests.log
