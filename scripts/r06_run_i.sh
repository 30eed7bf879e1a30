#!/bin/bash
# round 6, run i: workgroups per CU of the wave sweep once more (3 / 4 / 5 at cfg2, 3..6 at cfg3), then the narrow launch of the light HGT kernel
# beside the (shorter) sweep: events per thread at cfg2 with 4 workgroups per CU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V1=pansim_amd/libpansim_hip_v1.so
python scripts/ab_matrix.py cfg2 3 v1=$V1 n3=default,PANSIM_SWEEP_BLOCKS_PER_CU=3 n4=default,PANSIM_SWEEP_BLOCKS_PER_CU=4 n5=default,PANSIM_SWEEP_BLOCKS_PER_CU=5 > gpurun_out/r06_i_ab_cfg2.json 2>gpurun_out/r06_i_err.txt; cat gpurun_out/r06_i_ab_cfg2.json
python scripts/ab_matrix.py cfg3 3 v1=$V1 n3=default,PANSIM_SWEEP_BLOCKS_PER_CU=3 n4=default,PANSIM_SWEEP_BLOCKS_PER_CU=4 n5=default,PANSIM_SWEEP_BLOCKS_PER_CU=5 n6=default,PANSIM_SWEEP_BLOCKS_PER_CU=6 > gpurun_out/r06_i_ab_cfg3.json 2>>gpurun_out/r06_i_err.txt; cat gpurun_out/r06_i_ab_cfg3.json
export PANSIM_SWEEP_BLOCKS_PER_CU=4
python scripts/ab_matrix.py cfg2 3 e112=default e96=default,PANSIM_HGT_EVENTS_PER_THREAD=96 e80=default,PANSIM_HGT_EVENTS_PER_THREAD=80 e64=default,PANSIM_HGT_EVENTS_PER_THREAD=64 e144=default,PANSIM_HGT_EVENTS_PER_THREAD=144 > gpurun_out/r06_i_ab_cfg2_ept.json 2>>gpurun_out/r06_i_err.txt; cat gpurun_out/r06_i_ab_cfg2_ept.json
