#!/bin/bash
# round 6, run i: the narrow launch of the light HGT kernel beside the (shorter) sweep: events per thread at cfg2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python scripts/ab_matrix.py cfg2 3 e112=default e96=default,PANSIM_HGT_EVENTS_PER_THREAD=96 e80=default,PANSIM_HGT_EVENTS_PER_THREAD=80 e64=default,PANSIM_HGT_EVENTS_PER_THREAD=64 e48=default,PANSIM_HGT_EVENTS_PER_THREAD=48 > gpurun_out/r06_i_ab_cfg2_ept.json 2>gpurun_out/r06_i_err.txt; cat gpurun_out/r06_i_ab_cfg2_ept.json
