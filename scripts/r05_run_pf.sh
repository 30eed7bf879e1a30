cd $(dirname $0)/..
scripts/sweep_trace_ab.sh gpurun_out/r05_pf_a 2 pansim_amd/libpansim_hip_trim.so pansim_amd/libpansim_hip_nopf.so pansim_amd/libpansim_hip_pf7.so pansim_amd/libpansim_hip_pf6.so 2>&1 | tail -1
export PANSIM_SWEEP_BLOCKS_PER_CU=6
scripts/sweep_trace_ab.sh gpurun_out/r05_pf_b 2 pansim_amd/libpansim_hip_trim.so pansim_amd/libpansim_hip_pf7.so pansim_amd/libpansim_hip_pf6.so 2>&1 | tail -1
