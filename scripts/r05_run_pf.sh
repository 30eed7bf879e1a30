#!/bin/bash
# Round 5, experiment 10.4 (profiles/r05_sweep_experiments.md): the wave sweep with the next batch's rows requested behind the push, built for
# 7 / 6 waves per SIMD (variants pf7 / pf6 = -DPS_WAVE_LB=7 / 6 of a working tree that carried the prefetch; nopf = the same loop without it),
# against the kept build (trim), the sweep alone under --kernel-trace at 7 and 6 workgroups per CU.  The prefetch lost and its code was not
# committed: this script documents the command, it cannot be re-run from the repository as it is.
cd $(dirname $0)/..
scripts/sweep_trace_ab.sh gpurun_out/r05_pf_a 2 pansim_amd/libpansim_hip_trim.so pansim_amd/libpansim_hip_nopf.so pansim_amd/libpansim_hip_pf7.so pansim_amd/libpansim_hip_pf6.so 2>&1 | tail -1
export PANSIM_SWEEP_BLOCKS_PER_CU=6
scripts/sweep_trace_ab.sh gpurun_out/r05_pf_b 2 pansim_amd/libpansim_hip_trim.so pansim_amd/libpansim_hip_pf7.so pansim_amd/libpansim_hip_pf6.so 2>&1 | tail -1
