#!/bin/bash
# round 6, run t: phase 1 of D-avg with its pipeline carried over the j steps; defaults as committed: parity + both libraries on one box
cd "$(dirname "$0")/.."; O=gpurun_out/r06_t; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x ) > $O/parity.log 2>&1; grep -n "passed\|failed" $O/parity.log | tail -2
F=matrix_cores_nb2,matrix_cores_nb1,two_phase_nb2,two_phase_nb1,two_phase_nb4
for r in 1 2; do for lib in r6m default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$PWD/pansim_amd/libpansim_hip_$lib.so; fi
  echo "== $lib $r"; python scripts/davg_bench.py 65536 4000 $F 2>/dev/null | tail -1 | tee $O/davg_65536_${lib}_$r.json
done; done
for lib in r6m default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$PWD/pansim_amd/libpansim_hip_$lib.so; fi
  echo "== $lib 16384"; python scripts/davg_bench.py 16384 4000 $F 2>/dev/null | tail -1 | tee $O/davg_16384_$lib.json
done
unset PANSIM_HIP_LIBRARY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 scripts/davg_bench.py 65536 4000 two_phase_nb2,matrix_cores_nb2 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/davg_kernel_stats.csv; head -5 $O/davg_kernel_stats.csv | cut -c1-230
rm -rf $O/prof
python scripts/stress_parity.py 100 64 2>&1 | tail -1
