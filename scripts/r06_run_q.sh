#!/bin/bash
# round 6, run q: D-avg contraction software-pipelined (on top of run p's blocked rows): parity, both libraries on one box, kernel traces
cd "$(dirname "$0")/.."; O=gpurun_out/r06_q; mkdir -p $O
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
for lib in r6m default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$PWD/pansim_amd/libpansim_hip_$lib.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$lib -- python3 scripts/davg_bench.py 65536 4000 two_phase_nb2,matrix_cores_nb2 > $O/prof_$lib.log 2>&1
  find $O/prof_$lib -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/davg_kernel_stats_$lib.csv; head -6 $O/davg_kernel_stats_$lib.csv | cut -c1-220
  rm -rf $O/prof_$lib
done
unset PANSIM_HIP_LIBRARY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -- python3 bench.py --config cfg5pop --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $O/prof2.log 2>&1
find $O/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cfg5pop_kernel_stats.csv; head -8 $O/cfg5pop_kernel_stats.csv | cut -c1-200
rm -rf $O/prof2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3 -- python3 bench.py --config cfg2 --no-cpu-baseline --no-other-configs --steps 50 --warmup 3 > $O/prof3.log 2>&1
find $O/prof3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cfg2_kernel_stats.csv; grep -i "allpairs\|packT\|lookup\|sum_slices\|acc_pair" $O/cfg2_kernel_stats.csv | cut -c1-200
rm -rf $O/prof3
