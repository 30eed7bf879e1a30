#!/bin/bash
# round 6, run p: all-pairs partial counts in per-range slices (no atomics) + blocked D-avg rows: parity, then both libraries on one box
cd "$(dirname "$0")/.."; O=gpurun_out/r06_p; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x ) > $O/parity.log 2>&1; grep -n "passed\|failed" $O/parity.log | tail -2
for r in 1 2; do for lib in r6m default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$PWD/pansim_amd/libpansim_hip_$lib.so; fi
  python bench.py --config cfg2 --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 2>/dev/null | tail -1 > $O/cfg2_${lib}_$r.json
  python bench.py --config cfg5pop --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/cfg5pop_${lib}_$r.json
  for c in cfg2 cfg5pop; do python - $O/${c}_${lib}_$r.json $lib $c <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], "value", round(d["value"],1), "distance_ms", round(d["distance_ms"],4), "Mpairs/s", round(d["mpairs_per_s"],1), d.get("distance_roofline",{}).get("kernel_ms"))
P
  done
done; done
for lib in r6m default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$PWD/pansim_amd/libpansim_hip_$lib.so; fi
  python scripts/print_dist_bench.py 200 2>/dev/null | tail -1 > $O/print_dist_$lib.json; echo $lib print_dist; cut -c250-330 $O/print_dist_$lib.json
done
F=matrix_cores_nb2,two_phase_nb2,two_phase_nb1,two_phase_nb4
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
rocprofv3 --kernel-trace --stats -d $O/prof -o davg -- python3 scripts/davg_bench.py 65536 4000 two_phase_nb2 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/davg_kernel_stats.csv; head -8 $O/davg_kernel_stats.csv | cut -c1-200
rm -rf $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof2 -o c5 -- python3 bench.py --config cfg5pop --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $O/prof2.log 2>&1
find $O/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cfg5pop_kernel_stats.csv; head -8 $O/cfg5pop_kernel_stats.csv | cut -c1-200
rm -rf $O/prof2
