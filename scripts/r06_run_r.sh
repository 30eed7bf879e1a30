#!/bin/bash
# round 6, run r: signed FP4 all-pairs form (3 products per site): parity, per-form times, both libraries on one box, stress
cd "$(dirname "$0")/.."; O=gpurun_out/r06_r; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x ) > $O/parity.log 2>&1; grep -n "passed\|failed" $O/parity.log | tail -2
python scripts/allpairs_bench.py 8192 1200000 4194304 2,7,6 2>/dev/null | tail -1 | tee $O/allpairs_8192.json
python scripts/allpairs_bench.py 1000 1200000 100000 2,7,6 2>/dev/null | tail -1 | tee $O/allpairs_1000.json
python scripts/allpairs_bench.py 3000 300000 1000000 2,7,6 2>/dev/null | tail -1 | tee $O/allpairs_3000.json
for r in 1 2; do for lib in r6m default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$PWD/pansim_amd/libpansim_hip_$lib.so; fi
  python bench.py --config cfg2 --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 2>/dev/null | tail -1 > $O/cfg2_${lib}_$r.json
  python bench.py --config cfg5pop --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/cfg5pop_${lib}_$r.json
  for c in cfg2 cfg5pop; do python - $O/${c}_${lib}_$r.json $lib $c <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], "value", round(d["value"],1), "distance_ms", round(d["distance_ms"],4), "Mpairs/s", round(d["mpairs_per_s"],1), d.get("distance_roofline",{}).get("kernel_ms"), d.get("distance_roofline",{}).get("frac"))
P
  done
done; done
for lib in r6m default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$PWD/pansim_amd/libpansim_hip_$lib.so; fi
  python scripts/print_dist_bench.py 200 2>/dev/null | tail -1 > $O/print_dist_$lib.json; echo $lib print_dist; cut -c250-330 $O/print_dist_$lib.json
done
unset PANSIM_HIP_LIBRARY
python scripts/stress_parity.py 150 64 2>&1 | tail -1
