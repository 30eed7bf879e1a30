#!/bin/bash
# round 6, run n: blocked 2-bit strings for the matrix-core all-pairs kernels: parity, then the distance phase against the build before
cd "$(dirname "$0")/.."; O=gpurun_out/r06_n; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x ) > $O/parity.log 2>&1; tail -3 $O/parity.log
for r in 1 2 3; do for lib in r6m default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$PWD/pansim_amd/libpansim_hip_$lib.so; fi
  python bench.py --config cfg2 --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 2>/dev/null | tail -1 > $O/cfg2_${lib}_$r.json
  python - $O/cfg2_${lib}_$r.json $lib <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], "cfg2 value", round(d["value"],1), "distance_ms", round(d["distance_ms"],4), d.get("distance_roofline",{}).get("kernel_ms"))
P
done; done
for lib in r6m default; do
  if [ $lib = default ]; then unset PANSIM_HIP_LIBRARY; else export PANSIM_HIP_LIBRARY=$PWD/pansim_amd/libpansim_hip_$lib.so; fi
  python bench.py --config cfg5pop --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/cfg5pop_$lib.json
  python - $O/cfg5pop_$lib.json $lib <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], "cfg5pop value", round(d["value"],1), "distance_ms", round(d["distance_ms"],3), d.get("distance_roofline",{}).get("kernel_ms"))
P
  python scripts/print_dist_bench.py 200 2>/dev/null | tail -1 > $O/print_dist_$lib.json; echo $lib print_dist; cut -c1-300 $O/print_dist_$lib.json
done
