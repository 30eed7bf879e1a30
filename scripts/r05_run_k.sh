#!/bin/bash
O=gpurun_out/r05_k; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "hgt or acc_operators or loop or multi or config or stress or authors" > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
run () { tag=$1; cfg=$2; shift; shift; env "$@" python bench.py --config $cfg --no-cpu-baseline 2> /dev/null > $O/$tag.json; python - $O/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "gen/s %.1f period %.4f sweep %.4f exposed %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["ms_per_step"]-d["roofline"]["avg_launch_ms"]))
PY
}
for rep in 1 2; do
run cfg3_b2_$rep cfg3 X=1
run cfg3_b3_$rep cfg3 PANSIM_HIP_LIBRARY=pansim_amd/libpansim_hip_bin3.so
run cfg3_b4_$rep cfg3 PANSIM_HIP_LIBRARY=pansim_amd/libpansim_hip_bin4.so
done
run s8_b4 cfg4_shard8 PANSIM_HIP_LIBRARY=pansim_amd/libpansim_hip_bin4.so
run s8_b2 cfg4_shard8 X=1
