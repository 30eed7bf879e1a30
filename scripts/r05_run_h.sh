#!/bin/bash
# round 5, GPU call h: gap passes on the core stream + early exit of the window sweep's second launch
O=gpurun_out/r05_h; mkdir -p $O
( time python -m pytest tests -x -q -m gpu ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
run () { tag=$1; cfg=$2; shift; shift; env "$@" python bench.py --config $cfg --no-cpu-baseline 2> /dev/null > $O/$tag.json; python - $O/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "gen/s %.1f period %.4f sweep %.4f exposed %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["ms_per_step"]-d["roofline"]["avg_launch_ms"]))
PY
}
for rep in 1 2; do
run cfg3_gap1_$rep cfg3 PANSIM_GAP_ON_CORE_STREAM=1
run cfg3_gap0_$rep cfg3 PANSIM_GAP_ON_CORE_STREAM=0
run s8_gap1_$rep cfg4_shard8 PANSIM_GAP_ON_CORE_STREAM=1
run s8_gap0_$rep cfg4_shard8 PANSIM_GAP_ON_CORE_STREAM=0
done
run cfg5_gap1 cfg5pop PANSIM_GAP_ON_CORE_STREAM=1
run cfg5_gap0 cfg5pop PANSIM_GAP_ON_CORE_STREAM=0
run cfg4_gap1 cfg4 PANSIM_GAP_ON_CORE_STREAM=1
