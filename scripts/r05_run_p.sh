#!/bin/bash
O=gpurun_out/r05_p; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "hgt or acc_operators or loop or multi or config or stress" > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
run () { tag=$1; cfg=$2; shift; shift; env "$@" python bench.py --config $cfg --no-cpu-baseline 2> /dev/null > $O/$tag.json; python - $O/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "gen/s %.1f period %.4f sweep %.4f exposed %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["ms_per_step"]-d["roofline"]["avg_launch_ms"]))
PY
}
for rep in 1 2; do run s8_$rep cfg4_shard8 X=1; run cfg3_$rep cfg3 X=1; done
run cfg5 cfg5pop X=1
run cfg4 cfg4 X=1
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/s8prof -- python3 $R/bench.py --config cfg4_shard8 --no-cpu-baseline > /dev/null 2>&1
cd $R; f=$(ls $O/s8prof/*/*kernel_stats.csv | head -1); grep "apply\|reduce\|donor_bin" $f | cut -c1-150
