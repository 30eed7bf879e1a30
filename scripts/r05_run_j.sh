#!/bin/bash
O=gpurun_out/r05_j; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "hgt or acc_operators or loop or multi or config or stress or authors" 2>&1 | tail -4
run () { tag=$1; cfg=$2; shift; shift; env "$@" python bench.py --config $cfg --no-cpu-baseline 2> /dev/null > $O/$tag.json; python - $O/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "gen/s %.1f period %.4f sweep %.4f exposed %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["ms_per_step"]-d["roofline"]["avg_launch_ms"]))
PY
}
for rep in 1 2 3; do run cfg3_$rep cfg3 X=1; done
for rep in 1 2; do run s8_$rep cfg4_shard8 X=1; done
run cfg5 cfg5pop X=1
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/cfg3prof -- python3 $R/bench.py --config cfg3 --no-cpu-baseline > /dev/null 2>&1
cd $R; f=$(ls $O/cfg3prof/*/*kernel_stats.csv | head -1); head -6 $f | cut -c1-120
