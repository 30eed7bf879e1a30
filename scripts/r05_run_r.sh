#!/bin/bash
O=gpurun_out/r05_r; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "multi or rccl or config4 or shard or donor" > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
for rep in 1 2 3; do python bench.py --config cfg4_shard8 --no-cpu-baseline 2> /dev/null > $O/s8_$rep.json; python - $O/s8_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "gen/s %.1f period %.4f sweep %.4f exposed %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["ms_per_step"]-d["roofline"]["avg_launch_ms"]))
PY
done
python bench.py --config cfg4_shard8 --no-cpu-baseline --competition_strength 10 2> /dev/null > $O/s8_comp10.json; python - $O/s8_comp10.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("comp10 gen/s %.1f period %.4f sweep %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
