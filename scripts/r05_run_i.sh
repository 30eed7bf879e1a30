#!/bin/bash
# round 5, GPU call i: host buffers, then the round's rocprofv3 summaries and the driver-style lines
O=gpurun_out/r05_i; mkdir -p $O
for rep in 1 2; do python bench.py --config cfg4_shard8 --no-cpu-baseline 2> /dev/null > $O/s8_$rep.json; python - $O/s8_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "gen/s %.1f period %.4f sweep %.4f exposed %.4f host %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["ms_per_step"]-d["roofline"]["avg_launch_ms"], {k:round(v,3) for k,v in d["host_half_ms"].items() if isinstance(v,float)}))
PY
done
bash scripts/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bare_2ranks_one_gpu.json 2> $O/bare_2ranks.err; echo rc=$?
