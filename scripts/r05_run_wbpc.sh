#!/bin/bash
# Round 5: workgroups per CU of the WINDOW sweep after the scan push (PANSIM_WINDOW_BPC), in the loop: one rank of 8 at cfg4, cfg5's population, cfg4 whole
one() { python3 bench.py "$@" --no-cpu-baseline --no-other-configs --max_distances 1000 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'tag': '$TAG', 'gen_s': round(d['value'],2), 'sweep_ms': round(d['roofline']['avg_launch_ms'],4), 'period_ms': round(d['ms_per_step'],4)}))"; }
for rep in 1 2; do
for bpc in 5 6 7 8; do
  TAG="cfg4_shard8 wbpc $bpc" PANSIM_WINDOW_BPC=$bpc one --config cfg4_shard8 --steps 30 --warmup 3
  TAG="cfg5pop wbpc $bpc" PANSIM_WINDOW_BPC=$bpc one --config cfg5pop --steps 30 --warmup 3
done
done
for bpc in 6 7 8; do TAG="cfg4 wbpc $bpc" PANSIM_WINDOW_BPC=$bpc one --config cfg4 --steps 8 --warmup 2; done
