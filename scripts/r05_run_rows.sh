#!/bin/bash
# Round 5: rows per wave trip (PANSIM_SWEEP_ROWS) x workgroups per CU of the wave sweep after the scan push, cfg2 in the loop
one() { python3 bench.py "$@" --no-cpu-baseline --no-other-configs --steps 150 --warmup 10 --max_distances 1000 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'tag': '$TAG', 'gen_s': round(d['value'],1), 'sweep_ms': round(d['roofline']['avg_launch_ms'],4), 'period_ms': round(d['ms_per_step'],4)}))"; }
for rep in 1 2; do
for rows in 2 3 4; do for bpc in 5 6 7; do
  TAG="cfg2 rows $rows bpc $bpc" PANSIM_SWEEP_ROWS=$rows PANSIM_SWEEP_BLOCKS_PER_CU=$bpc one --config cfg2
done; done
done
