#!/bin/bash
# Round 5: 5 against 6 workgroups per CU of the wave sweep, alternating, cfg2 and HGT_rate 0.1 (the heaviest chain that still runs beside the sweep)
one() { python3 bench.py "$@" --no-cpu-baseline --no-other-configs --steps 150 --warmup 10 --max_distances 1000 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'tag': '$TAG', 'gen_s': round(d['value'],1), 'sweep_ms': round(d['roofline']['avg_launch_ms'],4), 'period_ms': round(d['ms_per_step'],4)}))"; }
for rep in 1 2 3 4; do
for bpc in 5 6; do
  TAG="cfg2 bpc $bpc" PANSIM_SWEEP_BLOCKS_PER_CU=$bpc one --config cfg2
  TAG="hgt 0.1 bpc $bpc" PANSIM_SWEEP_BLOCKS_PER_CU=$bpc one --config cfg2 --HGT_rate 0.1
done
done
