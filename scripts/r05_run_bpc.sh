#!/bin/bash
# Round 5: workgroups per CU of the wave sweep inside the generation loop, by the accessory chain's load (HGT_rate): generations/s
for rep in 1 2; do
for hgt in 0.0 0.05 0.15 0.3 0.5; do
  for bpc in 4 5 6 7; do
    PANSIM_SWEEP_BLOCKS_PER_CU=$bpc python3 bench.py --config cfg2 --HGT_rate $hgt --no-cpu-baseline --no-other-configs --steps 150 --warmup 10 --max_distances 1000 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'hgt': $hgt, 'bpc': $bpc, 'gen_s': round(d['value'],1), 'sweep_ms': round(d['roofline']['avg_launch_ms'],4), 'period_ms': round(d['ms_per_step'],4)}))"
  done
done
for bpc in 4 5 6 7; do
    PANSIM_SWEEP_BLOCKS_PER_CU=$bpc python3 bench.py --config authors --no-cpu-baseline --no-other-configs --steps 150 --warmup 10 --max_distances 1000 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'cfg': 'authors', 'bpc': $bpc, 'gen_s': round(d['value'],1), 'sweep_ms': round(d['roofline']['avg_launch_ms'],4), 'period_ms': round(d['ms_per_step'],4)}))"
done
done
