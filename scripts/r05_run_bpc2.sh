#!/bin/bash
# Round 5: (a) where should the sweep and the HGT chain start taking turns (PANSIM_HEAVY_HGT) -- HGT_rate 0.1 / 0.15 / 0.2 at N = 1000;
# (b) cfg3 and cfg2 + competition by workgroups per CU of the wave sweep
one() { python3 bench.py "$@" --no-cpu-baseline --no-other-configs --steps 150 --warmup 10 --max_distances 1000 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'tag': '$TAG', 'gen_s': round(d['value'],1), 'sweep_ms': round(d['roofline']['avg_launch_ms'],4), 'period_ms': round(d['ms_per_step'],4)}))"; }
for rep in 1 2; do
for hgt in 0.08 0.1 0.15 0.2; do
  for heavy in 0 1; do
    TAG="hgt $hgt heavy $heavy bpc 6" PANSIM_HEAVY_HGT=$heavy PANSIM_SWEEP_BLOCKS_PER_CU=6 one --config cfg2 --HGT_rate $hgt
  done
done
for bpc in 5 6 7; do TAG="cfg3 bpc $bpc" PANSIM_SWEEP_BLOCKS_PER_CU=$bpc one --config cfg3; done
for bpc in 4 5 6 7; do TAG="cfg2 comp10 bpc $bpc" PANSIM_SWEEP_BLOCKS_PER_CU=$bpc one --config cfg2 --competition_strength 10; done
done
