for bpc in 8 7 6 5; do for ROWS in 3; do
  echo "bpc=$bpc"; PANSIM_SWEEP_BLOCKS_PER_CU=$bpc timeout 300 python bench.py --HR_rate 0.5 --HGT_rate 0.5 --no-cpu-baseline --steps 60 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
