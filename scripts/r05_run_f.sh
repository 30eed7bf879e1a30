#!/bin/bash
# round 5, GPU call f: the whole GPU suite, smoke, the default bench line
O=gpurun_out/r05_f; mkdir -p $O
( time python -m pytest tests -x -q -m gpu ) > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value %.1f ms %.4f frac %.4f dist %.3f'%(d['value'], d['ms_per_step'], d['roofline']['frac'], d['distance_ms']))
for k,o in (d.get('other_configs') or {}).items():
    if 'error' in o: print('  ',k,o); continue
    print('  %-12s gen/s %8.1f period %.4f sweep %.4f frac %.4f exposed %.4f dist_ms %.2f'%(k,o['generations_per_s'],o['ms_per_generation'],o['sweep']['avg_launch_ms'],o['sweep']['frac'],o['exposed_non_sweep_ms'],o['distance_ms']))
PY
