#!/bin/bash
# round 5, GPU call e: D-avg two-phase (plain stores, deeper phase-2 pipeline), D-avg ahead of the sweep
O=gpurun_out/r05_e; mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_rccl_fake.py tests/test_gpu_multi.py -x -q -m gpu -k "average_distance or inside_the_generation_loop or competition" 2>&1 | tail -6 > $O/tests.log; cat $O/tests.log
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_davg -- python3 $R/scripts/davg_bench.py 65536 4000 matrix_cores_nb2,two_phase_nb2,two_phase_nb1 > $R/$O/davg_65536.json 2> $R/$O/prof_davg.log
cd $R
cat $O/davg_65536.json | tail -1
python - $O <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/prof_davg/*/*_kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'intersections' in n or 'from_counts' in n or 'average_distance_mfma' in n:
        d[(n[:48], r.get('Grid_Size_X',r.get('Grid_Size')))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,v in d.items(): print(k, len(v), round(sum(v)/len(v),3), 'ms')
PY
python scripts/davg_bench.py 16384 4000 matrix_cores_nb1,two_phase_nb2,two_phase_nb1 | tail -1 > $O/davg_16384.json; cat $O/davg_16384.json
for a in 1 0; do PANSIM_DAVG_AHEAD=$a python bench.py --config cfg4_shard8 --no-cpu-baseline --competition_strength 10 2> /dev/null > $O/s8_comp10_ahead$a.json; python - $O/s8_comp10_ahead$a.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "gen/s %.1f period %.4f sweep %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
done
python scripts/competition_bench.py > $O/competition_cfg2.json 2>/dev/null; tail -2 $O/competition_cfg2.json | cut -c1-600
python bench.py --config authors --no-cpu-baseline 2>/dev/null > $O/authors.json; python - $O/authors.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("authors gen/s %.1f period %.4f sweep %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
