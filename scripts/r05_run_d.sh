#!/bin/bash
# round 5, GPU call d: D-avg two-phase after the pitch fix (with the rocprof split of the phases), ps_multi sliced exchange
O=gpurun_out/r05_d; mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_multi.py -x -q -m gpu -k "average_distance or multi" 2>&1 | tail -6 > $O/tests.log; cat $O/tests.log
python scripts/davg_bench.py 65536 4000 matrix_cores_nb2,two_phase_nb2,two_phase_nb1 > $O/davg_65536.json 2> $O/davg.err; cat $O/davg_65536.json
python scripts/davg_bench.py 16384 4000 matrix_cores_nb1,two_phase_nb2,two_phase_nb1 > $O/davg_16384.json 2>> $O/davg.err; cat $O/davg_16384.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_davg -- python3 $R/scripts/davg_bench.py 65536 4000 two_phase_nb2,two_phase_nb1 > $R/$O/prof_davg.log 2>&1
cd $R
f=$(ls $O/prof_davg/*/*kernel_stats.csv | head -1); cp $f $O/davg_kernel_stats.csv; head -8 $O/davg_kernel_stats.csv | cut -c1-200
python bench.py --config cfg4_shard8 --no-cpu-baseline --competition_strength 10 2> /dev/null > $O/s8_comp10.json; python - $O/s8_comp10.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("comp10 gen/s %.1f period %.4f sweep %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
