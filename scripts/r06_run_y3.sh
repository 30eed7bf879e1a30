#!/bin/bash
# round 6, run y3: phase 2 of D-avg with 4 columns per staging thread: parity, times per (IB, QJ), kernel trace
cd "$(dirname "$0")/.."; O=gpurun_out/r06_y3; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x ) > $O/parity.log 2>&1; grep -n "passed\|failed" $O/parity.log | tail -2
F=two_phase_nb2,two_phase_nb2_ib32_qj8,two_phase_nb2_ib16_qj8,two_phase_nb2_ib32_qj4,two_phase_nb2_ib16_qj4,matrix_cores_nb2
for n in 65536 32768 16384; do python scripts/davg_bench.py $n 4000 $F 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['N'], {k.replace('two_phase_nb2','tp'):round(v,3) for k,v in d.items() if k.endswith('_ms')}, all(v for k,v in d.items() if k.endswith('equal')))" | tee -a $O/davg_ib_qj.txt; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 scripts/davg_bench.py 65536 4000 two_phase_nb2_ib32_qj8,two_phase_nb2_ib16_qj8,two_phase_nb2_ib32_qj4,two_phase_nb2_ib16_qj4 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/davg_kernel_stats.csv; grep "from_counts" $O/davg_kernel_stats.csv | cut -c1-70,190-290
rm -rf $O/prof
python scripts/stress_parity.py 100 95 2>&1 | tail -1
