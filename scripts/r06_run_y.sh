#!/bin/bash
# round 6, run y: phase 2 of D-avg with 16 individuals per workgroup where 32 leave the chip half empty: parity, times, kernel trace
cd "$(dirname "$0")/.."; O=gpurun_out/r06_y; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_multi.py -q -m gpu -x ) > $O/parity.log 2>&1; grep -n "passed\|failed" $O/parity.log | tail -2
F=two_phase_nb2,two_phase_nb2_ib32,two_phase_nb2_ib16,matrix_cores_nb2
for r in 1 2; do python scripts/davg_bench.py 65536 4000 $F 2>/dev/null | tail -1 | tee $O/davg_65536_$r.json; done
python scripts/davg_bench.py 16384 4000 $F 2>/dev/null | tail -1 | tee $O/davg_16384.json
python scripts/davg_bench.py 32768 4000 $F 2>/dev/null | tail -1 | tee $O/davg_32768.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 scripts/davg_bench.py 65536 4000 two_phase_nb2_ib32,two_phase_nb2_ib16 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/davg_kernel_stats.csv; head -6 $O/davg_kernel_stats.csv | cut -c1-60,150-260
rm -rf $O/prof
python scripts/stress_parity.py 100 81 2>&1 | tail -1
python bench.py --config cfg4_shard8 --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --competition_strength 10 2>/dev/null | tail -1 | cut -c1-300
