#!/bin/bash
# round 5, after the STASH = 2 build (cfg3's rates): whole GPU suite, cfg3 under rocprofv3, the default line again
O=gpurun_out/r05_s5; mkdir -p $O
( time python -m pytest tests -x -q -m gpu ) > $O/gpu_tests.log 2>&1; grep -n "passed\|failed" $O/gpu_tests.log | tail -2
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/cfg3 -- python3 $REPO/bench.py --config cfg3 --no-cpu-baseline --steps 100 --warmup 10 > $REPO/$O/cfg3.out 2> $REPO/$O/cfg3.err
cd $REPO
cp $(find $O/cfg3 -name "*kernel_stats.csv" | head -1) $O/cfg3_kernel_stats.csv; rm -rf $O/cfg3; tail -1 $O/cfg3.out > $O/cfg3_bench.json
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
