#!/bin/bash
# round 6, run z2: the code as committed last (D-avg phase 2 at 16 individuals per workgroup, two-phase threshold 52 K rows): whole GPU suite,
# default line, stress
cd "$(dirname "$0")/.."; O=gpurun_out/r06_z2; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" | tail -1
( time python -m pytest tests -q -m gpu ) > $O/gpu_tests.log 2>&1; grep -n "passed\|failed" $O/gpu_tests.log | tail -2
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python scripts/stress_parity.py 200 91 2>&1 | tail -1
