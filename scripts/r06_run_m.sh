#!/bin/bash
# round 6, run m: the very last code (block sweep with in-register mutations): the whole GPU suite and the default line once more
cd "$(dirname "$0")/.."
O=gpurun_out/r06_m; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python -m pytest tests -q -m gpu ) > $O/gpu_tests.log 2>&1; grep -n "passed\|failed" $O/gpu_tests.log | tail -2
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python scripts/stress_parity.py 300 64 2>&1 | tail -1
