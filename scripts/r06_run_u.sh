#!/bin/bash
# round 6, run u: the code as committed after the matrix-core work: whole GPU suite, default line, stress, D-avg and all-pairs times
cd "$(dirname "$0")/.."; O=gpurun_out/r06_u; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" | tail -1
( time python -m pytest tests -q -m gpu ) > $O/gpu_tests.log 2>&1; grep -n "passed\|failed" $O/gpu_tests.log | tail -2
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python scripts/stress_parity.py 200 64 2>&1 | tail -1
python scripts/davg_bench.py 65536 4000 matrix_cores_nb2,two_phase_nb2 2>/dev/null | tail -1 | tee $O/davg_65536.json
python scripts/allpairs_bench.py 8192 1200000 4194304 2,7,6 2>/dev/null | tail -1 | tee $O/allpairs_8192.json
