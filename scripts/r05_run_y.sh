#!/bin/bash
# round 5, final: whole GPU suite, smoke, per-configuration rocprofv3 summaries, default line, bare two-rank line
O=gpurun_out/r05_y; mkdir -p $O
( time python -m pytest tests -x -q -m gpu ) > $O/gpu_tests.log 2>&1; grep -n "passed\|failed" $O/gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile_round.sh r05y > $O/profile_round.log 2>&1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bare_2ranks_one_gpu.json 2> $O/bare_2ranks.err; echo rc=$?
