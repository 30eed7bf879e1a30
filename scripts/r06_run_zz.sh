#!/bin/bash
# round 6, the round's very last build (4 workgroups per CU, pair list mapped on the device): the whole GPU suite, the default line,
# per-configuration rocprofv3 summaries, the bare two-rank line, projected scaling points, counters of both libraries on this box
cd "$(dirname "$0")/.."
O=gpurun_out/r06_zz; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python -m pytest tests -q -m gpu ) > $O/gpu_tests.log 2>&1; grep -n "passed\|failed" $O/gpu_tests.log | tail -2
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
bash scripts/profile_round.sh r06zz > $O/profile_round.log 2>&1
python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bare_2ranks_one_gpu.json 2> $O/bare_2ranks.err; echo rc=$?
for k in 2 4 8; do python3 bench.py --config cfg4 --emulate-shard $k --no-cpu-baseline --no-other-configs --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/cfg4_shard0of$k.json; done
bash scripts/r06_run_pmc.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
