#!/bin/bash
# round 5, the very last build (table decode in the dense pass): smoke, per-configuration rocprofv3 summaries, default line, bare two-rank line,
# projected scaling points (the whole GPU suite of this build: profiles/r05_zz_gpu_tests.log)
O=gpurun_out/r05_zz; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile_round.sh r05zz > $O/profile_round.log 2>&1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bare_2ranks_one_gpu.json 2> $O/bare_2ranks.err; echo rc=$?
for k in 2 4 8; do python3 bench.py --config cfg4 --emulate-shard $k --no-cpu-baseline --no-other-configs --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/cfg4_shard0of$k.json; done
