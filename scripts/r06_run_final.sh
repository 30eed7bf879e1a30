#!/bin/bash
# round 6, the code as committed last: per-configuration rocprofv3 summaries + bench lines (scripts/profile_round.sh), the bare two-rank
# line on one GPU, the command line with --print_matrices at a 8192 x 200000 matrix
cd "$(dirname "$0")/.."
O=gpurun_out/r06_final; mkdir -p $O
bash scripts/profile_round.sh r06final > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bare_2ranks_one_gpu.json 2> $O/bare_2ranks.err; echo rc=$?
( time pansim_amd/pansim --pop_size 8192 --core_size 200000 --n_gen 2 --max_distances 100000 --print_matrices --outpref /tmp/pm_final ) > $O/cli_print_matrices.log 2>&1; ls -la /tmp/pm_final* >> $O/cli_print_matrices.log; rm -f /tmp/pm_final*; tail -8 $O/cli_print_matrices.log
