#!/bin/bash
# rocprofv3 summaries behind the round's numbers; usage (on the GPU box, from the repo root): scripts/profile_round.sh TAG
# Leaves gpurun_out/prof_TAG/{cfg2,cfg3,cfg4,cfg4_shard8,cfg5pop,competition,cfg5cli}_kernel_stats.csv + the bench lines
# of the same commands.  (PMC passes of the sweeps: scripts/pmc_block.sh)
TAG=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, then the program and its arguments
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.out 2> $OUT/$name.err
  local f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/$name
}
run cfg2 python3 $REPO/bench.py --no-cpu-baseline --no-other-configs --steps 100 --warmup 10
run cfg3 python3 $REPO/bench.py --config cfg3 --no-cpu-baseline --steps 100 --warmup 10
run cfg4 python3 $REPO/bench.py --config cfg4 --no-cpu-baseline --steps 10 --warmup 2
run cfg4_shard8 python3 $REPO/bench.py --config cfg4_shard8 --no-cpu-baseline --steps 20 --warmup 2
run cfg5pop python3 $REPO/bench.py --config cfg5pop --no-cpu-baseline --steps 10 --warmup 2
run competition python3 $REPO/bench.py --no-cpu-baseline --no-other-configs --competition_strength 10 --steps 100 --warmup 10
run authors python3 $REPO/bench.py --config authors --no-cpu-baseline --steps 100 --warmup 10
run cfg4_shard8_comp10 python3 $REPO/bench.py --config cfg4_shard8 --no-cpu-baseline --competition_strength 10 --steps 20 --warmup 2
run cfg5cli $REPO/pansim_amd/pansim --pop_size 8192 --max_distances 33554432 --n_gen 3 --outpref /tmp/cfg5_prof
rm -f /tmp/cfg5_prof*
cd $REPO
for n in cfg2 cfg3 cfg4 cfg4_shard8 cfg5pop competition authors cfg4_shard8_comp10; do tail -1 $OUT/$n.out > $OUT/${n}_bench.json; done
ls -la $OUT | head -40
