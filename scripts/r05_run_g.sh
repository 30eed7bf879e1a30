#!/bin/bash
# round 5, GPU call g: kernel timelines of one rank of 8 at cfg4 and of cfg3 (what sits between two sweeps)
O=gpurun_out/r05_g; mkdir -p $O
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/s8 -- python3 $R/bench.py --config cfg4_shard8 --no-cpu-baseline > $R/$O/s8.json 2> $R/$O/s8.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/cfg3 -- python3 $R/bench.py --config cfg3 --no-cpu-baseline > $R/$O/cfg3.json 2> $R/$O/cfg3.err
cd $R
ls $O/s8/*/ | head; ls $O/cfg3/*/ | head
