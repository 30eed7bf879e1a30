#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes only, over the generation loop (scripts/loop_only.py): scripts/pmc_fetch_only.sh OUTDIR N L [n] [kernel filter]
# (the library under test: PANSIM_HIP_LIBRARY, tuning through the PANSIM_* environment)
OUT=$1; N=$2; L=$3; n=${4:-4}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/pmc_$i -- python3 $REPO/scripts/loop_only.py $n 3000 $N $L > $REPO/$OUT/pmc_$i.log 2>&1
done
cd $REPO
python3 scripts/collect_pmc.py $OUT $OUT/summary.json ${5:-core_sweep_window_kernel} "N=$N L=$L lam_mut=60000 lam_hr=3000, generation loop"
