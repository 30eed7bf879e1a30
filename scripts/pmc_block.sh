#!/bin/bash
# PMC passes over a sweep kernel; usage: [RUNNER=scripts/loop_only.py] scripts/pmc_block.sh OUTDIR N L [n] [kernel name filter]
# (RUNNER defaults to scripts/sweep_only.py: ps_step on a bare handle; scripts/loop_only.py runs the generation loop)
OUT=$1; N=$2; L=$3; n=${4:-4}
RUNNER=${RUNNER:-scripts/sweep_only.py}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $REPO/$OUT/counters.txt 2>&1
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
         "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" \
         "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/pmc_$i -- python3 $REPO/$RUNNER $n 3000 $N $L > $REPO/$OUT/pmc_$i.log 2>&1
done
cd $REPO
python3 scripts/collect_pmc.py $OUT $OUT/summary.json ${5:-core_sweep_block_kernel} "N=$N L=$L lam_mut=60000 lam_hr=3000"
