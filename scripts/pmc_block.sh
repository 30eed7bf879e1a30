#!/bin/bash
# PMC passes over the block sweep (scripts/sweep_only.py n lam_hr N L); usage: scripts/pmc_block.sh OUTDIR N L [n]
OUT=$1; N=$2; L=$3; n=${4:-4}
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
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/pmc_$i -- python3 $REPO/scripts/sweep_only.py $n 3000 $N $L > $REPO/$OUT/pmc_$i.log 2>&1
done
cd $REPO
python3 scripts/collect_pmc.py $OUT $OUT/summary.json ${5:-core_sweep_block_kernel} "N=$N L=$L lam_mut=60000 lam_hr=3000"
