#!/bin/bash
# PMC passes over the matrix-core D-avg kernel (scripts/davg_bench.py): scripts/pmc_davg.sh OUTDIR [N] [G]
OUT=$1; N=${2:-65536}; G=${3:-4000}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/pmc_$i -- python3 $REPO/scripts/davg_bench.py $N $G > $REPO/$OUT/pmc_$i.log 2>&1
done
cd $REPO
python3 scripts/collect_pmc.py $OUT $OUT/summary_nb2.json "acc_average_distance_mfma_kernel<2u>" "N=$N G=$G"
python3 scripts/collect_pmc.py $OUT $OUT/summary_nb1.json "acc_average_distance_mfma_kernel<1u>" "N=$N G=$G"
