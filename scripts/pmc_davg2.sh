#!/bin/bash
# PMC passes over the two D-avg forms at N = 65536 (whole population: one-kernel form and phase 1 / phase 2 of the two-phase form)
OUT=$1; REPO=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" \
         "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/pmc_$i -- python3 $REPO/scripts/davg_bench.py 65536 4000 matrix_cores_nb2,two_phase_nb2 > $REPO/$OUT/pmc_$i.log 2>&1
done
cd $REPO
python3 - $OUT <<'PY'
import csv,glob,sys,collections,json
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
def key(n,grid):
    if 'average_distance_mfma' in n: return 'one_kernel_whole' if grid=='65536' else 'one_kernel_shard'
    if 'intersections' in n: return 'phase1_whole' if grid=='65536' else 'phase1_shard'
    if 'from_counts' in n: return 'phase2_whole' if int(grid) > 200000 else 'phase2_shard'
    return None
for f in glob.glob(out+'/pmc_*/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=key(r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size','')))
        if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out+'/pmc_*/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=key(r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size','')))
        if k: dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
res={k:dict({c:sum(v)/len(v) for c,v in d.items()}, kernel_us=sum(dur[k])/len(dur[k])) for k,d in agg.items()}
json.dump(res,open(out+'/summary.json','w'),indent=1)
for k in sorted(res):
    print(k, {c:float('%.4g'%v) for c,v in sorted(res[k].items())})
PY
