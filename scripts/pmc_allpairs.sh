#!/bin/bash
# PMC passes over the all-pairs matrix-core kernel at BASELINE configs[4]'s population (N = 8192, L = 1.2 M) and over D-avg's phase 1 at
# N = 65536 (whole + a shard of 8), separate passes as the guide prescribes: scripts/pmc_allpairs.sh OUTDIR
OUT=$1; REPO=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" \
         "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/pmc_a$i -- python3 $REPO/scripts/allpairs_bench.py 8192 1200000 4194304 2 > $REPO/$OUT/pmc_a$i.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/pmc_d$i -- python3 $REPO/scripts/davg_bench.py 65536 4000 two_phase_nb2 > $REPO/$OUT/pmc_d$i.log 2>&1
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
def key(n, grid):
    if 'core_allpairs_mfma_fp4_kernel' in n: return 'allpairs_fp4_N8192_L1200000'
    if 'intersections' in n: return 'davg_phase1_whole' if int(grid) >= 65536 else 'davg_phase1_shard_of_8'
    if 'from_counts' in n: return 'davg_phase2_whole' if int(grid) > 200000 else 'davg_phase2_shard_of_8'
    return None
for f in glob.glob(out + '/pmc_*/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '0')))
        if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out + '/pmc_*/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '0')))
        if k: dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
res = {}
for k in agg:
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}
    us = sum(dur[k]) / max(1, len(dur[k]))
    d = {"counters_mean_per_launch": m, "mean_kernel_us_under_pmc": us}
    if "GRBM_GUI_ACTIVE" in m and us > 0:
        d["effective_clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8.0 / us / 1e3      # (the counter is the sum over the 8 XCDs)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            d["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (m["GRBM_GUI_ACTIVE"] / 8.0)
    res[k] = d
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
for k, d in res.items():
    print(k, round(d["mean_kernel_us_under_pmc"], 1), "us", {x: d[x] for x in ("effective_clock_GHz", "mfma_busy_fraction") if x in d})
PY
find $OUT -name "pmc_*" -maxdepth 1 -type d | xargs rm -rf
