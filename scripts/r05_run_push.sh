#!/bin/bash
# Round 5: the scan push against the ballot push on ONE box: the sweep's skeleton (scripts/ubench/sweep_skeleton), the wave sweep
# alone (sweep_only.py under --kernel-trace) and its instruction counters, per library build.  usage: scripts/r05_run_push.sh OUTDIR LIB...
OUT=$1; shift
REPO=$(pwd)
mkdir -p $OUT
$REPO/scripts/ubench/sweep_skeleton 7 > $OUT/skeleton_7.jsonl
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  export PANSIM_HIP_LIBRARY=$REPO/$lib
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/$name/trace -- python3 $REPO/scripts/sweep_only.py 60 > $REPO/$OUT/${name}_trace.log 2>&1
  i=0
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/$name/pmc_$i -- python3 $REPO/scripts/sweep_only.py 12 > $REPO/$OUT/${name}_pmc_$i.log 2>&1
  done
done
cd $REPO
python3 - $OUT "$@" <<'PY'
import csv, glob, json, sys, os, collections
out = sys.argv[1]
res = {}
for lib in sys.argv[2:]:
    name = os.path.basename(lib)[:-3]
    d = {}
    for f in glob.glob(f"{out}/{name}/trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "core_sweep_wave_kernel" in r["Name"]:
                d["avg_us"] = float(r["AverageNs"]) / 1e3; d["calls"] = int(r["Calls"])
    cnt = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{name}/pmc_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "core_sweep_wave_kernel" in r["Kernel_Name"]:
                cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in cnt.items():
        v = v[2:] if len(v) > 4 else v
        d[k] = sum(v) / len(v)
    res[name] = d
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -mindepth 2 -type d -name "pmc_*" -prune -exec rm -rf {} \; 2>/dev/null
find $OUT -type d -name trace -prune -exec rm -rf {} \; 2>/dev/null
