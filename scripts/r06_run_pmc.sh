#!/bin/bash
# Round 6: instruction / LDS / traffic counters of the wave sweep (cfg2's matrix, sweep alone) and of the window sweep (one rank of 8
# at cfg4, in the loop) for the round-5 library and the bit-sliced level-1 build on ONE box: scripts/r06_run_pmc.sh OUT
OUT=$1; REPO=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in v1 new; do
  if [ $lib = v1 ]; then export PANSIM_HIP_LIBRARY=$REPO/pansim_amd/libpansim_hip_v1.so; else unset PANSIM_HIP_LIBRARY; fi
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/wave_${lib}_$i -- python3 $REPO/scripts/sweep_only.py 8 > $REPO/$OUT/wave_${lib}_$i.log 2>&1
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/window_${lib}_$i -- python3 $REPO/scripts/loop_only.py 6 0 65536 150000 0.05 0.05 > $REPO/$OUT/window_${lib}_$i.log 2>&1
  done
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for kern, flt in (("wave", "core_sweep_wave_kernel"), ("window", "core_sweep_window_kernel")):
    for lib in ("v1", "new"):
        cnt = collections.defaultdict(list); dur = []
        for d in sorted(glob.glob("%s/%s_%s_*/" % (out, kern, lib))):
            for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
                rows = [r for r in csv.DictReader(open(f)) if flt in r["Kernel_Name"] and "true>(core_sweep_args" not in r["Kernel_Name"].replace("true, true>", "X") or (kern == "wave" and flt in r["Kernel_Name"])]
                t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
                dur += t[1:]
            for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
                per = collections.defaultdict(list)
                for r in csv.DictReader(open(f)):
                    if flt in r["Kernel_Name"]:
                        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k, v in per.items():
                    cnt[k] += v
        # the window sweep comes as two launches per generation (WIDE = false / true): the second one leaves at once --
        # keep the launches whose value is at least a tenth of the largest one
        mean = {}
        for k, v in cnt.items():
            big = [x for x in v if x >= 0.1 * max(v)] if max(v) > 0 else v
            big = big[1:] if len(big) > 1 else big
            mean[k] = sum(big) / len(big)
        dbig = [x for x in dur if x >= 0.1 * max(dur)] if dur else []
        res["%s_%s" % (kern, lib)] = {"counters_mean_per_launch": mean, "mean_kernel_us_under_pmc": sum(dbig) / max(1, len(dbig))}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/wave_*/ $OUT/window_*/
