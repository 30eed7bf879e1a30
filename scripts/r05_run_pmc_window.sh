#!/bin/bash
# Round 5, after the scan push: instruction counters and clock of the window sweep (one rank of 8 at cfg4, HR on), the empty WIDE launch left out
OUT=$1; REPO=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/pmc_$i -- python3 $REPO/scripts/loop_only.py 6 0 65536 150000 0.05 0.05 > $REPO/$OUT/pmc_$i.log 2>&1
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
cnt = collections.defaultdict(list); dur = []
for d in sorted(glob.glob(out + "/pmc_*/")):
    t = {}
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "core_sweep_window_kernel" in r["Kernel_Name"] and "false>(core_sweep_args" in r["Kernel_Name"]:
                t[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    dur += list(t.values())[1:]
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "core_sweep_window_kernel" in r["Kernel_Name"] and "false>(core_sweep_args" in r["Kernel_Name"]:
                cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"kernel": "core_sweep_window_kernel<gather,mutate,HR> (the WIDE = false launch), scan push build", "workload": "N=65536 L=150000 (scripts/loop_only.py: one rank of 8 at cfg4)",
       "counters_mean_per_launch": {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in cnt.items()}, "mean_kernel_us_under_pmc": sum(dur) / max(1, len(dur))}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT/pmc_*/
