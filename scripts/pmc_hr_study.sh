#!/bin/bash
# Round 5, VERDICT item 6: what does HR cost the window sweep?  PMC passes of the generation loop at N = 65536 x 150000 sites
# (one rank of 8 at cfg4) with HR on (0.05) and off (0), separate passes per counter group (MI355X_MICROARCH.md, HBM section),
# plus GRBM_GUI_ACTIVE for the clock.  (A pass with TA_BUSY_avr / TA_*_STALLED_BY_TC_CYCLES / TCC_TAG_STALL never returned on
# this pool -- 37 minutes until the call's limit -- and is left out; every pass runs under `timeout`.)  usage: scripts/pmc_hr_study.sh OUTDIR [n_generations]
OUT=$1; n=${2:-6}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for hr in 0.05 0; do
  i=0
  for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/hr${hr}/pmc_$i -- python3 $REPO/scripts/loop_only.py $n 0 65536 150000 $hr 0.05 > $REPO/$OUT/hr${hr}_pmc_$i.log 2>&1
  done
  python3 $REPO/scripts/collect_pmc.py $REPO/$OUT/hr${hr} $REPO/$OUT/hr${hr}_summary.json core_sweep_window_kernel "N=65536 L=150000 lam_hr=${hr}x lam_mut (loop_only.py)" > /dev/null
done
# the wave sweep's two clock states (weak #3): GRBM_GUI_ACTIVE / 8 / kernel time over several PROCESSES (a process keeps its state)
for k in 1 2 3 4 5 6 7 8; do
  timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $REPO/$OUT/wave/pmc_$k -- python3 $REPO/scripts/sweep_only.py 30 > $REPO/$OUT/wave_$k.log 2>&1
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = []
for d in sorted(glob.glob(out + "/wave/pmc_*")):
    cnt, dur = [], {}
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "core_sweep_wave_kernel" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "core_sweep_wave_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt.append((float(r["Counter_Value"]), dur.get(r["Dispatch_Id"])))
    cnt = [(c, t) for c, t in cnt if t][5:]          # (skip the cold launches)
    if cnt:
        us = sum(t for _, t in cnt) / len(cnt)
        gui = sum(c for c, _ in cnt) / len(cnt)
        rows.append({"process": d.split("_")[-1], "launches": len(cnt), "kernel_us": us, "GRBM_GUI_ACTIVE": gui,
                     "effective_clock_GHz": gui / 8.0 / us / 1e3})
json.dump(rows, open(out + "/wave_clock_states.json", "w"), indent=1)
print(json.dumps(rows))
PY
