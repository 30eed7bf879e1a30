#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of scripts/sweep_only.py into profiles/<tag>_pmc_sweep.json.

Run on the GPU box (separate passes, as the microarch guide prescribes):
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" GRBM_GUI_ACTIVE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$(echo $c | cut -c1-10 | tr ' ' _) -- python3 scripts/sweep_only.py 6
  done
  python3 scripts/collect_pmc.py $OUT profiles/r01_pmc_sweep.json
"""
import collections
import csv
import glob
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
kname = sys.argv[3] if len(sys.argv) > 3 else "core_sweep_wave_kernel"      # kernel name filter
workload = sys.argv[4] if len(sys.argv) > 4 else "N=1000 L=1200000 lam_mut=60000 lam_hr=3000"
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(src + "/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(src + "/pmc_*/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
mean = {k: sum(v) / len(v) for k, v in agg.items()}
out = {"kernel": kname + "<gather,mutate,HR>", "workload": workload,
       "counters_mean_per_launch": mean, "launches_per_counter": {k: len(v) for k, v in agg.items()},
       "mean_kernel_us_under_pmc": sum(dur) / max(len(dur), 1)}
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports exactly half of a wide coalesced read stream
    out["hbm_read_bytes_per_launch"] = 2.0 * mean["FETCH_SIZE"] * 1024.0
    out["hbm_write_bytes_per_launch"] = mean["WRITE_SIZE"] * 1024.0
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
    out["source"] = ("rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over scripts/sweep_only.py; "
                     "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md HBM section)")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
