#!/bin/bash
# The wave sweep ALONE (scripts/sweep_only.py: nothing else on the GPU) under rocprofv3 --kernel-trace, per library build, alternating
# the builds over ROUNDS rounds: average kernel duration per build.  usage: scripts/sweep_trace_ab.sh OUTDIR ROUNDS LIB...
OUT=$1; R=$2; shift; shift
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for r in $(seq 1 $R); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    export PANSIM_HIP_LIBRARY=$REPO/$lib
    timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/t_${name}_$r -- python3 $REPO/scripts/sweep_only.py 80 > $REPO/$OUT/${name}_$r.log 2>&1
  done
done
cd $REPO
python3 - $OUT $R "$@" <<'PY'
import csv, glob, json, sys, os, statistics
out, R = sys.argv[1], int(sys.argv[2])
res = {}
for lib in sys.argv[3:]:
    name = os.path.basename(lib)[:-3]
    v = []
    for r in range(1, R + 1):
        for f in glob.glob(f"{out}/t_{name}_{r}/**/*kernel_stats.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "core_sweep_wave_kernel" in row["Name"]:
                    v.append(round(float(row["AverageNs"]) / 1e3, 2))
    res[name] = {"avg_us": v, "median_us": statistics.median(v) if v else None}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT/t_*
