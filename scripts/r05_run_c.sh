#!/bin/bash
# round 5, GPU call c: two-phase D-avg, CU-mask probe, cfg3 bin priority, cfg4_shard8 schedules
O=gpurun_out/r05_c; mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_rccl_fake.py -x -q -m gpu -k "average_distance or rccl or stress" 2>&1 | tail -8 > $O/tests.log; cat $O/tests.log
python scripts/davg_bench.py 65536 4000 > $O/davg_65536.json 2> $O/davg_65536.err; cat $O/davg_65536.json
python scripts/davg_bench.py 16384 4000 > $O/davg_16384.json 2>> $O/davg_65536.err; cat $O/davg_16384.json
for n in 1 2; do ./scripts/ubench/cu_mask $n 20 > $O/cu_mask_$n.json 2>&1; cat $O/cu_mask_$n.json; done
for k in 0 1 2 3; do PANSIM_HGT_BIN_PRIO=$k python bench.py --config cfg3 --no-cpu-baseline 2> /dev/null > $O/cfg3_prio$k.json; python - $O/cfg3_prio$k.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "gen/s %.1f ms %.4f sweep %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
done
run_s8 () { tag=$1; shift; env "$@" python bench.py --config cfg4_shard8 --no-cpu-baseline 2> /dev/null > $O/s8_$tag.json; python - $O/s8_$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "gen/s %.1f period %.4f sweep %.4f exposed %.4f link %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["ms_per_step"]-d["roofline"]["avg_launch_ms"], d["exchange"].get("modelled_link_ms_per_generation", 0)))
PY
}
run_s8 default X=1
run_s8 ring PANSIM_EMU_RING=1
run_s8 beside6 PANSIM_EXCHANGE_BESIDE_SWEEP=1
run_s8 beside7_free1 PANSIM_EXCHANGE_BESIDE_SWEEP=1 PANSIM_SWEEP_FREE_CUS=1
run_s8 beside7_free2 PANSIM_EXCHANGE_BESIDE_SWEEP=1 PANSIM_SWEEP_FREE_CUS=2
run_s8 free1_only PANSIM_SWEEP_FREE_CUS=1
python bench.py --config cfg4_shard8 --no-cpu-baseline --competition_strength 10 2> /dev/null > $O/s8_comp10.json; python - $O/s8_comp10.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("comp10 gen/s %.1f period %.4f sweep %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
# soft progress gate in the window sweep (variant build), same library with the gate off as the control
for g in 0 24 48 96 192; do run_s8 gate$g PANSIM_HIP_LIBRARY=pansim_amd/libpansim_hip_gate.so PANSIM_WINDOW_GATE=$g; done
