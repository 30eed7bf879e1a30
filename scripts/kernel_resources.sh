#!/bin/bash
# VGPR / SGPR / occupancy / spills of every kernel of libpansim_hip.so (compiler remarks): scripts/kernel_resources.sh [filter]
cd "$(dirname "$0")/../pansim_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -Rpass-analysis=kernel-resource-usage -c pansim_capi.hip -o /dev/null 2>&1 | python3 -c '
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
import subprocess
for name, r in rows.items():
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(.*", "", dem)
    if flt and flt not in dem:
        continue
    print("%-95s vgpr %3s agpr %3s sgpr %3s occ %s scratch %s sspill %s vspill %s" % (dem[:95], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
          r.get("Occupancy [waves/SIMD]"), r.get("ScratchSize [bytes/lane]"), r.get("SGPRs Spill"), r.get("VGPRs Spill")))
' "$1"
