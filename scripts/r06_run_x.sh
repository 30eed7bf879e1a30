#!/bin/bash
# round 6, run x: stress + soaks on the code as committed last
cd "$(dirname "$0")/.."; O=gpurun_out/r06_x; mkdir -p $O
( python scripts/stress_parity.py 300 71; python scripts/stress_parity.py 300 72; python scripts/soak_determinism.py ) > $O/stress_soak.log 2>&1; tail -4 $O/stress_soak.log
