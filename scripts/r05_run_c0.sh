#!/bin/bash
# PMC study first (short passes), then the run-c experiments
bash scripts/pmc_hr_study.sh gpurun_out/pmc_hr_r05b 6 2>&1 | tail -3
bash scripts/r05_run_c.sh
