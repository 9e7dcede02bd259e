#!/bin/bash
export TMPDIR=/tmp FIT_DTYPE=bfloat16
mkdir -p gpurun_out
bash tools/gpu_fit_breakdown.sh; cp gpurun_out/fit_step_breakdown.txt gpurun_out/r2u_fit_step_breakdown.txt; head -12 gpurun_out/r2u_fit_step_breakdown.txt
python tools/rocpd_stats.py $(find gpurun_out/prof_fit -name '*.db' | head -1) 2>/dev/null | head -14 | cut -c1-150
