#!/bin/bash
export TMPDIR=/tmp FIT_DTYPE=bfloat16
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fit.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 300 python tools/bench_fit_knobs.py 2>&1 | tail -4
bash tools/gpu_fit_breakdown.sh; cp gpurun_out/fit_step_breakdown.txt gpurun_out/r2v_fit_step_breakdown.txt; head -24 gpurun_out/r2v_fit_step_breakdown.txt
