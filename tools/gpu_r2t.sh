#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fit.py -m gpu -q -s -x -p no:cacheprovider > gpurun_out/r2t_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error|fused vs|assert" gpurun_out/r2t_pytest.log | tail -12
timeout 300 python tools/bench_fit_knobs.py 2>&1 | tail -5
