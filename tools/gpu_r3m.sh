#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fit.py -m gpu -q -p no:cacheprovider -k "lazy or long_run or fused_row or batched_fused" 2>&1 | tail -2
timeout 300 python tools/bench_fit_knobs.py 2>&1 | tail -4
