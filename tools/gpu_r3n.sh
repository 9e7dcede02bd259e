#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_kernels.py -m gpu -q -s -p no:cacheprovider -k "lazy or long_run or adam" 2>&1 | grep -E "passed|failed|2500 steps|Error|^E  " | cut -c1-220 | head -20
