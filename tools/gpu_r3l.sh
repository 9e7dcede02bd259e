#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fit.py -m gpu -q -s -p no:cacheprovider -k "long_run" 2>&1 | grep -E "passed|failed|Error|2500 steps|assert|^E " | cut -c1-250 | tail -10
