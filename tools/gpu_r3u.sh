#!/bin/bash
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_fit.py tests/test_gpu_parity_full.py -m gpu -q -p no:cacheprovider -k "not baseline_shapes and not end_to_end" 2>&1 | tail -3
