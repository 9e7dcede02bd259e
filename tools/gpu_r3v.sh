#!/bin/bash
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_stage1.py tests/test_gpu_stage2.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
