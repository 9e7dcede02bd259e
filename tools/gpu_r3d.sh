#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fit.py -m gpu -q -s -p no:cacheprovider -k "lazy or batched_fused" > gpurun_out/r3d_pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED|Error|lazy vs|assert|^E " gpurun_out/r3d_pytest.log | cut -c1-220 | tail -20
