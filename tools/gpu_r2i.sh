#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stage2.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r2i_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2i_pytest.log
grep -E "passed|failed|FAILED|Error|rc=|stage-2|assert" gpurun_out/r2i_pytest.log | tail -40
