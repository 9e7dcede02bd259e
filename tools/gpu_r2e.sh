#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vit.py -m gpu -q -p no:cacheprovider -k "attention or forward or wrapper" > gpurun_out/r2e_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2e_pytest.log
grep -E "passed|failed|FAILED|Error|rc=" gpurun_out/r2e_pytest.log | tail -5
timeout 300 python tools/bench_vit.py > gpurun_out/r2e_vit.log 2>&1; tail -4 gpurun_out/r2e_vit.log
