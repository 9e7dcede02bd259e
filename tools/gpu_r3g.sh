#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_vit.py tests/test_gpu_parity_full.py tests/test_gpu_stage1.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3g_pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED|Error|LN folded|outlier|^E |^\[end|^\[cat" gpurun_out/r3g_pytest.log | cut -c1-300 | tail -14
timeout 300 python tools/bench_vit.py 2>&1 | tail -3
