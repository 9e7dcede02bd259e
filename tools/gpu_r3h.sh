#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_parity_full.py -m gpu -q -s -p no:cacheprovider -k "layernorm_folded or outlier or forward_vs_oracle or wrapper" > gpurun_out/r3h_pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED|Error|LN folded|outlier|^E " gpurun_out/r3h_pytest.log | cut -c1-300 | tail -10
timeout 300 python tools/bench_vit.py 2>&1 | tail -3
