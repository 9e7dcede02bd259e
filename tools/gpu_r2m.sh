#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fit.py -m gpu -q -p no:cacheprovider -k "any_batch or fused" 2>&1 | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2m_smoke.log 2>&1; echo "smoke rc=$?"; grep smoke gpurun_out/r2m_smoke.log
