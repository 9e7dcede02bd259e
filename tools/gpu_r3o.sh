#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_fit.py tests/test_gpu_kernels.py tests/test_gpu_parity_full.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3o.log 2>&1; grep -E "^\[|passed|failed|FAILED|Error" gpurun_out/r3o.log | cut -c1-300 | tail -8
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-fit > gpurun_out/r3o_bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3o_bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"],3), d["config"]["t_extract_s_serial"], d["config"]["t_fit_s_serial"])
PY
