#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_fit.py tests/test_gpu_stage1.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r2y_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^\[|passed|failed|FAILED|Error" gpurun_out/r2y_pytest.log | cut -c1-330 | tail -12
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-fit > gpurun_out/r2y_bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2y_bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"],3), d["config"]["t_extract_s_serial"], d["config"]["t_fit_s_serial"], {n:(round(v.get("avg_us",0),1)) for n,v in d.get("kernels",{}).items()})
PY
done
