#!/bin/bash
export TMPDIR=/tmp FIT_DTYPE=bfloat16
mkdir -p gpurun_out
bash tools/gpu_fit_breakdown.sh; cp gpurun_out/fit_step_breakdown.txt gpurun_out/r2r_fit_step_breakdown.txt; cat gpurun_out/r2r_fit_step_breakdown.txt | head -20
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-fit > gpurun_out/r2r_bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2r_bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"],3), d["config"]["t_extract_s_serial"], d["config"]["t_fit_s_serial"], {n:(round(v.get("avg_us",0),1)) for n,v in d.get("kernels",{}).items()})
PY
