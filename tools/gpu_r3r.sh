#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vit.py -m gpu -q -p no:cacheprovider -k "forward_vs_oracle or layernorm_folded or wrapper" 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-fit > gpurun_out/r3r_bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3r_bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"],3), d["config"]["t_extract_s_serial"], d["config"]["t_fit_s_serial"], {n:(round(v.get("avg_us",0),1)) for n,v in d.get("kernels",{}).items()})
PY
