#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-fit --model vit_large_patch14_dinov2.lvd142m > gpurun_out/r3f_vitl.log 2>&1; echo rc=$?
python - <<PY
import json
d=json.loads(open("gpurun_out/r3f_vitl.log").read().strip().splitlines()[-1])
print("ViT-L value", round(d["value"],3), d["config"]["t_extract_s_serial"], d["config"]["t_fit_s_serial"], {n:(round(v.get("avg_us",0),1), round(v.get("rate",0),1)) for n,v in d.get("kernels_isolated",{}).items()})
PY
