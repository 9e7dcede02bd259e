#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for p in vit none vit; do
  DVT_STREAM_PRIO=$p timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-fit > gpurun_out/r3q_$p.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3q_$p.log").read().strip().splitlines()[-1])
print("prio=$p value", round(d["value"],3), {n:(round(v.get("avg_us",0),1)) for n,v in d.get("kernels",{}).items()})
PY
done
