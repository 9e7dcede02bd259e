#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for b in 1 0 1 0; do
  DVT_VIT_BALANCE=$b timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-fp32-fit --no-probes > gpurun_out/r3t.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3t.log").read().strip().splitlines()[-1])
print("balance=$b value", round(d["value"],3), "t_extract", round(d["config"]["t_extract_s_serial"],4), "t_fit", round(d["config"]["t_fit_s_serial"],4))
PY
done
