#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for t in "1=-61" "1=-60" "1=-61" "1=-60"; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-fit --tune=$t > gpurun_out/r3j_bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3j_bench.log").read().strip().splitlines()[-1])
print("tune $t value", round(d["value"],3), d["config"]["t_extract_s_serial"], d["config"]["t_fit_s_serial"], {n:(round(v.get("avg_us",0),1)) for n,v in d.get("kernels",{}).items()})
PY
done
