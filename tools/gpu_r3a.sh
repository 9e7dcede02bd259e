#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fit.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 300 python tools/bench_fit_knobs.py 2>&1 | head -3 | tail -2
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-fit > gpurun_out/r3a_bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3a_bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"],3), d["config"]["t_extract_s_serial"], d["config"]["t_fit_s_serial"], {n:(round(v.get("avg_us",0),1)) for n,v in d.get("kernels",{}).items()})
PY
done
