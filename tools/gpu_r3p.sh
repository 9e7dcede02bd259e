#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fit.py -m gpu -q -p no:cacheprovider -k "lazy or long_run or fused_row" 2>&1 | tail -2
timeout 200 python tools/bench_fit_knobs.py 2>&1 | tail -4
for t in "12=1" "12=0" "11=0" "12=1" "12=0"; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-fit --tune=$t > gpurun_out/r3p_bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3p_bench.log").read().strip().splitlines()[-1])
print("tune $t value", round(d["value"],3), d["config"]["t_extract_s_serial"], d["config"]["t_fit_s_serial"], {n:(round(v.get("avg_us",0),1)) for n,v in d.get("kernels",{}).items()})
PY
done
