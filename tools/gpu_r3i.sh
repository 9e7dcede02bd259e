#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r3i_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3i_pytest.log
grep -E "^\[|passed|failed|FAILED|ERROR|rc=" gpurun_out/r3i_pytest.log | cut -c1-330 | tail -16
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3i_smoke.log 2>&1; echo "smoke rc=$?"; grep smoke gpurun_out/r3i_smoke.log
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-fit > gpurun_out/r3i_bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3i_bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"],3), d["config"]["t_extract_s_serial"], d["config"]["t_fit_s_serial"], {n:(round(v.get("avg_us",0),1)) for n,v in d.get("kernels",{}).items()})
PY
done
