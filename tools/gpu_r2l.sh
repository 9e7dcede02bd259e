#!/bin/bash
# full GPU suite + smoke + bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r2l_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l_pytest.log
grep -E "^\[|passed|failed|FAILED|ERROR|rc=" gpurun_out/r2l_pytest.log | tail -30
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2l_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r2l_smoke.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/r2l_bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r2l_bench.log | cut -c1-1200
