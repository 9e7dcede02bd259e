#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > gpurun_out/final/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/final/bench_default.log > gpurun_out/final/bench_line.json; cut -c1-200 gpurun_out/final/bench_line.json
