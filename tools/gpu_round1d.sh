#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" gpurun_out/pytest_gpu.log | tail -16
timeout 900 python tools/bench_fit.py --iters 1000 --reps 1 > gpurun_out/bench_fit.log 2>&1
cat gpurun_out/bench_fit.log | cut -c1-300 | tail -16
timeout 900 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -2 gpurun_out/bench.log | cut -c1-2500
