#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_vit.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" gpurun_out/pytest_gpu.log | tail -6
timeout 600 python tools/bench_vit.py > gpurun_out/bench_vit.log 2>&1; grep variant gpurun_out/bench_vit.log
timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-220
