#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|cos mean|rc=" gpurun_out/pytest_gpu.log | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 300 python tools/bench_grid.py > gpurun_out/bench_grid.log 2>&1; cat gpurun_out/bench_grid.log | tail -12
timeout 300 python tools/bench_gemm_f32.py > gpurun_out/bench_gemm_f32.log 2>&1; cat gpurun_out/bench_gemm_f32.log | tail -30
timeout 300 python tools/bench_fit.py --iters 1000 --reps 2 > gpurun_out/bench_fit.log 2>&1
head -2 gpurun_out/bench_fit.log | cut -c1-200
timeout 900 python bench.py --steps 6 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -2 gpurun_out/bench.log
timeout 900 python bench.py --steps 4 --warmup 1 --pipeline-depth 1 --no-cpu-baseline > gpurun_out/bench_depth1.log 2>&1
tail -1 gpurun_out/bench_depth1.log | cut -c1-400
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_bench | head -3
