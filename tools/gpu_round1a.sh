#!/bin/bash
# first GPU contact: parity tests, smoke, fit microbench + kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 300 python tools/bench_fit.py --iters 1000 --reps 3 > gpurun_out/bench_fit.log 2>&1
tail -5 gpurun_out/bench_fit.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fit -o fit -- python $GRAFT_REPO_ROOT/tools/bench_fit.py --iters 1000 --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_fit.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_fit | head; 
f=$(find gpurun_out/prof_fit -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -30 "$f"
