#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_pipe -o pipe -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-probes $@ > $GRAFT_REPO_ROOT/gpurun_out/prof_pipe.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/stream_busy.py gpurun_out/prof_pipe > gpurun_out/prof_pipe_busy.txt 2>&1
rm -rf gpurun_out/prof_pipe
cat gpurun_out/prof_pipe_busy.txt
