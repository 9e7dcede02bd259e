#!/bin/bash
# round-1d evidence: kernel traces of the serial and the pipelined bench + gap attribution
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_serial -o serial -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes --pipeline-depth 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pipe -o pipe -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-probes > $GRAFT_REPO_ROOT/gpurun_out/prof_pipe.log 2>&1
cd $GRAFT_REPO_ROOT
for n in serial pipe; do python tools/rocpd_stats.py $(find gpurun_out/prof_$n -name '*.db' | head -1) > gpurun_out/prof_${n}_stats.txt; tail -1 gpurun_out/prof_$n.log | cut -c1-200; done
python tools/gap_attrib.py gpurun_out/prof_pipe > gpurun_out/prof_pipe_gaps.txt 2>&1
rm -rf gpurun_out/prof_serial gpurun_out/prof_pipe
