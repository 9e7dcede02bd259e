#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/bench_stage2.py > gpurun_out/r2j_s2.log 2>&1; tail -2 gpurun_out/r2j_s2.log
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s2 -o s2 -- python $R/tools/bench_stage2.py --steps 5 --warmup 1 > $R/gpurun_out/prof_s2.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_s2 -name '*.db' | head -1) > gpurun_out/r2j_s2_kernel_stats.txt; head -30 gpurun_out/r2j_s2_kernel_stats.txt | cut -c1-170
rm -rf gpurun_out/prof_s2
