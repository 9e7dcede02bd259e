#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stage2.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for st in 3 2; do timeout 300 python tools/bench_stage2.py --stages $st 2>&1 | tail -1 | cut -c1-330; done
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s2 -o s2 -- python $R/tools/bench_stage2.py --steps 5 --warmup 1 > $R/gpurun_out/prof_s2.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_s2 -name '*.db' | head -1) > gpurun_out/r2k_s2_kernel_stats.txt; head -22 gpurun_out/r2k_s2_kernel_stats.txt | cut -c1-170
rm -rf gpurun_out/prof_s2
