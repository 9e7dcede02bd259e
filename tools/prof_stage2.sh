export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r07c; mkdir -p $O; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_s2 -o s2 -- python $R/tools/bench_stage2.py > $O/s2.log 2>&1
cd $R; python tools/rocpd_stats.py $(find $O/prof_s2 -name '*.db' | head -1) > $O/s2_kernel_stats.txt; rm -rf $O/prof_s2; head -30 $O/s2_kernel_stats.txt | cut -c1-170
