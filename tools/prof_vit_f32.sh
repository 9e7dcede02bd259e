# rocprofv3 kernel table of the exact-fp32 extractor on one launch plan (tools/bench_vit_f32_ab.py, 160 views)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${DVT_TAG:-r07g}; mkdir -p $O; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_f32 -o f32 -- python $R/tools/bench_vit_f32_ab.py 160 > $O/f32.log 2>&1
cd $R; python tools/rocpd_stats.py $(find $O/prof_f32 -name '*.db' | head -1) > $O/f32_kernel_stats.txt; rm -rf $O/prof_f32; head -16 $O/f32_kernel_stats.txt | cut -c1-170; tail -5 $O/f32.log
