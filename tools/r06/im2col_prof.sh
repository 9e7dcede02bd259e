export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r07s; mkdir -p $O; cd $R
DVT_TAG=r07s bash tools/gpu.sh "test:vit_forward or strides or wrapper_api or stage1_driver or chain" 
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o serial -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probes --no-fp32-fit --no-vit-large --no-stage2 --pipeline-depth 1 > $O/prof_serial.log 2>&1
cd $R; python tools/rocpd_stats.py $(find $O/prof_serial -name '*.db' | head -1) > $O/serial_kernel_stats.txt; rm -rf $O/prof_serial; grep "im2col\|8p<4\|layernorm\|ln_" $O/serial_kernel_stats.txt | cut -c1-150
