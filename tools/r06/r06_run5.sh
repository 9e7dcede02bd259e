#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
DVT_TAG=r06h bash tools/gpu.sh "test:attention or vit or chain or stage1 or gemm_qkv or end_to_end or cat_demo"
B="--steps 12 --warmup 3 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes"
for pad in 128 32 128 32; do
  DVT_VIT_ROW_PAD=$pad python bench.py $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('row_pad $pad', round(d['value'],4), round(d['config']['t_extract_s_serial'],4), round(d['config']['t_fit_s_serial'],4), d['config']['extract_launch_views'])"
done 2>&1 | tee $O/pipelined_row_pad_ab.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_walks -o walks -- python $R/tools/pmc_target_walks.py > $O/pmc_walks.log 2>&1; echo "pmc rc=$?"
f=$(find $O/pmc_walks -name '*.db' | head -1); [ -n "$f" ] && python $R/tools/pmc_walk_stats.py $f > $O/pmc_walks_per_dispatch.txt 2>&1
rm -rf $O/pmc_walks; tail -10 $O/pmc_walks_per_dispatch.txt | cut -c1-330
