#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06g; mkdir -p $O; cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_walks -o walks -- python $R/tools/pmc_target_walks.py > $O/pmc_walks.log 2>&1; echo "pmc rc=$?"
f=$(find $O/pmc_walks -name '*.db' | head -1); [ -n "$f" ] && python $R/tools/pmc_walk_stats.py $f > $O/pmc_walks_per_dispatch.txt 2>&1
rm -rf $O/pmc_walks; tail -12 $O/pmc_walks_per_dispatch.txt | cut -c1-330
cd $R
python tools/lab_gemm8t.py 560384 0,2,3,4,8 2>&1 | grep -v amdgpu.ids | tee $O/lab_gemm8t_runs.txt
B="--steps 12 --warmup 3 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes"
for cfg in "" "--tune 1=11,1=-202" "--tune 1=11,1=-203" "" "--tune 1=11,1=-204" "--tune 1=11,1=-202"; do
  python tools/bench_lab.py $B $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg [$cfg]', round(d['value'],4), round(d['config']['t_extract_s_serial'],4), round(d['config']['t_fit_s_serial'],4))"
done 2>&1 | tee $O/pipelined_8t_runs.txt
