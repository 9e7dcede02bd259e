#!/bin/bash
# round 6: full GPU suite, the reference's literal defaults, PMC of the GEMM walks
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
export PYTHONPATH=$R/denoising-vit_amd:$R
DVT_TAG=r06f bash tools/gpu.sh suite
printf 'a.jpg\nb.jpg\nc.jpg\nd.jpg\n' > /dev/shm/dvt_list.txt
echo "=== literal defaults: --dtype float32 --num_iters 25000 --warmup_iters 2500 (4 synthetic images, fit_batch auto = 4)"
timeout 600 python -m dvt_amd.stage1 --synthetic --img_path /dev/shm/dvt_list.txt --data_root /dev/shm/dvt_in \
  --save_root /dev/shm/dvt_out --output_dir $O/fp32_literal_defaults > $O/fp32_literal_defaults.log 2>&1
echo rc=$?; tail -3 $O/fp32_literal_defaults.log | cut -c1-300
cp $O/fp32_literal_defaults/*.json $O/ 2>/dev/null; cp $O/fp32_literal_defaults/*.jsonl $O/ 2>/dev/null
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_walks -o walks -- python $R/tools/pmc_target_walks.py > $O/pmc_walks.log 2>&1; echo "pmc rc=$?"
f=$(find $O/pmc_walks -name '*.db' | head -1); [ -n "$f" ] && python $R/tools/pmc_stats.py $f > $O/pmc_walks.txt 2>&1 && python $R/tools/rocpd_stats.py $f > $O/pmc_walks_durations.txt 2>&1
rm -rf $O/pmc_walks
grep -A5 "gemm_bf16" $O/pmc_walks.txt | cut -c1-150 | head -60
