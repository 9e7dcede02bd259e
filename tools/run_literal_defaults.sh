#!/bin/bash
# the reference's literal defaults through the stage-1 CLI (main_img_denoising.py:173,180-198: --dtype float32 --num_iters 25000
# --warmup_iters 2500): 4 synthetic images; outputs under gpurun_out/$DVT_TAG/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out/${DVT_TAG:-literal}; mkdir -p $O; cd $R
export PYTHONPATH=$R/denoising-vit_amd:$R
rm -rf /dev/shm/dvt_out; printf 'a.jpg\nb.jpg\nc.jpg\nd.jpg\n' > /dev/shm/dvt_list.txt
timeout 900 python -m dvt_amd.stage1 --synthetic --img_path /dev/shm/dvt_list.txt --data_root /dev/shm/dvt_in \
  --save_root /dev/shm/dvt_out --output_dir $O/fp32_literal_defaults > $O/fp32_literal_defaults.log 2>&1
echo rc=$?; tail -4 $O/fp32_literal_defaults.log | cut -c1-300
cp $O/fp32_literal_defaults/*.json $O/ 2>/dev/null; cp $O/fp32_literal_defaults/*.jsonl $O/ 2>/dev/null
