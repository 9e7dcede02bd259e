#!/bin/bash
# Round-4 evidence lines: the reference's LITERAL defaults through the stage-1 driver, and BASELINE
# configs[2] (ViT-L/14, 4 concurrent fits) through bench.py at 1000 and 20000 iterations, + the K sweep of the fit at C = 1024.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${DVT_TAG:-r04j}; mkdir -p $O
cd $R
export PYTHONPATH=$R/denoising-vit_amd:$R
printf 'a.jpg\nb.jpg\nc.jpg\nd.jpg\n' > /dev/shm/dvt_list.txt
echo "=== literal defaults: --dtype float32 --num_iters 25000 --warmup_iters 2500 (4 synthetic images, fit_batch auto = 4)"
timeout 600 python -m dvt_amd.stage1 --synthetic --img_path /dev/shm/dvt_list.txt --data_root /dev/shm/dvt_in \
  --save_root /dev/shm/dvt_out --output_dir $O/fp32_literal_defaults > $O/fp32_literal_defaults.log 2>&1
echo rc=$?; tail -3 $O/fp32_literal_defaults.log | cut -c1-300
echo "=== configs[2]: ViT-L/14, 4 concurrent fits, 1000 iterations"
timeout 600 python bench.py --model vit_large_patch14_dinov2.lvd142m --fit-batch 4 --steps 8 --warmup 4 --no-cpu-baseline --no-fp32-fit \
  > $O/bench_vit_large_fb4_1000.log 2>&1; echo rc=$?; tail -1 $O/bench_vit_large_fb4_1000.log > $O/bench_vit_large_fb4_1000.json
python - $O/bench_vit_large_fb4_1000.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: d["config"].get(k) for k in ("t_extract_s_serial", "t_fit_s_serial", "fit_batch", "extract_launch_views")})
PY
echo "=== configs[2]: ViT-L/14, 4 concurrent fits, 20000 iterations (the paper's schedule)"
timeout 900 python bench.py --model vit_large_patch14_dinov2.lvd142m --fit-batch 4 --num-iters 20000 --warmup-iters 2000 --steps 8 --warmup 4 \
  --no-cpu-baseline --no-fp32-fit > $O/bench_vit_large_fb4_20000.log 2>&1; echo rc=$?; tail -1 $O/bench_vit_large_fb4_20000.log > $O/bench_vit_large_fb4_20000.json
python - $O/bench_vit_large_fb4_20000.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: d["config"].get(k) for k in ("t_extract_s_serial", "t_fit_s_serial", "fit_batch")})
PY
echo "=== fit throughput vs concurrent fits, C = 1024"
timeout 300 python tools/bench_fit_batch.py 1024 300 > $O/fit_throughput_vs_K_c1024.txt 2>&1; echo rc=$?; tail -12 $O/fit_throughput_vs_K_c1024.txt | cut -c1-200
