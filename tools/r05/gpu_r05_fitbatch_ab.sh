#!/bin/bash
# Same-box A/B of the pipelined bench at --fit-batch 1 / 2 / 4 (ViT-B, bf16 mode; side legs off).
#   gpurun --timeout 400 -- 'bash tools/r05/gpu_r05_fitbatch_ab.sh [steps]'
O=gpurun_out/r05_fitbatch; mkdir -p $O
S=${1:-20}
F="--steps $S --warmup 4 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes"
for i in 1 2; do for k in 1 2 4; do
  [ $i == 2 ] && [ $k == 4 ] && continue
  timeout 150 python bench.py $F --fit-batch $k > $O/s${S}_k${k}_$i.log 2>&1
  tail -1 $O/s${S}_k${k}_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fit-batch', d['config'].get('fit_batch'), 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1))"
done; done
