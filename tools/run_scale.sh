#!/bin/bash
# The 1 / 2 / 4 / 8-GPU lines of bench.py on ONE node (the driver's SCALE run does the same; this is for a builder with
# a multi-GPU node).  Prints one JSON line per N into profiles/<tag>_scale_N.json; scaling efficiency is NOT computed
# here -- the reader divides.
#   bash tools/run_scale.sh [tag] [steps] [warmup]
TAG=${1:-scale}; STEPS=${2:-6}; WARM=${3:-1}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/$TAG
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && { echo "only $NG GPU(s): skipping N=$N"; continue; }
  if [ "$N" == 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM > gpurun_out/$TAG/bench_$N.log 2>&1
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps $STEPS --warmup $WARM --no-cpu-baseline > gpurun_out/$TAG/bench_$N.log 2>&1
  fi
  tail -1 gpurun_out/$TAG/bench_$N.log > gpurun_out/$TAG/scale_$N.json
  python -c "import json,sys; d=json.loads(open(sys.argv[1]).read()); print('N =', d['n_gpus'], 'value =', round(d['value'], 3), d['unit'], 'ms/step =', round(d['ms_per_step'], 1))" gpurun_out/$TAG/scale_$N.json
done
