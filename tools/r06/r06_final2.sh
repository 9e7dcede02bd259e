#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export DVT_TAG=r06final3
O=$R/gpurun_out/$DVT_TAG; mkdir -p $O; cd $R
echo "=== python bench.py --gpus 2 on a one-GPU box"; python bench.py --gpus 2 --steps 1 --warmup 0 > $O/bench_gpus2.out 2> $O/bench_gpus2.err; echo "rc=$?"; tail -2 $O/bench_gpus2.err
bash tools/gpu.sh suite smoke bench
