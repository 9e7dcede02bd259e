#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_vit
timeout 400 python tools/bench_vit_order.py > gpurun_out/r2h_order.log 2>&1; cat gpurun_out/r2h_order.log | tail -20
R=$GRAFT_REPO_ROOT
cd /tmp
run() { # name mblock nt counters...
  name=$1; mb=$2; nt=$3; shift 3
  DVT_MBLOCK=$mb DVT_NT=$nt timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_vit/$name -o $name -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/pmc_vit/$name.log 2>&1
  echo "$name rc=$?"
  f=$(find $R/gpurun_out/pmc_vit/$name -name '*.db' | head -1)
  [ -n "$f" ] && python $R/tools/pmc_stats.py $f > $R/gpurun_out/pmc_vit/$name.txt 2>&1
  rm -rf $R/gpurun_out/pmc_vit/$name
  grep -A1 "gemm_bf16_kernel_8p" $R/gpurun_out/pmc_vit/$name.txt | cut -c1-120
}
run fetch_mb1_nt0 1 0 FETCH_SIZE
run fetch_mb4_nt0 4 0 FETCH_SIZE
run fetch_mb4_nt1 4 1 FETCH_SIZE
run fetch_mb8_nt1 8 1 FETCH_SIZE
