#!/bin/bash
# PMC refresh for the round-2 final kernel set: fetch / write / sq1 (separate passes), 128 views + 60 fit steps
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc3
R=$GRAFT_REPO_ROOT
cd /tmp
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc3/$name -o $name -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc3/$name.log 2>&1
  echo "$name rc=$?"
  f=$(find $R/gpurun_out/pmc3/$name -name '*.db' | head -1)
  [ -n "$f" ] && python $R/tools/pmc_stats.py $f > $R/gpurun_out/pmc3/$name.txt 2>&1
  rm -rf $R/gpurun_out/pmc3/$name
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
cd $R
grep -A1 -E "gemm_bf16_kernel_8p|attention_kernel|adam_|fit_rows|fit_backward|layernorm_kernel<false>|grid_sort|shadow" gpurun_out/pmc3/fetch.txt | cut -c1-130 | grep -v "^--"
