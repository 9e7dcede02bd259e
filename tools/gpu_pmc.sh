#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters_list.txt 2>&1
run() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$name -o $name -- python $GRAFT_REPO_ROOT/tools/pmc_target.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.log 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run sq2 SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $GRAFT_REPO_ROOT
for n in sq1 sq2 fetch write tcc grbm; do f=$(find gpurun_out/pmc/$n -name '*.db' | head -1); echo "== $n $f"; [ -n "$f" ] && python tools/pmc_stats.py $f > gpurun_out/pmc/$n.txt 2>&1; head -3 gpurun_out/pmc/$n.txt; done
for n in sq1 sq2 fetch write tcc grbm; do rm -rf gpurun_out/pmc/$n; done   # keep the summaries only (merge limit 64 MiB)
du -sh gpurun_out/pmc
