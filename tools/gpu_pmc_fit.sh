#!/bin/bash
# PMC passes over the fit step (separate rocprofv3 --pmc runs; kernel-trace only, no other trace domains)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fit
mkdir -p $OUT
cd /tmp
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- python $GRAFT_REPO_ROOT/tools/pmc_target_fit.py > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $GRAFT_REPO_ROOT
for n in sq tcp tcc ta fetch write; do f=$(find $OUT/$n -name '*.db' | head -1); [ -n "$f" ] && python tools/pmc_stats.py $f > $OUT/$n.txt 2>&1; rm -rf $OUT/$n; done
grep -A12 "fit_rows_kernel<768, false>" $OUT/sq.txt $OUT/tcp.txt $OUT/tcc.txt $OUT/ta.txt $OUT/fetch.txt $OUT/write.txt | cut -c1-150
