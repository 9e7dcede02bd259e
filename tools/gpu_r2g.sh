#!/bin/bash
# round 2, call G: evidence -- rocprofv3 kernel stats of the serial and the pipelined bench, PMC passes
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_serial -o serial -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-fit --no-probes --pipeline-depth 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pipe -o pipe -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-fit --no-probes > $GRAFT_REPO_ROOT/gpurun_out/prof_pipe.log 2>&1
cd $GRAFT_REPO_ROOT
for n in serial pipe; do python tools/rocpd_stats.py $(find gpurun_out/prof_$n -name '*.db' | head -1) > gpurun_out/prof_${n}_stats.txt; tail -1 gpurun_out/prof_$n.log | cut -c1-200; done
python tools/gap_attrib.py gpurun_out/prof_pipe > gpurun_out/prof_pipe_gaps.txt 2>&1
rm -rf gpurun_out/prof_serial gpurun_out/prof_pipe
cd /tmp
run() { # name, counters...
  name=$1; shift
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$name -o $name -- python $GRAFT_REPO_ROOT/tools/pmc_target.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.log 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
cd $GRAFT_REPO_ROOT
for n in sq1 fetch write tcc; do f=$(find gpurun_out/pmc/$n -name '*.db' | head -1); [ -n "$f" ] && python tools/pmc_stats.py $f > gpurun_out/pmc/$n.txt 2>&1; rm -rf gpurun_out/pmc/$n; done
head -30 gpurun_out/prof_pipe_stats.txt | cut -c1-150
