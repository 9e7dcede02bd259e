#!/bin/bash
# round-2 final evidence: full GPU suite, smoke, the default bench line, rocprofv3 kernel stats (serial / pipelined), PMC
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/final/gpu_suite.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/final/gpu_suite.txt
grep -E "passed|failed|FAILED|ERROR|rc=" gpurun_out/final/gpu_suite.txt | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; echo "smoke rc=$?"
timeout 1200 python bench.py > gpurun_out/final/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/final/bench_default.log > gpurun_out/final/bench_line.json; cut -c1-700 gpurun_out/final/bench_line.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_serial -o serial -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probes --no-fp32-fit --pipeline-depth 1 > $R/gpurun_out/final/prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_pipe -o pipe -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-probes --no-fp32-fit > $R/gpurun_out/final/prof_pipe.log 2>&1
cd $R
for n in serial pipe; do python tools/rocpd_stats.py $(find gpurun_out/final/prof_$n -name '*.db' | head -1) > gpurun_out/final/${n}_kernel_stats.txt; tail -1 gpurun_out/final/prof_$n.log | cut -c1-120; done
rm -rf gpurun_out/final/prof_serial gpurun_out/final/prof_pipe
mkdir -p gpurun_out/final/pmc
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/final/pmc/$c -o $c -- python $R/tools/pmc_target.py > $R/gpurun_out/final/pmc/$c.log 2>&1
  f=$(find $R/gpurun_out/final/pmc/$c -name '*.db' | head -1); [ -n "$f" ] && python $R/tools/pmc_stats.py $f > $R/gpurun_out/final/pmc/$c.txt 2>&1
  rm -rf $R/gpurun_out/final/pmc/$c
done
cd $R; head -14 gpurun_out/final/serial_kernel_stats.txt | cut -c1-150
