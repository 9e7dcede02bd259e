#!/bin/bash
# One parameterised GPU-box script (replaces the per-experiment gpu_r*.sh files of rounds 1-2).
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu.sh <stage> [<stage> ...]'
# Stages (outputs under gpurun_out/<tag>/, tag = $DVT_TAG or "run"):
#   suite        full `pytest -m gpu`                         smoke     __graft_entry__.smoke()
#   bench        the default bench line                       bench:<args>  bench.py with extra args (":" -> " ")
#   prof         rocprofv3 --kernel-trace --stats of a serial and a pipelined bench run (kernel tables kept)
#   pmc          FETCH_SIZE / WRITE_SIZE / SQ passes (separate rocprofv3 --pmc runs) over tools/pmc_target.py
#   py:<file>[:args]   python <file> args...                  test:<pytest node or -k expr>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${DVT_TAG:-run}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
n_bench=0; n_py=0
for stage in "$@"; do
  name=${stage%%:*}; arg=${stage#*:}; [ "$arg" == "$stage" ] && arg=""
  echo "=== stage $stage"
  case $name in
    suite)
      timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_suite.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_suite.txt
      grep -E "passed|failed|FAILED|ERROR|rc=" $O/gpu_suite.txt | tail -6 ;;
    test)
      f=$O/test_$(echo "$arg" | tr -c 'A-Za-z0-9' '_' | cut -c1-60).txt
      timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "$arg" > $f 2>&1; echo "pytest rc=$?" >> $f
      grep -E "^\[|passed|failed|FAILED|ERROR|rc=|Error" $f | cut -c1-400 | tail -40 ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -4 $O/smoke.txt ;;
    bench)
      n_bench=$((n_bench+1)); f=$O/bench${n_bench}_$(echo "$arg" | tr -c 'A-Za-z0-9=' '_' | rev | cut -c1-50 | rev).log
      timeout 1500 python bench.py $(echo "$arg" | tr ':' ' ') > $f 2>&1; echo "bench rc=$?"
      tail -1 $f > ${f%.log}.json; python - "${f%.log}.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print({k: d.get(k) for k in ("value", "ms_per_step", "value_fp32_fit", "value_fp32", "roofline")})
    print({k: (round(v["avg_us"], 1), round(v["achieved"], 1)) for k, v in d.get("kernels", {}).items()}, "isolated",
          {k: (round(v["avg_us"], 1), round(v["achieved"], 1)) for k, v in d.get("kernels_isolated", {}).items()})
    print({k: d["config"].get(k) for k in ("t_extract_s_serial", "t_fit_s_serial")})
except Exception as e:
    print("no JSON line:", e)
PY
      ;;
    prof)
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o serial -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probes --no-fp32-fit --no-vit-large --no-stage2 --pipeline-depth 1 > $O/prof_serial.log 2>&1
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_pipe -o pipe -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-probes --no-fp32-fit --no-vit-large --no-stage2 > $O/prof_pipe.log 2>&1
      cd $R
      for n in serial pipe; do python tools/rocpd_stats.py $(find $O/prof_$n -name '*.db' | head -1) > $O/${n}_kernel_stats.txt; tail -1 $O/prof_$n.log | cut -c1-120; done
      rm -rf $O/prof_serial $O/prof_pipe
      head -16 $O/serial_kernel_stats.txt | cut -c1-160 ;;
    pmc)
      mkdir -p $O/pmc; cd /tmp
      run() { n=$1; shift
        timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc/$n -o $n -- python $R/tools/pmc_target.py $DVT_PMC_ARGS > $O/pmc/$n.log 2>&1; echo "$n rc=$?"
        f=$(find $O/pmc/$n -name '*.db' | head -1); [ -n "$f" ] && python $R/tools/pmc_stats.py $f > $O/pmc/$n.txt 2>&1; rm -rf $O/pmc/$n; }
      run FETCH_SIZE FETCH_SIZE
      run WRITE_SIZE WRITE_SIZE
      run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
      cd $R; head -30 $O/pmc/FETCH_SIZE.txt | cut -c1-150 ;;
    py)
      file=${arg%%:*}; rest=${arg#*:}; [ "$rest" == "$arg" ] && rest=""
      n_py=$((n_py+1)); f=$O/$(basename $file .py)${n_py}_$(echo "$rest" | tr -c 'A-Za-z0-9=' '_' | cut -c1-40).log
      timeout 1500 python $file $(echo "$rest" | tr ':' ' ') > $f 2>&1; echo "py rc=$?"; tail -${DVT_TAIL:-60} $f | cut -c1-260 ;;
    *) echo "unknown stage $stage" ;;
  esac
  cd $R
done
