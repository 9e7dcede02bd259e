#!/bin/bash
# round 6 final evidence: full GPU suite, smoke, default bench line, rocprofv3 kernel tables (serial + pipelined), PMC traffic passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export DVT_TAG=r06final
O=$R/gpurun_out/$DVT_TAG; mkdir -p $O; cd $R
bash tools/gpu.sh suite smoke bench prof
DVT_PMC_ARGS=385 bash tools/gpu.sh pmc
python tools/pmc_traffic.py $O/pmc $O/pmc_traffic.json 385 | tail -8
