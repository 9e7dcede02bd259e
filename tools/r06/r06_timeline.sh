#!/bin/bash
# round 6: kernel-trace timeline of the pipelined bench at the driver's flags with the final binary (fit-batch 4, tapered tail)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r06k; mkdir -p $O
Q="--no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes"
cd /tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o pipe -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 $Q > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.log
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
python tools/pipe_timeline.py "$DB" > $O/timeline.txt 2>&1
rm -rf $O/prof
head -45 $O/timeline.txt | cut -c1-250
tail -1 $O/prof_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value under the profiler', d['value'])"
