#!/bin/bash
# fit stream at high HIP priority: does the fit then run inside attention launches, and what does a step cost there?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04q
mkdir -p $O
Q="--no-cpu-baseline --no-fp32-fit --no-probes"
DVT_STREAM_PRIO=fit rocprofv3 --kernel-trace -d $O/prof -o pipe -- python bench.py --steps 6 --warmup 1 $Q > $O/prof_bench_priofit.json 2> $O/prof.log
DB=$(find $O/prof -name "*.db" | head -1)
python tools/pipe_timeline.py "$DB" > $O/timeline_priofit.txt 2>&1
for rep in 1 2; do
  DVT_STREAM_PRIO=fit python bench.py --steps 10 --warmup 2 $Q > $O/ab_priofit_$rep.json 2>> $O/ab.log
  python bench.py --steps 10 --warmup 2 $Q > $O/ab_none_$rep.json 2>> $O/ab.log
  DVT_STREAM_PRIO=vit python bench.py --steps 10 --warmup 2 $Q > $O/ab_priovit_$rep.json 2>> $O/ab.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04q/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value %.3f  ms/step %.1f' % (d['value'], d['ms_per_step']))
    except Exception as e:
        print(f, 'unreadable', e)
PY
tail -22 $O/timeline_priofit.txt
