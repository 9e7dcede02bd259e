#!/bin/bash
# Round 4: (1) the bench line at the flags the round-end driver uses, (2) a kernel trace of the pipelined region and
# its timeline (tools/pipe_timeline.py), (3) A/B of pipeline depth and of the in-region probes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04p
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.log
Q="--no-cpu-baseline --no-fp32-fit"
rocprofv3 --kernel-trace -d $O/prof -o pipe -- python bench.py --steps 6 --warmup 1 $Q --no-probes > $O/prof_bench.json 2> $O/prof.log
DB=$(find $O/prof -name "*.db" | head -1)
python tools/pipe_timeline.py "$DB" > $O/timeline.txt 2>&1
for rep in 1 2; do
  python bench.py --steps 10 --warmup 2 $Q --no-probes > $O/ab_noprobes_$rep.json 2>> $O/ab.log
  python bench.py --steps 10 --warmup 2 $Q > $O/ab_probes_$rep.json 2>> $O/ab.log
  python bench.py --steps 10 --warmup 2 $Q --no-probes --pipeline-depth 3 > $O/ab_depth3_$rep.json 2>> $O/ab.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04p/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value %.3f  ms/step %.1f' % (d['value'], d['ms_per_step']))
    except Exception as e:
        print(f, 'unreadable', e)
PY
tail -30 $O/timeline.txt
