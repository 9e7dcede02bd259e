#!/bin/bash
# dvt_tune_set(14, 1): small-footprint fit workgroups.  Equality test, then the pipelined bench with / without them and with the
# fit stream at high priority; kernel trace + timeline of the last combination (do fit steps now run inside attention launches,
# and what do they cost there?).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04r
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fit.py -x -q -m gpu -k "small_footprint or fused_row_kernel" -s > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
Q="--no-cpu-baseline --no-fp32-fit --no-probes"
for rep in 1 2; do
  python bench.py --steps 10 --warmup 2 $Q > $O/ab_default_$rep.json 2>> $O/ab.log
  python bench.py --steps 10 --warmup 2 $Q --tune 14=1 > $O/ab_small_$rep.json 2>> $O/ab.log
  DVT_STREAM_PRIO=fit python bench.py --steps 10 --warmup 2 $Q --tune 14=1 > $O/ab_small_priofit_$rep.json 2>> $O/ab.log
done
DVT_STREAM_PRIO=fit rocprofv3 --kernel-trace -d $O/prof -o pipe -- python bench.py --steps 6 --warmup 1 $Q --tune 14=1 > $O/prof_bench.json 2> $O/prof.log
DB=$(find $O/prof -name "*.db" | head -1)
python tools/pipe_timeline.py "$DB" > $O/timeline_small_priofit.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04r/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value %.3f  ms/step %.1f' % (d['value'], d['ms_per_step']), 'serial fit %.1f ms extract %.1f ms' % (1e3*d['config']['t_fit_s_serial'], 1e3*d['config']['t_extract_s_serial']))
    except Exception as e:
        print(f, 'unreadable', e)
PY
tail -16 $O/timeline_small_priofit.txt
