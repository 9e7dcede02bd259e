#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pipe -o pipe -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-probes > $GRAFT_REPO_ROOT/gpurun_out/prof_pipe.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_pipe.log | cut -c1-200
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py $(find gpurun_out/prof_pipe -name '*.db' | head -1) > gpurun_out/prof_pipe_stats.txt; head -22 gpurun_out/prof_pipe_stats.txt | cut -c1-150
python - <<'PY'
import sqlite3, glob
db=sqlite3.connect(glob.glob('gpurun_out/prof_pipe/**/*.db', recursive=True)[0])
rows=list(db.execute("select name,start,end,stream_id from kernels order by start"))
t0=rows[0][1]; t1=max(r[2] for r in rows)
# busy time per stream and overlap
import collections
bys=collections.defaultdict(list)
for n,s,e,st in rows: bys[st].append((s,e))
for st,iv in bys.items():
    print("stream",st,"kernels",len(iv),"busy ms",sum(e-s for s,e in iv)/1e6, "span ms",(max(e for s,e in iv)-min(s for s,e in iv))/1e6)
PY
python tools/gap_attrib.py gpurun_out/prof_pipe; rm -rf gpurun_out/prof_pipe
