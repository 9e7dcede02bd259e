#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
run() { echo "== $1"; timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-probes $1 > gpurun_out/b.log 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/b.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("   images/s", round(d["value"],3), "ms/img", round(d["ms_per_step"],1))
else:
    print(open('gpurun_out/b.log').read()[-800:])
PY
}
run ""
run "--vit-cus-per-32 30"
run "--vit-cus-per-32 28"
run "--vit-cus-per-32 24"
run "--vit-cus-per-32 16"
