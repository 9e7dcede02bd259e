#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-fit $EXTRA > gpurun_out/r2p_$tag.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2p_$tag.log").read().strip().splitlines()[-1])
    k=d.get("kernels",{})
    print("$tag value", round(d["value"],3), {n:(round(v.get("avg_us",0),1)) for n,v in k.items()})
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/r2p_$tag.log").read()[-600:])
PY
}
run fit16 DVT_FIT_CUS=16 DVT_STREAM_PRIO=none
run fit8 DVT_FIT_CUS=8 DVT_STREAM_PRIO=none
run fit24 DVT_FIT_CUS=24 DVT_STREAM_PRIO=none
EXTRA="--vit-cus-per-32 24" run fit8_vit24 DVT_FIT_CUS=8 DVT_STREAM_PRIO=none
EXTRA="--vit-cus-per-32 28" run fit4_vit28 DVT_FIT_CUS=4 DVT_STREAM_PRIO=none
