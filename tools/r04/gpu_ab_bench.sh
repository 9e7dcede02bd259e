#!/bin/bash
# same-box A/B of bench.py under developer knobs: bash tools/r04/gpu_ab_bench.sh "<args A>" "<args B>" ... (each run: 8 images)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${DVT_TAG:-ab}; mkdir -p $O
cd $R
i=0
for rnd in 1 2; do
for args in "$@"; do
  i=$((i+1))
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-fit $args > $O/ab_$i.log 2>&1
  python - "$O/ab_$i.log" "$args" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get("kernels", {})
    print(f"[{sys.argv[2] or 'default':40s}] value {d['value']:.3f} images/s  ms/step {d['ms_per_step']:.1f}  in-region gemm {k.get('vit_gemm', {}).get('achieved', 0):.0f} TF/s attn {k.get('vit_attn', {}).get('achieved', 0):.0f} TF/s  serial extract {d['config']['t_extract_s_serial']*1e3:.0f} ms fit {d['config']['t_fit_s_serial']*1e3:.0f} ms", flush=True)
except Exception as e:
    print("failed", sys.argv[2], e)
PY
done
done
