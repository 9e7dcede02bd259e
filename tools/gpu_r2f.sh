#!/bin/bash
# round 2, call F: fp32 extractor parity + full-fp32 bench leg
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vit.py -m gpu -q -s -p no:cacheprovider -k "f32" > gpurun_out/r2f_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f_pytest.log
grep -E "fp32 ViT|passed|failed|FAILED|Error|rc=" gpurun_out/r2f_pytest.log | cut -c1-300 | tail -12
timeout 900 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/r2f_bench.log 2>&1; echo "bench rc=$?"
tail -3 gpurun_out/r2f_bench.log | cut -c1-400
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r2f_bench.log').read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ('value','value_fp32_fit','value_fp32')}, d['config'].get('value_fp32_detail'))
except Exception as e: print('parse fail', e)
PY
