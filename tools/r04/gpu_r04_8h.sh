#!/bin/bash
# "8h" (dvt_tune_set(1, 10)): the 8p GEMM ring walked in two phases per k-tile.  GEMM tests, per-shape A/B, pipelined bench A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04z
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vit.py -x -q -m gpu -k "gemm" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python tools/lab_gemm8p_variants.py $((110*1408)) 5,10 > $O/variants_110.txt 2>&1; grep -v "amdgpu.ids" $O/variants_110.txt
python tools/lab_gemm8p_variants.py $((398*1408)) 10 > $O/variants_398.txt 2>&1; grep -v "amdgpu.ids" $O/variants_398.txt
Q="--no-cpu-baseline --no-fp32-fit --no-probes"
for rep in 1 2 3; do
  python bench.py --steps 10 --warmup 2 $Q > $O/ab_8p_$rep.json 2>> $O/ab.log
  python bench.py --steps 10 --warmup 2 $Q --tune 1=10 > $O/ab_8h_$rep.json 2>> $O/ab.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04z/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value %.3f  ms/step %.1f' % (d['value'], d['ms_per_step']), 'serial extract %.1f ms fit %.1f ms' % (1e3*d['config']['t_extract_s_serial'], 1e3*d['config']['t_fit_s_serial']))
    except Exception as e:
        print(f, 'unreadable', e)
PY
