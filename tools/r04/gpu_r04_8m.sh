#!/bin/bash
# "8m" (dvt_tune_set(1, 5)): the 8p GEMM ring with the DMA issue inside the MFMA segments.  Bit-identity + variant tests, per-shape
# timing next to 8p, then the pipelined bench A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04v
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vit.py -x -q -m gpu -k "gemm" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python tools/bench_vit_gemm.py 4,5 $((110*1408)) > $O/gemm_110.txt 2>&1; grep -v "^$" $O/gemm_110.txt | tail -12
python tools/bench_vit_gemm.py 4,5 $((398*1408)) > $O/gemm_398.txt 2>&1; grep -v "^$" $O/gemm_398.txt | tail -12
Q="--no-cpu-baseline --no-fp32-fit --no-probes"
for rep in 1 2; do
  python bench.py --steps 10 --warmup 2 $Q > $O/ab_8p_$rep.json 2>> $O/ab.log
  python bench.py --steps 10 --warmup 2 $Q --tune 1=5 > $O/ab_8m_$rep.json 2>> $O/ab.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04v/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value %.3f  ms/step %.1f' % (d['value'], d['ms_per_step']), 'serial extract %.1f ms fit %.1f ms' % (1e3*d['config']['t_extract_s_serial'], 1e3*d['config']['t_fit_s_serial']))
    except Exception as e:
        print(f, 'unreadable', e)
PY
