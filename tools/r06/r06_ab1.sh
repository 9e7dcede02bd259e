export TMPDIR=/tmp; O=gpurun_out/r06c; mkdir -p $O
B="--steps 12 --warmup 3 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes"
for i in 1 2; do
  python tools/bench_lab.py $B --tune 1=13 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('walk13', d['value'], d['config']['t_extract_s_serial'], d['config']['t_fit_s_serial'])"
  python tools/bench_lab.py $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('walk8b', d['value'], d['config']['t_extract_s_serial'], d['config']['t_fit_s_serial'])"
done
B2="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes"
for k in 1 4 2 1 4; do
  python bench.py $B2 --fit-batch $k 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fitbatch $k', d['value'])"
done
