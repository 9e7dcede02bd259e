export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes --no-npy"
for m in 0 1 2 4 7 0 3 5 6; do
  python tools/bench_lab.py $B --tune 15=$m 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('skip mask $m (1 rows, 2 backward, 4 adam): value', round(d['value'],4), ' ms/image', round(d['ms_per_step'],1), ' serial fit s', round(d['config']['t_fit_s_serial'],4))"
done
