export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes --no-npy"
for cfg in "" "--tune 6=0" "--tune 13=0" "--tune 14=1" "" "--tune 6=0" "--fit-batch 1 --tune 6=0" "--fit-batch 2"; do
  python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg [$cfg]: value', round(d['value'],4), ' ms/image', round(d['ms_per_step'],1), ' serial fit s', round(d['config']['t_fit_s_serial'],4))"
done
