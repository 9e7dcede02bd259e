export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes"
for t in 1 0 1 0; do
  DVT_FIT_TAPER=$t python bench.py $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('taper $t', round(d['value'],4), round(d['config']['t_extract_s_serial'],4))"
done
