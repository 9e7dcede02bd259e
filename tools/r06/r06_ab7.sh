export TMPDIR=/tmp
DVT_TAG=r06n bash tools/gpu.sh "test:concurrent_fits or fit_many or stage1_driver_end_to_end" 2>&1 | tail -4
python tools/bench_fit_batch.py 768 300 2>&1 | grep -v amdgpu | tail -12
B="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes"
for cfg in "--tune 16=0" "" "--tune 16=0" "" "--tune 16=0" ""; do
  python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg [$cfg]: value', round(d['value'],4), ' ms/image', round(d['ms_per_step'],1))"
done
