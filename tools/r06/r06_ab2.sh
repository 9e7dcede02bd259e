export TMPDIR=/tmp; O=gpurun_out/r06e; mkdir -p $O
python tools/lab_gemm8t.py 560384 0,2,4,8,16 2>&1 | grep -v amdgpu.ids
B="--steps 12 --warmup 3 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2 --no-probes"
for cfg in "" "--tune 1=11" "--tune 1=11,1=-204" "--tune 1=11,1=-208" "" "--tune 1=11,1=-216" "--tune 1=11,1=-202"; do
  python tools/bench_lab.py $B $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg [$cfg]', round(d['value'],4), round(d['config']['t_extract_s_serial'],4), round(d['config']['t_fit_s_serial'],4))"
done
