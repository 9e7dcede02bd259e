# pipelined bench A/B on one box: steps between full sweeps of the lazily updated (fine) grid levels, dvt_tune_set(9, n): 32 (default) / 64 / 128 / 16
F="--no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2"
for n in 32 64 128 16 32 64; do
  bash tools/gpu.sh "bench:$(echo $F --tune 9=$n | tr ' ' ':')" | grep "'value'" | cut -c1-60 | sed "s/^/refresh $n: /"
done
