F="--no-cpu-baseline --no-fp32-fit --no-vit-large --no-stage2"
for i in 1 2; do
  bash tools/gpu.sh "bench:$(echo $F --tune 1=-530 | tr ' ' ':')" "bench:$(echo $F | tr ' ' ':')"
done
