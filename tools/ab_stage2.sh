# stage-2 training step A/B on one box: data-gradient GEMMs on the 64 x 64 tile (DVT_S2_BIG_BWD=0) against the 128 x 128 x 32 tile
for i in 1 2; do
  for v in 1 0; do echo -n "DVT_S2_BIG_BWD=$v: "; DVT_S2_BIG_BWD=$v python tools/bench_stage2.py 2>/dev/null | tail -1; done
done
