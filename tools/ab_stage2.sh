# stage-2 training step A/B on one box: weight-gradient / data-gradient GEMMs on the 64 x 64 tile (DVT_S2_BIG_WGRAD=0 / DVT_S2_BIG_BWD=0)
# against the 128 x 128 x 32 tile (defaults)
for i in 1 2; do
  for v in "1 1" "0 1" "0 0"; do set -- $v; echo -n "DVT_S2_BIG_WGRAD=$1 DVT_S2_BIG_BWD=$2: "; DVT_S2_BIG_WGRAD=$1 DVT_S2_BIG_BWD=$2 python tools/bench_stage2.py 2>/dev/null | tail -1 | cut -c1-200; done
done
