# stage-2 training step A/B on one box (defaults: everything on): DVT_S2_FORK_WGRAD=0 = weight gradients on the caller's stream;
# DVT_S2_ATTN_ROWS=0 = the [Tp][Tp] products on the 64 x 64 GEMM tile + softmax passes; DVT_S2_BIG_WGRAD=0 / DVT_S2_BIG_BWD=0 = weight- /
# data-gradient GEMMs on the 64 x 64 tile
for i in 1 2; do
  for v in "" "DVT_S2_FORK_WGRAD=0" "DVT_S2_FORK_WGRAD=0 DVT_S2_ATTN_ROWS=0"; do echo -n "[$v] "; env $v python tools/bench_stage2.py 2>/dev/null | tail -1 | cut -c1-150; done
done
