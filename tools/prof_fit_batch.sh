#!/bin/bash
# rocprofv3 kernel stats of the fit alone at K = 1 and K = 4 concurrent fits (shared launches), C = 768, 300 steps
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${DVT_TAG:-run}; mkdir -p $O
cd /tmp
for c in 1:1 4:4; do
  n=$(echo $c | tr ':' '_')
  DVT_FB_CASES=$c timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fb_$n -o fb -- python $R/tools/bench_fit_batch.py 768 300 > $O/prof_fb_$n.log 2>&1
  python $R/tools/rocpd_stats.py $(find $O/prof_fb_$n -name '*.db' | head -1) > $O/fit_kernel_stats_K$n.txt
  rm -rf $O/prof_fb_$n
  echo "== K $c"; tail -1 $O/prof_fb_$n.log; head -14 $O/fit_kernel_stats_K$n.txt | cut -c1-170
done
