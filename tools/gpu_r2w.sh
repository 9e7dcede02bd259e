#!/bin/bash
export TMPDIR=/tmp FIT_DTYPE=bfloat16
mkdir -p gpurun_out
cat > /tmp/fit_only.py <<'PY'
import os, sys, time, numpy as np, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.fit import FitEngine, FitSettings
dev = torch.device("cuda:0"); n_rows = 769 * 1369
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, 768, device=dev, generator=g); xy = torch.rand(n_rows, 2, device=dev, generator=g)
eng = FitEngine(FitSettings(num_iters=1000, warmup_iters=100, mlp_dtype="bfloat16"), n_rows, dev)
np.random.seed(0)
eng.reset(g); torch.cuda.synchronize()
eng.fit(feat, xy, None, log_every=1000); torch.cuda.synchronize()
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fit2 -o fit -- python /tmp/fit_only.py > $GRAFT_REPO_ROOT/gpurun_out/prof_fit2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_fit2 -name '*.db' | head -1) > gpurun_out/r2w_fit_kernel_stats.txt; head -16 gpurun_out/r2w_fit_kernel_stats.txt | cut -c1-160
rm -rf gpurun_out/prof_fit2
