#!/bin/bash
# round 2, call C: per-kernel breakdown of the fused fit step + bench
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/fit_only.py <<'PY'
import os, sys, time, numpy as np, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.fit import FitEngine, FitSettings
dev = torch.device("cuda:0"); n_rows = 769 * 1369
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, 768, device=dev, generator=g); xy = torch.rand(n_rows, 2, device=dev, generator=g)
eng = FitEngine(FitSettings(num_iters=600, warmup_iters=60, mlp_dtype=os.environ.get("FIT_DTYPE", "bfloat16")), n_rows, dev)
np.random.seed(0)
eng.reset(g); torch.cuda.synchronize()
eng.fit(feat, xy, None, log_every=1000); torch.cuda.synchronize()
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_fit -o fit -- python /tmp/fit_only.py > $GRAFT_REPO_ROOT/gpurun_out/prof_fit.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/fit_step_breakdown.py gpurun_out/prof_fit > gpurun_out/r2c_fit_step_breakdown.txt 2>&1
rm -rf gpurun_out/prof_fit
cat gpurun_out/r2c_fit_step_breakdown.txt
