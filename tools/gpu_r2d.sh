#!/bin/bash
# round 2, call D: fused row kernel v2 (deep prefetch, L2 warm-up): parity, per-kernel breakdown, fit timing, bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fit.py tests/test_gpu_parity_full.py -m gpu -q -s -p no:cacheprovider -k "fused or baseline or batched" > gpurun_out/r2d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d_pytest.log
grep -E "^\[|passed|failed|FAILED|ERROR|rc=|fused vs" gpurun_out/r2d_pytest.log | cut -c1-300 | tail -12
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
eng.fit(feat, xy, None, log_every=0); torch.cuda.synchronize()
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_fit -o fit -- python /tmp/fit_only.py > $GRAFT_REPO_ROOT/gpurun_out/prof_fit.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/fit_step_breakdown.py gpurun_out/prof_fit > gpurun_out/r2d_fit_step_breakdown.txt 2>&1
rm -rf gpurun_out/prof_fit
cat gpurun_out/r2d_fit_step_breakdown.txt
timeout 600 python tools/bench_fit_modes.py > gpurun_out/r2d_fit_modes.log 2>&1; tail -4 gpurun_out/r2d_fit_modes.log
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r2d_bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r2d_bench.log | cut -c1-260
