#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" gpurun_out/pytest_gpu.log | tail -6
timeout 300 python tools/bench_gemm_f32.py > gpurun_out/bench_gemm_f32.log 2>&1; grep -E " -1 | layer" gpurun_out/bench_gemm_f32.log
timeout 300 python /dev/stdin > gpurun_out/fit_quick.log 2>&1 <<'PY'
import os, sys, time, numpy as np, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.fit import FitEngine, FitSettings
dev = torch.device("cuda:0"); n_rows = 769 * 1369
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, 768, device=dev, generator=g); xy = torch.rand(n_rows, 2, device=dev, generator=g)
eng = FitEngine(FitSettings(num_iters=1000, warmup_iters=100), n_rows, dev)
np.random.seed(0)
for rep in range(3):
    eng.reset(g); torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.fit(feat, xy, None, log_every=1000); torch.cuda.synchronize()
    print("fit us/step", (time.perf_counter() - t0) * 1e3)
PY
cat gpurun_out/fit_quick.log | grep fit
timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-220
