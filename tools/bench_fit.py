"""Fit-only microbenchmark (developer tool): BASELINE config-2 sizes, synthetic features."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.fit import FitEngine, FitSettings  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=1000)
ap.add_argument("--views", type=int, default=769)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
C, HW = 768, 1369
n_rows = a.views * HW
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, C, device=dev, generator=g)
xy = torch.rand(n_rows, 2, device=dev, generator=g)
s = FitSettings(num_iters=a.iters, warmup_iters=a.iters // 10)
eng = FitEngine(s, n_rows, dev)
np.random.seed(0)
for rep in range(a.reps):
    eng.reset(g)
    idx = torch.from_numpy(FitEngine.sample_indices(n_rows, a.iters, 2048)).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.fit(feat, xy, idx, log_every=1000)
    t_launch = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"rep {rep}: {a.iters} steps in {t*1e3:.1f} ms ({t/a.iters*1e6:.1f} us/step), host launch {t_launch*1e3:.1f} ms",
          flush=True)
print(eng.loss_log())
