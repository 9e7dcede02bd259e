"""Fit-only microbenchmark (developer tool): BASELINE config-2 sizes, synthetic features.
Variants: Adam zero_all on/off x grid LDS threshold; and k concurrent fits on k streams."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
from dvt_amd.fit import FitEngine, FitSettings, fit_many  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=1000)
ap.add_argument("--views", type=int, default=769)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--concurrent", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
C, HW = 768, 1369
n_rows = a.views * HW
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, C, device=dev, generator=g)
xy = torch.rand(n_rows, 2, device=dev, generator=g)
s = FitSettings(num_iters=a.iters, warmup_iters=a.iters // 10)
L = _lib.lib()
np.random.seed(0)
eng = FitEngine(s, n_rows, dev)
def run_once(tag):
    for rep in range(a.reps):
        eng.reset(g)
        idx = torch.from_numpy(FitEngine.sample_indices(n_rows, a.iters, 2048)).to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.fit(feat, xy, idx, log_every=1000)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    print(f"{tag}: {t/a.iters*1e6:.1f} us/step", flush=True)
    return idx


run_once("single fit")
# k batched fits (shared launches, dvt_fit_run_batched)
for k in (2, 3, 4)[: max(0, a.concurrent - 1)]:
    engines = [eng] + [FitEngine(s, n_rows, dev) for _ in range(k - 1)]
    for rep in range(a.reps):
        for e in engines:
            e.reset(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit_many(engines, [feat] * k, [xy] * k, None, log_every=1000)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    print(f"{k} batched fits: {t/a.iters*1e6:.1f} us/step = {t/a.iters*1e6/k:.1f} us/step/image", flush=True)
    del engines
