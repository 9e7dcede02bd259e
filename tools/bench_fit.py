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
ap.add_argument("--concurrent", type=int, default=2)
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
for zero_all, lds_max in ((1, 0), (0, 0), (1, 40960), (0, 40960)):
    L.dvt_tune_set(3, zero_all)
    L.dvt_tune_set(2, lds_max)
    for rep in range(a.reps):
        eng.reset(g)
        idx = torch.from_numpy(FitEngine.sample_indices(n_rows, a.iters, 2048)).to(dev)
        _lib.prof_enable(["adam", "grid", "fit_gemm"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.fit(feat, xy, idx, log_every=1000)
        t_launch = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        pr = {n: _lib.prof_collect(n) for n in ("adam", "grid", "fit_gemm")}
        _lib.prof_enable([])
    print(f"zero_all={zero_all} lds_max={lds_max}: {t/a.iters*1e6:.1f} us/step (host enqueue {t_launch*1e3:.0f} ms); "
          + ", ".join(f"{n} {p['total_ms']/a.iters*1e3:.1f} us/step" for n, p in pr.items()), flush=True)
L.dvt_tune_set(3, 1)
L.dvt_tune_set(2, 0)
# host-side enqueue cost with the GPU idle-ish: tiny number of steps
eng.reset(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.fit(feat, xy, idx, log_every=0, step_begin=0, step_end=50)
print(f"host enqueue of 50 steps (queue not full): {(time.perf_counter()-t0)/50*1e6:.1f} us/step", flush=True)
torch.cuda.synchronize()
# k concurrent fits on k streams
k = a.concurrent
engines = [eng] + [FitEngine(s, n_rows, dev) for _ in range(k - 1)]
streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
for rep in range(a.reps):
    for e, st in zip(engines, streams):
        with torch.cuda.stream(st):
            e.reset(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit_many(engines, [feat] * k, [xy] * k, streams, log_every=1000)
    t_launch = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{k} concurrent fits: {t*1e3:.1f} ms total = {t/k*1e3:.1f} ms per image ({t/a.iters*1e6:.1f} us per lock-step), host enqueue {t_launch*1e3:.0f} ms", flush=True)
print({i: v for i, v in engines[-1].loss_log().items()})
