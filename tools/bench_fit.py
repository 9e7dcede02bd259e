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


for lds_max in (40960, 0, 300, 1500, 5000, 400000):
    L.dvt_tune_set(2, lds_max)
    idx = run_once(f"grid lds_level_max={lds_max}")
L.dvt_tune_set(2, 40960)
for target in (1024, 16384):
    L.dvt_tune_set(2, -target)
    run_once(f"grid lds atomics/block target={target}")
L.dvt_tune_set(2, -4096)
for cfg in (0, 1, 2):
    L.dvt_tune_set(0, cfg)
    run_once(f"f32 gemm tile cfg={cfg}")
L.dvt_tune_set(0, -1)
eng.reset(g)
_lib.prof_enable(["adam", "grid", "fit_gemm"])
eng.fit(feat, xy, idx, log_every=1000)
torch.cuda.synchronize()
print("probed (inflated by event overhead): " + ", ".join(
    f"{n} {_lib.prof_collect(n)['total_ms']/a.iters*1e3:.1f} us/step" for n in ("adam", "grid", "fit_gemm")), flush=True)
_lib.prof_enable([])
# host-side enqueue cost with the GPU idle-ish: tiny number of steps
eng.reset(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.fit(feat, xy, idx, log_every=0, step_begin=0, step_end=50)
print(f"host enqueue of 50 steps (queue not full): {(time.perf_counter()-t0)/50*1e6:.1f} us/step", flush=True)
torch.cuda.synchronize()
print({i: v for i, v in eng.loss_log().items()})
