"""Developer tool: host enqueue time vs GPU execution time of one 1000-step fit, on the legacy
default stream and on a created stream."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.fit import FitEngine, FitSettings  # noqa: E402

dev = torch.device("cuda:0")
C, HW, views = 768, 1369, 769
n_rows = views * HW
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, C, device=dev, generator=g)
xy = torch.rand(n_rows, 2, device=dev, generator=g)
s = FitSettings(num_iters=1000, warmup_iters=100)
eng = FitEngine(s, n_rows, dev)
np.random.seed(0)
idx = torch.from_numpy(FitEngine.sample_indices(n_rows, 1000, 2048)).to(dev)
for name, st in (("default stream", torch.cuda.current_stream()), ("created stream", torch.cuda.Stream()),
                 ("created stream", torch.cuda.Stream(priority=-1))):
    with torch.cuda.stream(st):
        for rep in range(2):
            eng.reset(g)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.fit(feat, xy, idx, log_every=1000)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"{name}: host enqueue {1e3*(t1-t0):.1f} ms, until done {1e3*(t2-t0):.1f} ms", flush=True)
