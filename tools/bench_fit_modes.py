"""Developer tool: fit-only timing at BASELINE configs[1] sizes, fused row kernel vs layer-by-layer
(bf16 mode) and the fp32 mode: us per step in each phase (one stream, nothing else on the GPU)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
from dvt_amd.fit import FitEngine, FitSettings  # noqa: E402

dev = torch.device("cuda:0")
n_rows = 769 * 1369
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, 768, device=dev, generator=g)
xy = torch.rand(n_rows, 2, device=dev, generator=g)
idx = np.random.RandomState(0).randint(0, n_rows, (1000, 2048)).astype(np.int32)
for name, dtype, fused in (("bf16 fused", "bfloat16", 1), ("bf16 layer-by-layer", "bfloat16", 0), ("fp32", "float32", 1)):
    _lib.check(_lib.lib().dvt_tune_set(6, fused))
    eng = FitEngine(FitSettings(num_iters=1000, warmup_iters=100, mlp_dtype=dtype), n_rows, dev)
    didx = torch.from_numpy(idx).to(dev)
    for rep in range(2):
        eng.reset(g)
        torch.cuda.synchronize()
        ts = []
        for lo, hi in ((0, 500), (500, 1000)):
            t0 = time.perf_counter()
            eng.fit(feat, xy, didx, log_every=0, step_begin=lo, step_end=hi)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / (hi - lo) * 1e6)
    print(f"{name:22s}: phase 1 {ts[0]:7.1f} us/step, phase 2 {ts[1]:7.1f} us/step, image {(ts[0] + ts[1]) * 0.5:7.1f} ms", flush=True)
    del eng
_lib.lib().dvt_tune_set(6, 1)
