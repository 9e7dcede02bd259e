"""Developer tool: fit-only us/step for a few dvt_tune_set settings (bf16 fused mode), BASELINE sizes."""
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
idx = torch.from_numpy(np.random.RandomState(0).randint(0, n_rows, (1000, 2048)).astype(np.int32)).to(dev)
eng = FitEngine(FitSettings(num_iters=1000, warmup_iters=100, mlp_dtype="bfloat16"), n_rows, dev)
L = _lib.lib()
for name, knobs in (("default (shadow in Adam)", {}), ("shadow_build_kernel", {12: 0}), ("default again", {}), ("separate catch-up + shadow", {11: 0})):
    for k, v in {7: 1, 8: 1, 9: 32, 10: 0, 11: 1, 12: 1, **knobs}.items():
        _lib.check(L.dvt_tune_set(k, v))
    for rep in range(2):
        eng.reset(g)
        torch.cuda.synchronize()
        ts = []
        for lo, hi in ((0, 500), (500, 1000)):
            t0 = time.perf_counter()
            eng.fit(feat, xy, idx, log_every=0, step_begin=lo, step_end=hi)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / (hi - lo) * 1e6)
    print(f"{name:32s}: phase 1 {ts[0]:7.1f} us/step, phase 2 {ts[1]:7.1f} us/step", flush=True)
L.dvt_tune_set(7, 1)
L.dvt_tune_set(9, 32)
L.dvt_tune_set(10, 0)
L.dvt_tune_set(11, 1)
L.dvt_tune_set(12, 1)
L.dvt_tune_set(8, 1)
