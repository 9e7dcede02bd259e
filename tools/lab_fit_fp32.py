"""Lab: fit-only timing at BASELINE configs[1] sizes in both operand precisions, with the round-4 fp32 path (sorted grid
lists + lazy IEEE Adam, dvt_tune_set(7, 1) / (9, 1)) against the round-3 one (atomics + dense Adam: (7, 0), (9, 0)); us per
step in each phase (one stream, nothing else on the GPU) and per-kernel launch statistics from the library's probes."""
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
L = _lib.lib()
n_rows = 769 * 1369
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, 768, device=dev, generator=g)
xy = torch.rand(n_rows, 2, device=dev, generator=g)
idx = np.random.RandomState(0).randint(0, n_rows, (1000, 2048)).astype(np.int32)
didx = torch.from_numpy(idx).to(dev)
for name, dtype, lists, lazy in (("bf16 fused (default)", "bfloat16", 1, 1), ("fp32, lists + lazy IEEE Adam (round 4)", "float32", 1, 1),
                                 ("fp32, lists, dense Adam", "float32", 1, 0), ("fp32, atomics + dense Adam (round 3)", "float32", 0, 0)):
    _lib.check(L.dvt_tune_set(7, lists))
    _lib.check(L.dvt_tune_set(9, lazy))
    eng = FitEngine(FitSettings(num_iters=1000, warmup_iters=100, mlp_dtype=dtype), n_rows, dev)
    for rep in range(2):
        eng.reset(torch.Generator(device=dev).manual_seed(1))
        torch.cuda.synchronize()
        ts = []
        for lo, hi in ((0, 500), (500, 1000)):
            t0 = time.perf_counter()
            eng.fit(feat, xy, didx, log_every=0, step_begin=lo, step_end=hi)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / (hi - lo) * 1e6)
    out = eng.infer(xy[:1369]).float()
    print(f"{name:42s}: phase 1 {ts[0]:7.1f} us/step, phase 2 {ts[1]:7.1f} us/step, image {(ts[0] + ts[1]) * 0.5:7.1f} ms; "
          f"checksum {float(out.double().sum()):.6f} grads max {float(eng.grads.abs().max()):.1e}", flush=True)
    del eng
L.dvt_tune_set(7, 1)
L.dvt_tune_set(9, 1)
