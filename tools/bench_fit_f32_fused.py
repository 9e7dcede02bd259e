"""Developer tool: the fp32-operand fit (reference default `--dtype float32`) at the BASELINE configuration, fused row kernel
(round 5, dvt_tune_set(6, 3)) against the layer-by-layer launches (dvt_tune_set(6, 2)): ms per 1000-step fit, interleaved."""
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
C = int(sys.argv[1]) if len(sys.argv) > 1 else 768
V, H, T = 769, 37, 1000
n_rows = V * H * H
feats = torch.randn(n_rows, C, device=dev)
xy = torch.rand(n_rows, 2, device=dev)
idx = np.random.RandomState(0).randint(0, n_rows, (T, 2048)).astype(np.int32)
res = {}
for mode in ("float32", "bfloat16"):
    for rnd in range(3):
        for fused in ((3, 2) if mode == "float32" else (3,)):
            L.dvt_tune_set(6, fused)
            eng = FitEngine(FitSettings(feat_dim=C, num_iters=T, warmup_iters=100, mlp_dtype=mode), n_rows, dev)
            eng.reset(torch.Generator(device=dev).manual_seed(0))
            eng.fit(feats, xy, idx, log_every=0, step_begin=0, step_end=50)
            torch.cuda.synchronize()
            eng.reset(torch.Generator(device=dev).manual_seed(0))
            t0 = time.perf_counter()
            eng.fit(feats, xy, idx, log_every=0)
            torch.cuda.synchronize()
            res.setdefault((mode, fused), []).append((time.perf_counter() - t0) * 1e3)
            del eng
L.dvt_tune_set(6, 3)
for (mode, fused), v in res.items():
    name = "fused row kernel" if fused == 3 else "layer-by-layer"
    print(f"C={C} {mode:9s} {name:17s}: {sorted(v)[len(v) // 2]:7.1f} ms per 1000-step fit (min {min(v):7.1f}) = {sorted(v)[len(v) // 2]:.0f} us per step")
