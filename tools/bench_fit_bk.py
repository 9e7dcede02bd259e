"""Developer tool: serial fit us/step for the fp32 GEMM BK variants + per-kernel probes."""
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
C, HW, views = 768, 1369, 769
n_rows = views * HW
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, C, device=dev, generator=g)
xy = torch.rand(n_rows, 2, device=dev, generator=g)
s = FitSettings(num_iters=1000, warmup_iters=100, mlp_dtype=os.environ.get("FIT_DTYPE", "float32"))
eng = FitEngine(s, n_rows, dev)
print("mlp_dtype", s.mlp_dtype)
np.random.seed(0)
idx = torch.from_numpy(FitEngine.sample_indices(n_rows, 1000, 2048)).to(dev)
L = _lib.lib()
for key, val in [(5, 32)] + [tuple(map(int, a.split("="))) for a in sys.argv[1:]]:
    L.dvt_tune_set(key, val)
    for rep in range(2):
        eng.reset(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.fit(feat, xy, idx, log_every=1000)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    eng.reset(g)
    _lib.prof_enable(["fit_gemm"])
    eng.fit(feat, xy, idx, log_every=1000)
    torch.cuda.synchronize()
    pg = _lib.prof_collect("fit_gemm")
    _lib.prof_enable([])
    print(f"tune {key}={val}: {t*1e3:.1f} us/step; fit_gemm probes: {pg['total_ms']:.1f} ms / {pg['launches']} launches "
          f"= {pg['total_ms']/pg['launches']*1e3:.2f} us each", flush=True)
