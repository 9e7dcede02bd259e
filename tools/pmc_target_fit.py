"""Small workload for PMC collection on the fit step: 40 Adam steps (20 per phase) at BASELINE sizes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.fit import FitEngine, FitSettings  # noqa: E402

dev = torch.device("cuda:0")
n_rows = 128 * 1369
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, 768, device=dev, generator=g)
xy = torch.rand(n_rows, 2, device=dev, generator=g)
eng = FitEngine(FitSettings(num_iters=40, warmup_iters=4, mlp_dtype=os.environ.get("FIT_DTYPE", "bfloat16")), n_rows, dev)
eng.reset(g)
np.random.seed(0)
eng.fit(feat, xy, None, log_every=0)
torch.cuda.synchronize()
