"""Developer tool: extractor time for 769 views as a function of the views-per-launch batch (wave
quantisation of the 256x256 GEMM tiles over 256 CUs)."""
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.models import PretrainedViTWrapper  # noqa: E402

dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
x = torch.randn(769, 3, 518, 518, device=dev)
out = torch.empty(769, 37, 37, 768, device=dev)
eng = vit._engine(dev)
for mb in [int(a) for a in sys.argv[1:]] or [128, 124, 186, 128, 124, 186]:
    eng.forward_features(x[:mb], n_blocks=12, out=out[:mb], max_batch=mb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.forward_features(x, n_blocks=12, out=out, max_batch=mb)
    torch.cuda.synchronize()
    print(f"max_batch {mb}: 769 views {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
