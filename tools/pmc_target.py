"""Small workload for PMC collection: one ViT forward of NV views in ONE launch (bench: 769 views = 398 + 371; default 110) + 60
fit steps.  argv[1] overrides the view count."""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.fit import FitEngine, FitSettings  # noqa: E402
from dvt_amd.models import PretrainedViTWrapper  # noqa: E402

dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
NV = int(sys.argv[1]) if len(sys.argv) > 1 else 110
x = torch.randn(NV, 3, 518, 518, device=dev)
out = torch.empty(NV, 37, 37, 768, device=dev)
vit.features_nhwc(x, out=out, max_batch=400)  # ONE launch of NV views (the driver's cap; the library default is 128)
torch.cuda.synchronize()
n_rows = NV * 1369
eng = FitEngine(FitSettings(num_iters=60, warmup_iters=6, mlp_dtype="bfloat16"), n_rows, dev)
eng.reset(torch.Generator(device=dev).manual_seed(0))
np.random.seed(0)
eng.fit(out.view(-1, 768), torch.rand(n_rows, 2, device=dev), None, log_every=0)
torch.cuda.synchronize()
