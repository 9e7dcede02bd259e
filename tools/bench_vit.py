"""Developer tool: ViT extractor timing (one batch of 128 views) with probes; L2 group sweep."""
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
from dvt_amd.models import PretrainedViTWrapper  # noqa: E402

dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
x = torch.randn(256, 3, 518, 518, device=dev)
out = torch.empty(256, 37, 37, 768, device=dev)
from tools.labenv import use_lab_library  # noqa: E402
L = use_lab_library()  # schedules / timing builds of csrc/lab/: the developer library, not the product one
for kib in (0, 4, 0, 4):
    L.dvt_tune_set(1, kib)  # GEMM variant: 0 = 256x256 2-stage, 4 = 256x256 8-phase
    vit.features_nhwc(x, out=out)
    torch.cuda.synchronize()
    _lib.prof_enable(["vit_gemm", "vit_attn"])
    t0 = time.perf_counter()
    vit.features_nhwc(x, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    g, a = _lib.prof_collect("vit_gemm"), _lib.prof_collect("vit_attn")
    _lib.prof_enable([])
    print(f"gemm variant {kib}: 256 views {t*1e3:7.1f} ms ({t/256*769*1e3:6.1f} ms per 769 views); "
          f"gemm {g['total_ms']:6.1f} ms {g['work']/g['total_ms']/1e9:6.1f} TF/s; attn {a['total_ms']:6.1f} ms "
          f"{a['work']/a['total_ms']/1e9:6.1f} TF/s", flush=True)
L.dvt_tune_set(1, 4)
