"""Developer tool: extractor time vs (a) the L2 N-tile group budget of the GEMM rasterisation and (b) the
views-per-launch batch (memory-side-cache residency of the activations)."""
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
L = _lib.lib()


def run(tag, mb):
    vit.features_nhwc(x, out=out, max_batch=mb)
    torch.cuda.synchronize()
    _lib.prof_enable(["vit_gemm", "vit_attn"])
    t0 = time.perf_counter()
    vit.features_nhwc(x, out=out, max_batch=mb)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    g, a = _lib.prof_collect("vit_gemm"), _lib.prof_collect("vit_attn")
    _lib.prof_enable([])
    print(f"{tag:28s}: 256 views {t*1e3:7.1f} ms ({t/256*769*1e3:6.1f} ms per 769 views); gemm {g['total_ms']:6.1f} ms "
          f"{g['work']/g['total_ms']/1e9:6.1f} TF/s; attn {a['total_ms']:6.1f} ms {a['work']/a['total_ms']/1e9:6.1f} TF/s",
          flush=True)


for kib in (2400, 4800, 9600, 1200, 2400):
    L.dvt_tune_set(1, kib)
    run(f"group budget {kib} KiB, b128", 128)
L.dvt_tune_set(1, 2400)
for mb in (64, 32, 16, 128):
    run(f"batch {mb}", mb)
