"""Developer tool: extractor GEMM rate vs the tile order of the 8-phase kernel (M panels per block) and
non-temporal output stores."""
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
ref = torch.empty_like(out)
L = _lib.lib()


def run(tag, check=False):
    vit.features_nhwc(x, out=out)
    torch.cuda.synchronize()
    _lib.prof_enable(["vit_gemm", "vit_attn"])
    t0 = time.perf_counter()
    vit.features_nhwc(x, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    g, a = _lib.prof_collect("vit_gemm"), _lib.prof_collect("vit_attn")
    _lib.prof_enable([])
    same = "" if not check else f" bit-equal to default: {bool(torch.equal(out, ref))}"
    print(f"{tag:34s}: {t/256*769*1e3:6.1f} ms per 769 views; gemm {g['total_ms']:6.1f} ms "
          f"{g['work']/g['total_ms']/1e9:6.1f} TF/s; attn {a['work']/a['total_ms']/1e9:6.1f} TF/s{same}", flush=True)


run("default (mblock 1, nt 0)")
ref.copy_(out)
for nt in (0, 1):
    L.dvt_tune_set(1, -50 - nt)
    for mb in (1, 2, 4, 8, 16):
        L.dvt_tune_set(1, -100 - mb)
        run(f"mblock {mb}, nt {nt}", check=True)
for kib in (1200, 2400):
    L.dvt_tune_set(1, kib)
    for mb in (1, 4, 8):
        L.dvt_tune_set(1, -100 - mb)
        run(f"group {kib} KiB, mblock {mb}, nt 1", check=True)
