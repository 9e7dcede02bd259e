"""Developer tool: fp32 extractor (`--dtype float32`), A/B of the linear layers' GEMM kernel -- dvt_tune_set(4, 10): 64 x 64 x 64
LDS-DMA tile (rounds 2-4), (4, 11): 128 x 128 x 32 tile (round 5, default) -- on one launch plan of `N` views, interleaved.
    python tools/bench_vit_f32_ab.py [views]"""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = _lib.lib()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
x = torch.randn(N, 3, 518, 518, device=dev)
outs = {}
res = {10: [], 11: []}
for rnd in range(3):
    for v in (10, 11):
        assert L.dvt_tune_set(4, v) == 0
        out = torch.empty(N, 37, 37, 768, device=dev)
        vit.features_nhwc(x, out=out, dtype="float32")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vit.features_nhwc(x, out=out, dtype="float32")
        torch.cuda.synchronize()
        res[v].append(time.perf_counter() - t0)
        outs[v] = out
L.dvt_tune_set(4, 11)
d = float((outs[10] - outs[11]).abs().max() / outs[10].abs().max())
gemm_flops = 179.9e12 / 769 * N
for v, name in ((10, "64x64x64 tile"), (11, "128x128x32 tile")):
    t = sorted(res[v])[1]
    print(f"fp32 extractor, {N} views, {name}: {t * 1e3:.1f} ms (min {min(res[v]) * 1e3:.1f}) = {t / N * 769:.3f} s per 769 views", flush=True)
print(f"max |difference| between the two / max |value|: {d:.2e} (summation order only)")
