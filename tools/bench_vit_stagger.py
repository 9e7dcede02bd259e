"""Developer tool (product library): de-synchronised start of the 8p GEMM workgroups (dvt_tune_set(1, -700 - pct)) -- per-shape
timing of the four ViT-B GEMMs with their real epilogues at M = 384 views, interleaved rounds, then the whole extractor.
    python tools/bench_vit_stagger.py [pct,pct,...]"""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
import dvt_amd.vit  # noqa: E402,F401
from dvt_amd.models import PretrainedViTWrapper  # noqa: E402

dev = torch.device("cuda:0")
L, S = _lib.lib(), _lib.stream
pcts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,50,100,150").split(",")]
M = 384 * 1408
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape, scale=1.0):
    return ((torch.rand(*shape, device=dev, generator=g) * 2 - 1) * scale)


def timeit(fn, reps=8):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e3


shapes = [("qkv-type (bias)", 2304, 768, "bias"), ("fc1 (LN fold + GELU)", 3072, 768, "gelu"), ("proj (residual)", 768, 768, "resid"),
          ("fc2 (residual)", 768, 3072, "resid")]
for name, n, k, kind in shapes:
    x = rnd(M, k).bfloat16()
    w = rnd(n, k, scale=k ** -0.5).bfloat16()
    b = rnd(n)
    if kind == "resid":
        gm, xr = rnd(n, scale=1e-3), rnd(M, n)
        fn = lambda: L.dvt_vit_gemm_residual(x.data_ptr(), w.data_ptr(), b.data_ptr(), gm.data_ptr(), xr.data_ptr(), M, n, k, S())  # noqa: E731
    else:
        y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        if kind == "gelu":
            st = torch.stack([rnd(M, scale=0.3), rnd(M) * 0.5 + 1.0], 1).contiguous()
            cs = w.float().sum(1).contiguous()
            fn = lambda: L.dvt_vit_gemm_lnfold(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, st.data_ptr(), cs.data_ptr(), 1, S())  # noqa: E731
        else:
            fn = lambda: L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, S())  # noqa: E731
    assert fn() == 0
    res = {p: [] for p in pcts}
    for rnd_i in range(4):
        for p in pcts:
            assert L.dvt_tune_set(1, -700 - p) == 0
            res[p].append(timeit(fn))
    L.dvt_tune_set(1, -700)
    print(f"{name:22s} M={M} N={n} K={k}: " + "  ".join(f"stagger {p:3d} %: {sorted(v)[len(v) // 2]:7.1f} us (min {min(v):7.1f})" for p, v in res.items()), flush=True)
    del x, w

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
xv = torch.randn(398, 3, 518, 518, device=dev)
out = torch.empty(398, 37, 37, 768, device=dev)
vit.features_nhwc(xv, out=out, max_batch=400)
torch.cuda.synchronize()
res = {p: [] for p in pcts}
for rnd_i in range(3):
    for p in pcts:
        L.dvt_tune_set(1, -700 - p)
        res[p].append(timeit(lambda: vit.features_nhwc(xv, out=out, max_batch=400), reps=2) / 1e3)
L.dvt_tune_set(1, -700)
print("extractor, 398 views, ms per launch: " + "  ".join(f"stagger {p:3d} %: {sorted(v)[len(v) // 2]:7.2f} (min {min(v):7.2f})" for p, v in res.items()))
