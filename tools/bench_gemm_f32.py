"""Developer tool: time the fit's fp32-MFMA linear kernels per shape and tile config."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402

L = _lib.lib()
dev = "cuda"
B = 2048
shapes = [("field1", 384, 128), ("field2", 768, 384), ("h1", 192, 768), ("h2", 192, 192), ("h3", 768, 192)]
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"{'layer':8s} {'cfg':>4s} {'fwd us':>8s} {'wgrad us':>9s} {'dgrad us':>9s}   (ideal us @155TF: fwd=wgrad=dgrad)")
for name, n, k in shapes:
    x, w, b = torch.randn(B, k, device=dev), torch.randn(n, k, device=dev), torch.randn(n, device=dev)
    y, dy = torch.empty(B, n, device=dev), torch.randn(B, n, device=dev)
    dw, db, dx = torch.zeros(n, k, device=dev), torch.zeros(n, device=dev), torch.empty(B, k, device=dev)
    ideal = 2.0 * B * n * k / 155e12 * 1e6
    for cfg in (-1, 0, 1, 2, 3):
        L.dvt_tune_set(0, cfg)
        f = timeit(lambda: L.dvt_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, n, k, 1, st))
        wg = timeit(lambda: L.dvt_linear_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dw.data_ptr(), db.data_ptr(), None, None, B, n, k, st))
        dg = timeit(lambda: L.dvt_linear_bwd(dy.data_ptr(), None, w.data_ptr(), None, None, dx.data_ptr(), x.data_ptr(), B, n, k, st))
        print(f"{name:8s} {cfg:4d} {f:8.2f} {wg:9.2f} {dg:9.2f}   ({ideal:.2f})")
L.dvt_tune_set(0, -1)
