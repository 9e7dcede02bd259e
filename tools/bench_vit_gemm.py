"""Developer tool: per-shape timing + correctness of the bf16 ViT GEMM variants (dvt_vit_gemm_bias)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
from tools.labenv import use_lab_library  # noqa: E402
L = use_lab_library()  # schedules / timing builds of csrc/lab/: the developer library, not the product one
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,4").split(",")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 128 * 1408
PAD = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L.dvt_tune_set(1, -PAD - 1)
if len(sys.argv) > 4:
    L.dvt_tune_set(1, int(sys.argv[4]))  # L2 group budget, KiB
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("patch", 768, 640)]
torch.manual_seed(0)
for name, n, k in shapes:
    xs = torch.randn(M, k + PAD, device=dev).bfloat16()
    ws = (torch.randn(n, k + PAD, device=dev) / k ** 0.5).bfloat16()
    x, w = xs[:, :k], ws[:, :k]
    b = torch.randn(n, device=dev)
    y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    # reference on a row sample (first / last tiles and a random middle block)
    rows = torch.cat([torch.arange(0, 512), torch.arange(M // 2 - 256, M // 2 + 256), torch.arange(M - 512, M)]).to(dev)
    want = x[rows].float() @ w.float().t() + b
    for v in variants:
        L.dvt_tune_set(1, v)
        y.zero_()
        rc = L.dvt_vit_gemm_bias(xs.data_ptr(), ws.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
        assert rc == 0, rc
        torch.cuda.synchronize()
        got = y[rows].float()
        err = float((got - want).abs().max() / want.abs().max())
        # full-matrix check through a checksum against torch's own bf16 GEMM (catches a wrong tile anywhere)
        ref_full = torch.addmm(b.bfloat16(), x, w.t()).float()
        bad = int(((y.float() - ref_full).abs() > 0.05 * ref_full.abs().max()).sum())
        del ref_full
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            ev0.record()
            for _ in range(5):
                L.dvt_vit_gemm_bias(xs.data_ptr(), ws.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
            ev1.record()
            torch.cuda.synchronize()
            best = min(best, ev0.elapsed_time(ev1) / 5)
        print(f"{name:6s} pad={PAD} M={M} N={n:5d} K={k:5d} variant {v}: {best*1e3:8.1f} us {2.0*M*n*k/best/1e9:7.1f} TF/s  "
              f"rel err {err:.2e} bad {bad}", flush=True)
L.dvt_tune_set(1, 4)
