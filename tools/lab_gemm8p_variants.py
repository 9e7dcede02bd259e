"""Developer tool: A/B timing of the 8p GEMM and its experiment builds (dvt_tune_set(1, 5) + dvt_tune_set(1, -300 - abl)):
variant 5 = "8m" (DMA issue inside the MFMA segments), 10 = "8h" (two phases per k-tile).  Round 4 also measured, and removed: accumulators in AGPRs, lgkmcnt(0) right after
the fragment reads / before the barrier (profiles/r04/r04w_*): all within 2 % of 8p.  Variants are interleaved; min and median."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
from tools.labenv import use_lab_library  # noqa: E402
L = use_lab_library()  # schedules / timing builds of csrc/lab/: the developer library, not the product one
M = int(sys.argv[1]) if len(sys.argv) > 1 else 110 * 1408
cases = [(4, 0)] + [(int(v), 0) for v in (sys.argv[2] if len(sys.argv) > 2 else "5,10").split(",")]
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
torch.manual_seed(0)
for name, n, k in shapes:
    x = torch.randn(M, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    b = torch.randn(n, device=dev)
    y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    ref = None
    times = {c: [] for c in cases}
    for rnd in range(6):
        for c in cases:
            L.dvt_tune_set(1, c[0])
            L.dvt_tune_set(1, -300 - c[1])
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
            ev0.record()
            for _ in range(4):
                L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
            ev1.record()
            torch.cuda.synchronize()
            if rnd:
                times[c].append(ev0.elapsed_time(ev1) / 4 * 1e3)
            if rnd == 0:
                if ref is None:
                    ref = y.clone()
                else:
                    assert torch.equal(ref.view(torch.int16), y.view(torch.int16)), (name, c)
    for c in cases:
        t = np.array(times[c])
        print(f"{name:5s} M={M} N={n:5d} K={k:5d}  variant {c[0]} abl {c[1]}: min {t.min():8.1f} us  median {np.median(t):8.1f} us  "
              f"{2.0 * M * n * k / np.median(t) / 1e6:7.1f} TF/s", flush=True)
L.dvt_tune_set(1, 4)
L.dvt_tune_set(1, -300)
