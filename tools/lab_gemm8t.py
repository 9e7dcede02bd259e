"""Developer tool: the persistent 8t GEMM (lab schedule 11) at several run lengths (tiles per workgroup, dvt_tune_set(1, -200 - n);
0 = one workgroup per CU for the whole launch) against the product's 8p (schedule 4), interleaved; bit-equality checked."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
from tools.labenv import use_lab_library  # noqa: E402

dev = torch.device("cuda:0")
L = use_lab_library()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 398 * 1408
runs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,2,4,8,16").split(",")]
cases = [(4, 0)] + [(11, r) for r in runs]
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
            L.dvt_tune_set(1, -200 - c[1])
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            y.zero_()
            L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
            if rnd == 0:
                torch.cuda.synchronize()
                if ref is None:
                    ref = y.clone()
                else:
                    assert torch.equal(ref.view(torch.int16), y.view(torch.int16)), (name, c)
            ev0.record()
            for _ in range(4):
                L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
            ev1.record()
            torch.cuda.synchronize()
            if rnd:
                times[c].append(ev0.elapsed_time(ev1) / 4 * 1e3)
    for c in cases:
        t = np.array(times[c])
        print(f"{name:5s} M={M} N={n:5d} K={k:5d}  schedule {c[0]:2d} tiles/wg {c[1]:2d}: min {t.min():8.1f} us  median {np.median(t):8.1f} us  "
              f"{2.0 * M * n * k / np.median(t) / 1e6:7.1f} TF/s", flush=True)
L.dvt_tune_set(1, 4)
L.dvt_tune_set(1, -200)
