"""interleaved: product 8p (4) vs nt on the A stream (14), each with / without nt output stores (-51 / -50); bias + resid + gelu epilogues"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib
from tools.labenv import use_lab_library
dev = torch.device("cuda:0")
L = use_lab_library()
M = 398 * 1376 // 256 * 256
torch.manual_seed(0)
S = _lib.stream
cases = [("qkv bias", "bias", 2304, 768), ("fc1 gelu", "gelu", 3072, 768), ("proj resid", "resid", 768, 768), ("fc2 resid", "resid", 768, 3072)]
for name, kind, n, k in cases:
    x = torch.randn(M, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16(); b = torch.randn(n, device=dev)
    if kind == "resid":
        gm, xres = torch.randn(n, device=dev) * 0.1, torch.randn(M, n, device=dev)
        call = lambda: L.dvt_vit_gemm_residual(x.data_ptr(), w.data_ptr(), b.data_ptr(), gm.data_ptr(), xres.data_ptr(), M, n, k, S())
    elif kind == "gelu":
        y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        stats = torch.stack([torch.randn(M, device=dev) * 0.3, torch.rand(M, device=dev) + 0.5], 1).contiguous(); cs = w.float().sum(1).contiguous()
        call = lambda: L.dvt_vit_gemm_lnfold(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, stats.data_ptr(), cs.data_ptr(), 1, S())
    else:
        y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        call = lambda: L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, S())
    cfgs = [(4, -50), (14, -50), (4, -51), (14, -51)]
    times = {c: [] for c in cfgs}
    for rnd in range(6):
        for c in cfgs:
            L.dvt_tune_set(1, c[0]); L.dvt_tune_set(1, c[1])
            call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): call()
            e1.record(); torch.cuda.synchronize()
            if rnd: times[c].append(e0.elapsed_time(e1) / 3 * 1e3)
    for c in cfgs:
        t = np.array(times[c]); print(f"{name:11s} schedule {c[0]:2d} nt-stores {'on ' if c[1] == -51 else 'off'}: min {t.min():8.1f} median {np.median(t):8.1f} us", flush=True)
L.dvt_tune_set(1, 4); L.dvt_tune_set(1, -50)
