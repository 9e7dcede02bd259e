"""interleaved A/B: product 8p (schedule 4) vs its stamp build (12: + s_waitcnt vmcnt(0) before exit + 5 stamps)"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib
from tools.labenv import use_lab_library
dev = torch.device("cuda:0")
L = use_lab_library()
L.dvt_vit_debug_buffer.argtypes = [C.c_void_p]; L.dvt_vit_debug_buffer.restype = C.c_int
M = 398 * 1376 // 256 * 256
torch.manual_seed(0)
for name, n, k in (("qkv", 2304, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    x = torch.randn(M, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    b = torch.randn(n, device=dev); y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    dbg = torch.zeros((M // 256) * (n // 256) * 8, device=dev, dtype=torch.int32)
    L.dvt_vit_debug_buffer(dbg.data_ptr())
    times = {4: [], 12: []}
    for rnd in range(7):
        for v in (4, 12):
            L.dvt_tune_set(1, v)
            L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
            e1.record(); torch.cuda.synchronize()
            if rnd: times[v].append(e0.elapsed_time(e1) / 4 * 1e3)
    for v in (4, 12):
        t = np.array(times[v]); print(f"{name} schedule {v:2d}: min {t.min():8.1f} median {np.median(t):8.1f} us", flush=True)
    L.dvt_vit_debug_buffer(None)
L.dvt_tune_set(1, 4)
