"""Developer tool: grid backward variants (LDS-level threshold) at BASELINE sizes."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402

L = _lib.lib()
dev = "cuda"
tbl = _lib.grid_table(16, 8, 16, 1024, 20)
n = 2048
torch.manual_seed(0)
xy = torch.rand(n, 2, device=dev)
denc = torch.randn(n, 128, device=dev)
g = torch.zeros(int(tbl.n_entries_total) * 8, device=dev)
touched = torch.zeros((int(tbl.n_entries_total) + 31) // 32 + 8, device=dev, dtype=torch.int32)
params = torch.randn_like(g)
enc = torch.empty(n, 128, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=100):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("fwd us:", timeit(lambda: L.dvt_grid_fwd(C.byref(tbl), xy.data_ptr(), params.data_ptr(), enc.data_ptr(), n, st)))
ref = None
for thr in (0, 300, 5000, 40960, 70000, 120000, 400000, 2000000):
    L.dvt_tune_set(2, thr)
    g.zero_(); touched.zero_()
    L.dvt_grid_bwd(C.byref(tbl), xy.data_ptr(), denc.data_ptr(), g.data_ptr(), touched.data_ptr(), n, st)
    torch.cuda.synchronize()
    if ref is None:
        ref = g.clone()
    err = float((g - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: L.dvt_grid_bwd(C.byref(tbl), xy.data_ptr(), denc.data_ptr(), g.data_ptr(), touched.data_ptr(), n, st))
    print(f"lds_level_max={thr:8d}: bwd {t:8.2f} us  (max rel diff vs all-atomic {err:.2e})")
L.dvt_tune_set(2, 40960)
