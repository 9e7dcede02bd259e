"""Developer tool (CAVEAT: the stamps perturb the loop -- 3300-3600 cycles per k-tile with them, 2360-2530 without, see
tools/lab_gemm8p_ablate.py and profiles/r04/r04w_*; use the intervals as a picture of the instrumented build only): cycle anatomy of the 8p GEMM's k-loop from its timing build (dvt_tune_set(1, 5) + (1, -303)): s_memtime
after each of the 8 barriers of k-tiles 4 and 5, for one wave of each wave group (wave 0 = group 0, wave 4 = group 1, which runs
half a phase behind), every workgroup.  Prints the eight barrier-to-barrier intervals of a k-tile (shader cycles).

Interval i runs from the stamp after barrier i to the stamp after barrier i + 1.  For group 0: even i = MFMA segment of phase
i / 2 + 1 (the stamp sits after the first barrier of the phase = after its load segment), odd i = load segment of the next phase.
"""
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
import ctypes as C  # noqa: E402
L.dvt_vit_debug_buffer.argtypes = [C.c_void_p]
L.dvt_vit_debug_buffer.restype = C.c_int
M = int(sys.argv[1]) if len(sys.argv) > 1 else 110 * 1408
shapes = [("qkv", 2304, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
if len(sys.argv) > 2:  # "name:N:K,..." -- e.g. 2048 rows x "wide:8192:768,wide:8192:3072": 256 tiles whose operands stay on chip
    shapes = [(a.split(":")[0], int(a.split(":")[1]), int(a.split(":")[2])) for a in sys.argv[2].split(",")]
torch.manual_seed(0)
for name, n, k in shapes:
    x = torch.randn(M, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    b = torch.randn(n, device=dev)
    y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    tiles = (M // 256) * (n // 256)
    dbg = torch.zeros(tiles * 2 * 24, device=dev, dtype=torch.int32)
    assert L.dvt_vit_debug_buffer(dbg.data_ptr()) == 0
    for variant, abl in ((4, 0), (5, 3)):
        L.dvt_tune_set(1, variant)
        L.dvt_tune_set(1, -300 - abl)
        for _ in range(2):
            assert L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream()) == 0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
        ev1.record()
        torch.cuda.synchronize()
        print(f"{name} M={M} N={n} K={k}: {'timing build' if abl else '8p          '} {ev0.elapsed_time(ev1) / 5 * 1e3:8.1f} us per launch")
    L.dvt_tune_set(1, 4)
    L.dvt_tune_set(1, -300)
    L.dvt_vit_debug_buffer(None)
    st = dbg.cpu().numpy().astype(np.uint32).reshape(tiles, 2, 24)
    for grp in (0, 1):
        s = st[:, grp, :].astype(np.int64)
        ok = (s[:, :12] != 0).all(axis=1) & (s[:, 13] != 0)
        s = s[ok]
        dif = lambda a, b: np.median((s[:, a] - s[:, b]) & 0xFFFFFFFF)
        print(f"  wave group {grp}: {ok.sum()} workgroups; cycles per k-tile (after barrier 0 of tile 4 -> of tile 5): {dif(13, 0):.0f}")
        print("    barrier-to-barrier intervals 0..6 (M1 L2 M2 L3 M3 L4 M4), median cycles:", " ".join(f"{dif(i + 1, i):5.0f}" for i in range(7)),
              f"| L1 of the next tile: {dif(13, 7):5.0f}")
        print(f"    inside P1 of tile 4: fragment reads issued + completed (from the previous barrier is not stamped; to stamp 8) ... "
              f"stage issue {dif(9, 8):.0f}, vmcnt wait {dif(10, 9):.0f}, barrier 0 {dif(0, 10):.0f}, "
              f"lgkm + 16 MFMAs issued {dif(11, 0):.0f}, barrier 1 {dif(1, 11):.0f}")
    del dbg
