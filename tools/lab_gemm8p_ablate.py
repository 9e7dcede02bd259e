"""Developer tool: the 8p GEMM's k-loop in shader cycles per k-tile, measured INSIDE the kernel (s_memtime around the loop), for
the kernel as it is (build 8) and its ablation builds (dvt_tune_set(1, 5) + (1, -300 - build); timing only, results wrong):
6 = no LDS-DMA inside the k-loop, 7 = 6 + ring parity frozen (compile-time fragment-read addresses), 9 = no fragment reads,
3 = the kernel with s_memtime stamps after every barrier.  2048 cycles per k-tile = the matrix pipe's rate (2 waves per SIMD x 64
v_mfma_f32_16x16x32_bf16 x 16 cycles); tools/probes/cu_pipe.hip's replica of the loop runs at 2227."""
import ctypes as C
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
L.dvt_vit_debug_buffer.argtypes = [C.c_void_p]
L.dvt_vit_debug_buffer.restype = C.c_int
shapes = [("qkv 110 views", 110 * 1408, 2304, 768), ("fc2 110 views", 110 * 1408, 768, 3072), ("on chip", 2048, 8192, 768),
          ("on chip", 2048, 8192, 3072)]
builds = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "8,6,7,9").split(",")]
torch.manual_seed(0)
for name, M, n, k in shapes:
    x = torch.randn(M, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    b = torch.randn(n, device=dev)
    y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    tiles = (M // 256) * (n // 256)
    dbg = torch.zeros(tiles * 2 * 24, device=dev, dtype=torch.int32)
    assert L.dvt_vit_debug_buffer(dbg.data_ptr()) == 0
    for build in builds:
        L.dvt_tune_set(1, 5)
        L.dvt_tune_set(1, -300 - build)
        dbg.zero_()
        for _ in range(2):
            assert L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream()) == 0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(4):
            L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream())
        ev1.record()
        torch.cuda.synchronize()
        wall_us = ev0.elapsed_time(ev1) / 4 * 1e3
        st = dbg.cpu().numpy().astype(np.uint32).reshape(tiles, 2, 24)
        cyc = st[:, :, 20].astype(np.float64) / np.maximum(st[:, :, 21], 1)
        tot = st[:, 0, 19].astype(np.float64)  # ticks from kernel entry to the last store retired, per workgroup
        if build == 8:  # per CU: how much of the launch's span (in ticks) is spent INSIDE workgroups, and what clock does the span imply
            ent = st[:, 0, 18].astype(np.int64)
            cu = (st[:, 0, 16].astype(np.int64) & 0xF) << 16 | (st[:, 0, 17].astype(np.int64) & 0xFF00)  # xcc | se / sh / cu bits
            tt = st[:, 0, 19].astype(np.int64)
            util, spans = [], []
            for c in np.unique(cu):  # s_memtime is not synchronised between XCDs: spans per CU
                sel = cu == c
                e = ent[sel]
                rel = ((e - e[0] + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)
                rel = rel - rel.min()
                sp = float((rel + tt[sel]).max())
                spans.append(sp)
                util.append(tt[sel].sum() / sp)
            util, spans = np.array(util), np.array(spans)
            print(f"    {len(util)} CUs; per CU: first entry -> last exit = {np.median(spans):.0f} ticks (median) = {np.median(spans) / wall_us:.0f} MHz "
                  f"against the launch's wall time; inside workgroups {np.median(util) * 100:.1f} % of that span (min {util.min() * 100:.1f} %, max "
                  f"{util.max() * 100:.1f} %), {sel.sum()} workgroups on the last CU", flush=True)
        mhz = tot.sum() / 256.0 / wall_us
        print(f"{name:14s} M={M:6d} N={n:5d} K={k:5d}  build {build}: k-loop {np.median(cyc):7.0f} cycles per k-tile (p10 "
              f"{np.percentile(cyc, 10):.0f}, p90 {np.percentile(cyc, 90):.0f}); whole workgroup {np.median(tot):8.0f} ticks; launch "
              f"{wall_us:7.1f} us -> {mhz:5.0f} MHz if the CUs were never idle", flush=True)
    L.dvt_tune_set(1, 4)
    L.dvt_tune_set(1, -300)
    L.dvt_vit_debug_buffer(None)
    del dbg
