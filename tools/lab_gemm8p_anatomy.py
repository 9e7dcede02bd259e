"""Developer tool: where a 256 x 256 tile of the product's 8p GEMM spends its WALL time (developer library schedule 12 = the product
kernel + five s_memrealtime stamps per workgroup, 100 MHz: independent of the shader clock): prologue (kernel entry -> first MFMA),
k-loop, epilogue (-> last store issued), store drain (-> retired), and per CU the gap between one workgroup's end and the next
one's entry.  Epilogues: bias (dvt_vit_gemm_bias), folded LayerNorm + GELU (dvt_vit_gemm_lnfold), LayerScale + residual
(dvt_vit_gemm_residual)."""
import ctypes as C
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
L.dvt_vit_debug_buffer.argtypes = [C.c_void_p]
L.dvt_vit_debug_buffer.restype = C.c_int
M = int(sys.argv[1]) if len(sys.argv) > 1 else 398 * 1376 // 256 * 256
ABL = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # timing-only ablation mask of the epilogues (dvt_tune_set(1, -560 - mask)): 1 = no parking writes
assert L.dvt_tune_set(1, -560 - ABL) == 0
PF = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # experiment: L2 prefetch of the next workgroup's first PF k-tiles (dvt_tune_set(1, -570 - n))
assert L.dvt_tune_set(1, -570 - PF) == 0
print(f"L2 prefetch of the next workgroup's first {PF} k-tile(s) late in the epilogue (0 = off)")
if ABL:
    print(f"TIMING ONLY: epilogue ablation mask {ABL} (outputs are garbage)")
cases = [("qkv  bias", "bias", 2304, 768), ("fc1  ln+gelu", "gelu", 3072, 768), ("proj resid", "resid", 768, 768),
         ("fc2  resid", "resid", 768, 3072), ("fc2  bias", "bias", 768, 3072)]
torch.manual_seed(0)
S = _lib.stream
for name, kind, n, k in cases:
    x = torch.randn(M, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    b = torch.randn(n, device=dev)
    tiles = (M // 256) * (n // 256)
    dbg = torch.zeros(tiles * 8, device=dev, dtype=torch.int32)
    if kind == "resid":
        gm, xres = torch.randn(n, device=dev) * 0.1, torch.randn(M, n, device=dev)
        call = lambda: L.dvt_vit_gemm_residual(x.data_ptr(), w.data_ptr(), b.data_ptr(), gm.data_ptr(), xres.data_ptr(), M, n, k, S())
    elif kind == "gelu":
        y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        stats = torch.stack([torch.randn(M, device=dev) * 0.3, torch.rand(M, device=dev) + 0.5], 1).contiguous()
        cs = w.float().sum(1).contiguous()
        call = lambda: L.dvt_vit_gemm_lnfold(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, stats.data_ptr(), cs.data_ptr(), 1, S())
    else:
        y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        call = lambda: L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, S())
    times = {4: [], 12: []}
    assert L.dvt_vit_debug_buffer(dbg.data_ptr()) == 0
    for rnd in range(4):  # interleaved (the clock drifts over consecutive launches: profiles/r06/README.md); round 0 dropped
        for variant in (4, 12):
            L.dvt_tune_set(1, variant)
            assert call() == 0
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(3):
                call()
            ev1.record()
            torch.cuda.synchronize()
            if rnd:
                times[variant].append(ev0.elapsed_time(ev1) / 3 * 1e3)
    times = {k_: float(np.median(v_)) for k_, v_ in times.items()}
    L.dvt_tune_set(1, 4)
    L.dvt_vit_debug_buffer(None)
    d = dbg.cpu().numpy().astype(np.uint32).reshape(tiles, 8)
    t = d[:, :5].astype(np.int64)
    seg = ((t[:, 1:] - t[:, :-1]) & 0xFFFFFFFF) * 0.01  # us
    cu = (d[:, 6].astype(np.int64) << 32) | (d[:, 5].astype(np.int64) & 0x0000FF00 | (d[:, 5].astype(np.int64) >> 13 & 0x7) << 16)  # xcc | (cu_id, sh_id) | se_id
    gaps, per_cu = [], []
    for c in np.unique(cu):
        rows = t[cu == c]
        rows = rows[np.argsort(rows[:, 0])]
        per_cu.append(len(rows))
        if len(rows) > 1:
            gaps.append(((rows[1:, 0] - rows[:-1, 4]) & 0xFFFFFFFF) * 0.01)
    gaps = np.concatenate(gaps) if gaps else np.zeros(1)
    gaps = gaps[gaps < 1000]
    med = np.median(seg, axis=0)
    setup_us, issued_us = np.median(d[:, 7] & 0xFFFF) * 0.01, np.median(d[:, 7] >> 16) * 0.01
    tile_us = times[4] / (tiles / 256.0)
    print(f"{name:13s} M={M} N={n:5d} K={k:5d}: 8p {times[4]:8.1f} us per launch (stamp build {times[12]:8.1f}) = {tile_us:6.2f} us per tile round; "
          f"median per workgroup: prologue {med[0]:5.2f} (tile map + lane setup done at {setup_us:4.2f}, 12 LDS-DMAs issued at {issued_us:4.2f})  k-loop {med[1]:6.2f} ({med[1] / (k // 64):5.3f} per k-tile)  epilogue {med[2]:5.2f}  store drain {med[3]:5.2f}  "
          f"gap to the CU's next workgroup {np.median(gaps):5.2f} (p90 {np.percentile(gaps, 90):5.2f})  sum {med.sum() + np.median(gaps):6.2f} us; "
          f"{len(per_cu)} CUs, {min(per_cu)}-{max(per_cu)} workgroups each", flush=True)
