"""Developer probe: how well does one fit (300 steps, its own stream) overlap with a loop of extractor kernels of ONE kind?
efficiency = (t_fit_alone + t_kernels_alone) / t_both_concurrent: 1.0 = the two time-slice the machine, 2.0 = free overlap.
Question behind it: the fit slows the extractor's GEMMs by 22 % in the pipelined bench and its attention kernels not at all --
would gating the fit's steps into the attention windows pay?"""
import os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib
import dvt_amd.vit  # noqa
from dvt_amd.fit import FitEngine, FitSettings
L = _lib.lib(); dev = torch.device("cuda:0")
T = 600
n_rows = 769 * 1369
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, 768, device=dev, generator=g); xy = torch.rand(n_rows, 2, device=dev, generator=g)
eng = FitEngine(FitSettings(num_iters=T, warmup_iters=T // 10, mlp_dtype="bfloat16"), n_rows, dev)
idx = np.random.RandomState(0).randint(0, n_rows, (T, 2048)).astype(np.int32)
M = 110 * 1408
x = torch.randn(M, 768, device=dev).bfloat16(); w = (torch.randn(2304, 768, device=dev) / 28).bfloat16()
xh = torch.randn(M, 3072, device=dev).bfloat16(); w2 = (torch.randn(768, 3072, device=dev) / 55).bfloat16()
b = torch.randn(3072, device=dev); y = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
xr = torch.randn(M, 768, device=dev); gm = torch.randn(768, device=dev) * 1e-3
qk = torch.randn(M, 1536, device=dev).bfloat16(); vt = torch.randn(110, 12, 64, 1408, device=dev).bfloat16()
out = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
s_fit, s_x = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

def k_qkv(): L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, 2304, 768, s_x.cuda_stream)
def k_fc2(): L.dvt_vit_gemm_residual(xh.data_ptr(), w2.data_ptr(), b.data_ptr(), gm.data_ptr(), xr.data_ptr(), M, 768, 3072, s_x.cuda_stream)
def k_attn(): L.dvt_vit_attention(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), 110, 12, 1408, 1370, s_x.cuda_stream)

def run_fit():
    torch.cuda.set_device(dev)
    with torch.cuda.stream(s_fit):
        eng.reset(g)
        eng.fit(feat, xy, idx, log_every=0)

def timed(do_fit, kern, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = None
    if do_fit:
        th = threading.Thread(target=run_fit); th.start()
    if kern is not None:
        for _ in range(n): kern()
    if th: th.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0

timed(True, None, 0)
t_fit = min(timed(True, None, 0) for _ in range(2))
print(f"fit alone: {t_fit * 1e3:.1f} ms for {T} steps ({t_fit / T * 1e6:.1f} us per step)", flush=True)
for name, kern in (("qkv GEMM (K = 768)", k_qkv), ("fc2 GEMM (K = 3072, residual epilogue)", k_fc2), ("attention", k_attn)):
    for _ in range(3): kern()
    t1 = timed(False, kern, 20) / 20
    n = max(4, int(round(t_fit / t1)))
    t_x = min(timed(False, kern, n) for _ in range(2))
    t_both = min(timed(True, kern, n) for _ in range(2))
    print(f"{name:40s}: alone {t_x / n * 1e6:7.1f} us per launch x {n}; fit + kernels concurrently {t_both * 1e3:7.1f} ms vs "
          f"{(t_fit + t_x) * 1e3:7.1f} ms back to back -> efficiency {(t_fit + t_x) / t_both:.2f}", flush=True)
