"""Developer probe: clock / power while one kernel class loops (is the chip power-limited under the extractor's kernels?)."""
import os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib
import dvt_amd.vit  # noqa
L = _lib.lib(); dev = torch.device("cuda:0"); S = _lib.stream
M = 110 * 1408
x = torch.randn(M, 768, device=dev).bfloat16(); w = (torch.randn(2304, 768, device=dev) / 28).bfloat16()
b = torch.randn(2304, device=dev); y = torch.empty(M, 2304, device=dev, dtype=torch.bfloat16)
qk = torch.randn(M, 1536, device=dev).bfloat16(); vt = torch.randn(110, 12, 64, 1408, device=dev).bfloat16()
out = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
zx = torch.zeros_like(x); zw = torch.zeros_like(w)

def smi():
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in r.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "power", "junction"))]
    return " | ".join(keep)[:400]

def loop(fn, secs, label):
    stop = [False]; n = [0]
    def sampler():
        time.sleep(secs * 0.5); print(label, "mid-run:", smi(), flush=True)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n[0] += 50
    th.join()
    print(f"{label}: {(time.perf_counter() - t0) / n[0] * 1e6:.1f} us per launch", flush=True)

print("idle:", smi(), flush=True)
loop(lambda: L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, 2304, 768, S()), 4.0, "qkv gemm, random operands")
loop(lambda: L.dvt_vit_gemm_bias(zx.data_ptr(), zw.data_ptr(), b.data_ptr(), y.data_ptr(), M, 2304, 768, S()), 4.0, "qkv gemm, ZERO operands")
loop(lambda: L.dvt_vit_attention(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), 110, 12, 1408, 1370, S()), 4.0, "attention v2")
