"""Developer probe (round 6): does the extractor's attention kernel (VALU-issue bound, 1.01 x algorithmic HBM bytes, hardly any L2 -> LDS
traffic) overlap with its K = 768 GEMMs (bound by the shared L2 -> LDS / memory-side path, profiles/r06/r07a_*) when the two run on
two streams?  efficiency = (t_a alone + t_b alone) / t_both concurrently: 1.0 = they time-slice the machine, 2.0 = free overlap.
If it did, the extractor's two launches per image (396 + 373 views) could walk the layers half a layer apart."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib
import dvt_amd.vit  # noqa
L = _lib.lib(); dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 198
S_PAD = 1376
M = V * S_PAD // 256 * 256
x = torch.randn(M, 768, device=dev).bfloat16(); w = (torch.randn(2304, 768, device=dev) / 28).bfloat16()
w1 = (torch.randn(3072, 768, device=dev) / 28).bfloat16()
xh = torch.randn(M, 3072, device=dev).bfloat16(); w2 = (torch.randn(768, 3072, device=dev) / 55).bfloat16()
b = torch.randn(3072, device=dev); y = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
xr = torch.randn(M, 768, device=dev); gm = torch.randn(768, device=dev) * 1e-3
qk = torch.randn(V * S_PAD + 128, 1536, device=dev).bfloat16(); vt = torch.randn(V, 12, 64, S_PAD, device=dev).bfloat16()
out = torch.empty(V * S_PAD + 128, 768, device=dev, dtype=torch.bfloat16)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def k_qkv(s): L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, 2304, 768, s.cuda_stream)
def k_fc1(s): L.dvt_vit_gemm_bias(x.data_ptr(), w1.data_ptr(), b.data_ptr(), y.data_ptr(), M, 3072, 768, s.cuda_stream)
def k_fc2(s): L.dvt_vit_gemm_residual(xh.data_ptr(), w2.data_ptr(), b.data_ptr(), gm.data_ptr(), xr.data_ptr(), M, 768, 3072, s.cuda_stream)
def k_attn(s): L.dvt_vit_attention_log2q(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), V, 12, S_PAD, 1370, s.cuda_stream)


def timed(ka, na, kb, nb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(max(na, nb)):  # interleaved submission, the streams run free
        if ka is not None and i < na: ka(sa)
        if kb is not None and i < nb: kb(sb)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for name, kern in (("qkv GEMM (K = 768)", k_qkv), ("fc1 GEMM (K = 768, N = 3072)", k_fc1), ("fc2 GEMM (K = 3072, residual)", k_fc2)):
    for _ in range(3): kern(sa); k_attn(sb)
    ta1 = timed(kern, 20, None, 0) / 20
    tb1 = timed(None, 0, k_attn, 20) / 20
    nb = 24
    na = max(4, int(round(nb * tb1 / ta1)))  # equal alone-time on both streams
    ta = min(timed(kern, na, None, 0) for _ in range(2))
    tb = min(timed(None, 0, k_attn, nb) for _ in range(2))
    tboth = min(timed(kern, na, k_attn, nb) for _ in range(2))
    print(f"{name:32s} x {na} ({ta1 * 1e6:7.1f} us each) beside attention x {nb} ({tb1 * 1e6:7.1f} us each), {V} views: "
          f"{tboth * 1e3:7.1f} ms concurrently vs {(ta + tb) * 1e3:7.1f} ms back to back -> efficiency {(ta + tb) / tboth:.3f}", flush=True)
