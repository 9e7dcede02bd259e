"""Developer tool: one fp32 extractor pass (`--dtype float32`) over N views, for rocprofv3 --kernel-trace --stats."""
import os, sys, time, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.models import PretrainedViTWrapper  # noqa: E402
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
x = torch.randn(N, 3, 518, 518, device=dev)
out = torch.empty(N, 37, 37, 768, device=dev)
for mm in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["highest"]):
    vit.features_nhwc(x, out=out, dtype="float32", matmul=mm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vit.features_nhwc(x, out=out, dtype="float32", matmul=mm)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"fp32 extractor (matmul {mm}): {N} views in {t * 1e3:.1f} ms = {t / N * 769:.3f} s per 769 views", flush=True)
