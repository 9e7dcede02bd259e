"""Lab: extractor time for one image (769 views, ViT-B/14 518^2) under the launch plans of dvt_amd.vit.plan_launches --
DVT_VIT_BALANCE=1 (round 3: 7 equal launches of 110) vs 2 (round 4: tile-round aware, 124 x 5 + 103 + 46) -- and under the
GEMM variants (4 = 8p default, 7 = 4w deferred where it applies).  Interleaved rounds, same process, same weights."""
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
from dvt_amd.models import PretrainedViTWrapper  # noqa: E402
from dvt_amd.vit import plan_launches  # noqa: E402

dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
x = torch.randn(769, 3, 518, 518, device=dev)
out = torch.empty(769, 37, 37, 768, device=dev)
from tools.labenv import use_lab_library  # noqa: E402
L = use_lab_library()  # schedules / timing builds of csrc/lab/: the developer library, not the product one
configs = [("equal launches (round 3)", "1", 128, 4), ("planned launches", "2", 128, 4), ("planned, cap 160", "2", 160, 4),
           ("planned, cap 192", "2", 192, 4), ("planned, cap 256", "2", 256, 4), ("planned, cap 400", "2", 400, 4),
           ("equal, cap 256", "1", 256, 4), ("equal, cap 192", "1", 192, 4)]
if len(sys.argv) > 1:
    configs += [("planned + 4w GEMM (variant 7)", "2", 128, 7), ("planned + 4w fast GELU (variant 9)", "2", 128, 9)]
res = {c[0]: [] for c in configs}
ref = None
for rnd in range(4):
    for name, mode, cap, var in configs:
        os.environ["DVT_VIT_BALANCE"] = mode
        assert L.dvt_tune_set(1, var) == 0
        vit.features_nhwc(x[:cap], out=out[:cap], max_batch=cap)  # workspace / warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vit.features_nhwc(x, out=out, max_batch=cap)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) * 1e3)
        if rnd == 0:
            if ref is None:
                ref = out.clone()
            else:
                same = bool(torch.equal(ref, out))
                err = float((ref - out).abs().max() / ref.abs().max())
                print(f"{name}: identical to the first config: {same} (max rel diff {err:.2e}); plan {plan_launches(769, cap)}", flush=True)
L.dvt_tune_set(1, 4)
for name, v in res.items():
    v = sorted(v[1:])
    print(f"extract 769 views, {name:36s}: median {v[len(v) // 2]:7.1f} ms  min {v[0]:7.1f} ms", flush=True)
