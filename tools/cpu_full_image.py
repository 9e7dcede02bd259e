""""run one complete image once for the record" -- the CPU oracle (oracle/: fp32 ViT-B/14 +
pure-PyTorch hash-grid field + torch.optim.Adam; the reference has no CPU path of its own, tiny-cuda-nn is CUDA-only)
on ONE full BASELINE configs[1] image: 768 synthetic views + the original through the 12-block extractor, then the
1000-step fit (warm-up 100, B = 2048, L = 16 / 2^20).  No extrapolation.  Writes a JSON record.

    python tools/cpu_full_image.py out.json [threads]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.vit import random_state_dict  # noqa: E402  (host-side weight init only)
from oracle import fit as ofit  # noqa: E402
from oracle import vit as ovit  # noqa: E402
from oracle.models import NeuralFeatureFieldOracle, SingleImageDenoiserOracle  # noqa: E402

threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 8)
torch.set_num_threads(threads)
V, T, WARM, B, H, C = 768, 1000, 100, 2048, 37, 768
g = torch.Generator().manual_seed(0)
sd = random_state_dict(768, 12, 14, 1370, seed=0, well_conditioned=True)
coords = torch.rand(V + 1, H, H, 2, generator=g)
coords[-1] = ofit.make_patch_coordinates(H, H, 0, 1)
feats = torch.empty(V + 1, H, H, C)
t0 = time.perf_counter()
with torch.no_grad():
    for i in range(V + 1):
        view = torch.randn(1, 3, 518, 518, generator=g)  # synthetic N(0,1) views, as in bench.py
        feats[i] = ovit.forward_features(sd, view, 14, 14)[0]
        if i % 64 == 0:
            print(f"view {i}: {time.perf_counter() - t0:.0f} s", flush=True)
t_extract = time.perf_counter() - t0
torch.manual_seed(0)
d, f = SingleImageDenoiserOracle(H, H, C, 11), NeuralFeatureFieldOracle(feat_dim=C, n_levels=16)
idx = np.random.RandomState(0).randint(0, (V + 1) * H * H, (T, B))
t0 = time.perf_counter()
logs = ofit.fit_image(d, f, feats, coords, idx, num_iters=T, warmup_iters=WARM, log_every=100)
den = ofit.final_denoised_feats(d, f, feats, coords)
t_fit = time.perf_counter() - t0
rec = {"what": "one COMPLETE BASELINE configs[1] image on the CPU oracle, no extrapolation (SURVEY 8d)", "threads": threads,
       "host_cpus": os.cpu_count(), "views": V + 1, "num_iters": T, "t_extract_s": t_extract, "t_fit_s": t_fit,
       "seconds_per_image": t_extract + t_fit, "images_per_s": 1.0 / (t_extract + t_fit),
       "loss_first": logs[0]["loss"], "loss_last": logs[T - 1]["loss"], "denoised_feats_norm": float(den.norm())}
open(sys.argv[1], "w").write(json.dumps(rec, indent=1))
print(json.dumps(rec))
