"""Developer tool (product library): the L2 tile-order knobs of the 8p GEMM at the bench's launch size -- W bytes per N-tile group
(dvt_tune_set(1, KiB)) x M panels per block (dvt_tune_set(1, -100 - b), 0 = auto) -- on one 398-view extractor launch."""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
from dvt_amd.models import PretrainedViTWrapper  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
x = torch.randn(398, 3, 518, 518, device=dev)
out = torch.empty(398, 37, 37, 768, device=dev)


def t_ms():
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(2):
        vit.features_nhwc(x, out=out, max_batch=400)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / 2


vit.features_nhwc(x, out=out, max_batch=400)
cases = [(4800, 0), (2400, 0), (9600, 0), (1200, 0), (4800, 1), (4800, 2), (4800, 8), (4800, 16), (9600, 8), (2400, 2)]
res = {c: [] for c in cases}
for rnd in range(3):
    for kib, mb in cases:
        L.dvt_tune_set(1, kib)
        L.dvt_tune_set(1, -100 - mb)
        res[(kib, mb)].append(t_ms())
L.dvt_tune_set(1, 4800)
L.dvt_tune_set(1, -100)
for (kib, mb), v in res.items():
    print(f"group {kib:5d} KiB, m-block {mb:2d} ({'auto' if mb == 0 else 'fixed'}): {sorted(v)[1]:7.2f} ms (min {min(v):7.2f})")
