"""PMC workload for the clock question of VERDICT r4 #1 / weak #7: the qkv GEMM (110 views: M = 154 880, N = 2304, K = 768) launched
20 times on random operands, then 20 times on ZERO operands, then a 385-view launch of each ViT-B GEMM shape once.  Run under
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- python tools/pmc_clock_target.py
and reduce with tools/pmc_clock_stats.py: effective shader clock of a dispatch = GRBM_GUI_ACTIVE / its duration (the guide's
method, MI355X_MICROARCH.md "DVFS give-back").  Profiled dispatches are serialised; the clock is read inside the same run."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
import dvt_amd.vit  # noqa: E402,F401

L, S = _lib.lib(), _lib.stream
dev = torch.device("cuda:0")
M = 110 * 1408
g = torch.Generator(device=dev).manual_seed(0)


def ops(m, n, k, zero=False):
    if zero:
        return (torch.zeros(m, k, device=dev, dtype=torch.bfloat16), torch.zeros(n, k, device=dev, dtype=torch.bfloat16),
                torch.zeros(n, device=dev), torch.empty(m, n, device=dev, dtype=torch.bfloat16))
    return ((torch.rand(m, k, device=dev, generator=g) * 2 - 1).bfloat16(), ((torch.rand(n, k, device=dev, generator=g) * 2 - 1) / k ** 0.5).bfloat16(),
            torch.randn(n, device=dev, generator=g), torch.empty(m, n, device=dev, dtype=torch.bfloat16))


for zero in (False, True):
    x, w, b, y = ops(M, 2304, 768, zero)
    # warm the clocks: the first launches after an idle period run at the idle DPM state
    for _ in range(20):
        assert L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, 2304, 768, S()) == 0
    torch.cuda.synchronize()
    time.sleep(0.2)
Mb = 385 * 1408
for n, k in ((2304, 768), (3072, 768), (768, 3072)):
    x, w, b, y = ops(Mb, n, k)
    for _ in range(3):
        assert L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), Mb, n, k, S()) == 0
    torch.cuda.synchronize()
    del x, w, b, y
print("done")
