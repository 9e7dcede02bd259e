"""PMC workload of round 6 (SQ_BUSY_CYCLES + SQ_VALU_MFMA_BUSY_CYCLES before / after): 385-view launches of the ViT-B
GEMM shapes on uniform random operands with the DEVELOPER library, four interleaved rounds per shape over the schedules 13 = round 5's walk of the 8-phase
ring (kernel gemm_bf16_kernel_8p_lab<EPI, 0>), 4 = the product's walk since round 6 (gemm_bf16_kernel_8p<EPI>), 11 = the same as a
persistent workgroup (gemm_bf16_kernel_8t<EPI>).  Run under
    rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python tools/pmc_target_walks.py
and reduce with tools/pmc_stats.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
from tools.labenv import use_lab_library  # noqa: E402

L = use_lab_library()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M = 385 * 1408
for n, k in ((2304, 768), (3072, 768), (768, 3072)):
    x = (torch.rand(M, k, device=dev, generator=g) * 2 - 1).bfloat16()
    w = ((torch.rand(n, k, device=dev, generator=g) * 2 - 1) / k ** 0.5).bfloat16()
    b = torch.randn(n, device=dev, generator=g)
    y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    for rnd in range(4):  # schedules INTERLEAVED (the clock drifts over consecutive launches); tools/pmc_walk_stats.py drops round 0
        for sched in (13, 4, 11):
            assert L.dvt_tune_set(1, sched) == 0
            assert L.dvt_vit_gemm_bias(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, n, k, _lib.stream()) == 0
            torch.cuda.synchronize()
    del x, w, b, y
L.dvt_tune_set(1, 4)
print("done")
