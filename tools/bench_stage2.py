"""Stage-2 (N3) step benchmark: BASELINE configs[4] shapes -- ViT-B/14 feature maps 37 x 37 x 768, one Block,
batch 32 per GPU (main_denoiser.py:43) -- forward + loss + backward + (all-reduce) + AdamW per step on synthetic
pairs resident in HBM.  One process per GPU (`python -m torch.distributed.run --nproc-per-node N tools/bench_stage2.py`).
Prints one JSON line; not the repository's headline metric (bench.py measures stage 1)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import dist as D  # noqa: E402
from dvt_amd.models import Denoiser  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--blocks", type=int, default=1)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--grid", type=int, default=37)
ap.add_argument("--stages", type=int, default=0, help="LDS stages of the GEMMs (2 / 3; 0 = library default)")
a = ap.parse_args()
rank, world, local = D.env_ranks()
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
D.init(dev, world)
if a.stages:
    from dvt_amd import _lib
    _lib.lib().dvt_tune_set(4, a.stages)
m = Denoiser(a.grid, a.grid, a.dim, None, True, a.blocks, device=dev, seed=0)
if world > 1:
    dist.broadcast(m.engine.params, src=0)
g = torch.Generator(device=dev).manual_seed(rank)
x = torch.randn(a.batch, a.grid, a.grid, a.dim, device=dev, generator=g)
t = torch.randn(a.batch, a.grid, a.grid, a.dim, device=dev, generator=g)


def step():
    loss = m.training_step(x, t)
    if world > 1:
        dist.all_reduce(m.engine.grads)
    m.engine.adamw_step(1e-4, 1e-5, grad_scale=1.0 / world)
    return loss


for _ in range(a.warmup):
    step()


def region():
    for _ in range(a.steps):
        step()
    return a.steps * a.batch


n, elapsed, per_rank = D.timed(region, dev)
C, T, F, H = a.dim, a.grid * a.grid, 4 * a.dim, a.dim // 64
lin = 2.0 * a.batch * T * (3 * C * C + C * C + 2 * C * F)   # forward linear layers
att = 4.0 * a.batch * H * T * T * 64                        # q k^T and P v
flops = 3.0 * (lin * a.blocks) + (att + 1.5 * att * 2) * a.blocks  # backward: 2x linear, 4 attention products
if rank == 0:
    print(json.dumps({
        "metric": "stage-2 denoiser training samples/s", "value": world * n / elapsed, "unit": "samples/s",
        "n_gpus": world, "steps": a.steps, "ms_per_step": 1e3 * elapsed / a.steps, "dtype": "f32",
        "config": {"workload": f"Denoiser({a.grid}x{a.grid}x{a.dim}, {a.blocks} block) fwd+loss+bwd+AdamW",
                   "batch_per_gpu": a.batch, "parallelism": f"dp{world}"},
        "algorithmic_tflop_per_step": flops / 1e12,
        "achieved_tflops_per_gpu": flops / (elapsed / a.steps) / 1e12,
        "loss": float(step()[0])}))
D.finish()
