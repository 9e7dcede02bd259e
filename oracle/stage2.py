"""Stage-2 oracle -- TEST INFRASTRUCTURE ONLY (nothing under denoising-vit_amd/ may import this).

CPU restatement, in plain PyTorch with autograd, of the reference's stage-2 model and training step:

* `Denoiser` (dvt/models/online_denoiser.py:13-104): learnable `pos_embed` [1, h*w, C] (randn * 0.02, :54-57)
  added to the flattened feature map (:86-89), then `num_blocks` timm `Block`s (:24-52) with
  dim=C, num_heads=C//64, mlp_ratio=4, qkv_bias=True, qk_norm=False, init_values=None (no LayerScale),
  LayerNorm(eps=1e-6), nn.GELU (exact erf), Mlp.
* timm 1.0.7 `Block` / `Attention` / `Mlp` (requirements.txt:3) are an ABSENT third party: restated from the
  published module (pre-norm: x + attn(norm1(x)); x + mlp(norm2(x)); attention = softmax((q * d^-0.5) k^T) v,
  fused qkv Linear, proj Linear).  Parity status: UNPINNED against timm itself; pinned against the
  independent `transformers.ViTLayer`, which computes the same block (tests/test_oracle_stage2.py).
* one optimisation step (main_denoiser.py:204-221): lr from `CosineScheduler` (dvt/utils/misc.py:211-241,
  pinned bit-equal through tests/golden/stage2_host.npz), `F.mse_loss + (1 - F.cosine_similarity.mean())`,
  `torch.optim.AdamW(betas=(0.9, 0.999), weight_decay)` -- the reference's own optimizer, used unchanged.
* samplers (dvt/dataset/sampler.py:7-45), pinned through the same golden file.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

BLOCK_TENSORS = ["norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                 "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                 "mlp.fc2.weight", "mlp.fc2.bias"]


class Attention(nn.Module):
    """timm.models.vision_transformer.Attention (qkv_bias=True, qk_norm=False, no dropout)."""

    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    """timm Block as online_denoiser.py:24-34 configures it."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class Denoiser(nn.Module):
    """online_denoiser.py:13-104 with vit=None (main_denoiser.py:127-135 trains it that way)."""

    def __init__(self, noise_map_height=37, noise_map_width=37, feat_dim=768, enable_pe=True, num_blocks=1):
        super().__init__()
        self.denoiser = Block(feat_dim, feat_dim // 64)
        if num_blocks > 1:
            self.denoiser = nn.Sequential(*[Block(feat_dim, feat_dim // 64) for _ in range(num_blocks)])
        self.pos_embed = None
        if enable_pe:
            self.pos_embed = nn.Parameter(torch.randn(1, noise_map_height * noise_map_width, feat_dim) * 0.02)

    def forward(self, x):
        b, h, w, c = x.shape
        x = x.reshape(b, h * w, c)
        if self.pos_embed is not None:
            x = x + self.pos_embed  # resample_abs_pos_embed is the identity at the training resolution
        return self.denoiser(x).reshape(b, h, w, c)


def loss_fn(pred: torch.Tensor, target: torch.Tensor):
    """main_denoiser.py:213-217."""
    l2 = F.mse_loss(pred, target)
    cos = 1 - F.cosine_similarity(pred, target, dim=-1).mean()
    return l2 + cos, l2, cos


class CosineScheduler:
    """dvt/utils/misc.py:211-241."""

    def __init__(self, base_value, final_value, total_iters, warmup_iters=0, start_warmup_value=0, freeze_iters=0):
        self.final_value, self.total_iters = final_value, total_iters
        freeze = np.zeros((freeze_iters))
        warm = np.linspace(start_warmup_value, base_value, warmup_iters)
        iters = np.arange(total_iters - warmup_iters - freeze_iters)
        sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
        self.schedule = np.concatenate((freeze, warm, sched))
        assert len(self.schedule) == self.total_iters

    def __getitem__(self, it):
        return self.final_value if it >= self.total_iters else self.schedule[it]


def infinite_indices(n: int, count: int) -> list:
    """InfiniteSampler (sampler.py:7-16): 0..n-1 forever."""
    return [i % n for i in range(count)]


def distributed_infinite_indices(n: int, num_replicas: int, rank: int, count: int, epoch: int = 0) -> list:
    """DistributedInfiniteSampler (sampler.py:19-45): rank-strided subset, shuffled ONCE with
    default_rng(epoch) -- the rng is consumed only by the shuffle of this rank's subset -- then cycled."""
    rng = np.random.default_rng(epoch)
    subsets = [list(range(n))[i::num_replicas] for i in range(num_replicas)]
    rng.shuffle(subsets[rank])
    own = subsets[rank]
    return [own[i % len(own)] for i in range(count)]


def scaled_lr(blr: float, batch_size: int, world_size: int) -> float:
    """main_denoiser.py:173."""
    return blr * math.sqrt(batch_size * world_size / 256)


def train(model: Denoiser, batches, num_iterations: int, lr_base: float, min_lr: float, weight_decay: float):
    """main_denoiser.py:174-221 on an iterable of (original_feats, denoised_feats); returns per-step
    (loss, l2, cos, lr)."""
    opt = torch.optim.AdamW(model.parameters(), betas=(0.9, 0.999), weight_decay=weight_decay)
    sched = CosineScheduler(lr_base, min_lr, num_iterations, warmup_iters=int(num_iterations * 0.15))
    log = []
    for step, (orig, den) in enumerate(batches):
        if step >= num_iterations:
            break
        lr = float(sched[step])
        for g in opt.param_groups:
            g["lr"] = lr
        loss, l2, cos = loss_fn(model(orig), den)
        opt.zero_grad()
        loss.backward()
        opt.step()
        log.append((float(loss.detach()), float(l2.detach()), float(cos.detach()), lr))
    return log
