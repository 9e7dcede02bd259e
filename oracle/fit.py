"""Oracle of the stage-1 inner loop (reference main_img_denoising.py:28-149), CPU PyTorch.

Follows `denoise_an_image` line by line with two injectable pieces so that the HIP path
and the oracle can consume IDENTICAL randomness (SURVEY.md 8c parity protocol):
the initial modules and the [num_iters, B] index stream.  torch.optim.Adam and the LR
schedule are the reference's own (torch is its optimizer; schedule = misc.py:306-322).
On a CPU-only box `GradScaler("cuda")` disables itself, so the x1024 loss scale that reaches
Adam un-unscaled on the GPU reference (quirk Q1, :55/:88) is applied explicitly.
"""
from __future__ import annotations

import math
from itertools import chain

import numpy as np
import torch


def make_patch_coordinates(height, width, start=-1.0, end=1.0):
    """main_img_denoising.py:21-25 -- (x, y) lattice."""
    py, px = torch.linspace(start, end, height), torch.linspace(start, end, width)
    py, px = torch.meshgrid(py, px, indexing="ij")
    return torch.stack([px, py], dim=-1)


def lr_at(step, lr, min_lr, warmup_iters, num_iters):
    """misc.py:306-315."""
    if step < warmup_iters:
        return lr * step / warmup_iters
    return min_lr + (lr - min_lr) * 0.5 * (
        1.0 + math.cos(math.pi * (step - warmup_iters) / (num_iters - warmup_iters)))


def fit_image(denoiser, neural_field, all_raw_features, all_pixel_coords, idx_stream, *,
              num_iters, warmup_iters, lr=0.01, min_lr=0.001, weight_decay=1e-5,
              freeze_shared_artifacts_after=0.5, grad_scale=1024.0, log_every=0,
              autocast_dtype=None):
    """Runs the loop in place on the given modules; returns {step: {loss scalars}}.

    all_raw_features [V, H, W, C], all_pixel_coords [V, H, W, 2]; idx_stream [num_iters, B]
    int (row indices into the flattened [V*H*W] rows, :73).  autocast_dtype=torch.bfloat16 restates
    the reference's `--dtype bfloat16` mode (:78 `with autocast(enabled=dtype != float32)`)."""
    H, W = all_raw_features.shape[1:3]
    optimizer = torch.optim.Adam(  # :48-54
        chain(denoiser.parameters(), neural_field.parameters()),
        lr=lr, eps=1e-15, weight_decay=weight_decay, betas=(0.9, 0.99))
    sa = make_patch_coordinates(H, W)  # :58-62
    num_views = all_raw_features.shape[0]
    batched_sa = sa.unsqueeze(0).repeat(num_views, 1, 1, 1).reshape(-1, 2)
    batched_raw = all_raw_features.reshape(-1, all_raw_features.shape[-1])  # :64-65
    batched_xy = all_pixel_coords.reshape(-1, 2)
    logs = {}
    for step in range(num_iters):  # :67
        if step > int(freeze_shared_artifacts_after * num_iters):  # :70-72
            denoiser.stop_shared_artifacts_grad()
            denoiser.start_residual_predictor()
        ridx = torch.as_tensor(np.asarray(idx_stream[step]), dtype=torch.long)  # :73
        raw, sac, xy = batched_raw[ridx], batched_sa[ridx], batched_xy[ridx]  # :74-76
        cur = lr_at(step, lr, min_lr, warmup_iters, num_iters)  # :77
        for g in optimizer.param_groups:
            g["lr"] = cur
        with torch.autocast("cpu", dtype=autocast_dtype or torch.bfloat16,
                            enabled=autocast_dtype is not None):  # :78
            out = denoiser(raw_vit_outputs=raw, global_pixel_coords=xy, neural_field=neural_field,
                           shared_artifact_coords=sac, return_visualization=False)  # :79-85
        optimizer.zero_grad()  # :87
        (out["loss"] * grad_scale).backward()  # :88 (scale, never unscaled)
        optimizer.step()  # :89
        if log_every and (step % log_every == 0 or step == num_iters - 1):
            logs[step] = {k: float(v.detach()) for k, v in out.items()}
    return logs


@torch.no_grad()
def final_denoised_feats(denoiser, neural_field, all_raw_features, all_pixel_coords):
    """:121-130 -- F on the ORIGINAL image's lattice (the last sample), quirk Q7."""
    out = denoiser(raw_vit_outputs=all_raw_features[-1:], global_pixel_coords=all_pixel_coords[-1:],
                   neural_field=neural_field, return_visualization=True)
    return out["denoised_feats"].float()
