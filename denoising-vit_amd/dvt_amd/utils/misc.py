"""Host-side helpers on the stage-1 path (the three functions of the reference's
dvt/utils/misc.py that main_img_denoising.py calls, plus the shard arithmetic of
sample_scripts/stage1.sh)."""
from __future__ import annotations

import math
import os
import random

import numpy as np
import torch


def fix_random_seeds(seed: int = 31) -> None:
    """misc.py:19-23 -- torch, torch.cuda, numpy and python RNGs."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def lr_schedule(iteration: int, lr: float, min_lr: float, warmup_iters: int, num_iters: int) -> float:
    """Linear warm-up then half-cycle cosine (misc.py:306-315), evaluated in python floats."""
    if iteration < warmup_iters:
        return lr * iteration / warmup_iters
    return min_lr + (lr - min_lr) * 0.5 * (
        1.0 + math.cos(math.pi * (iteration - warmup_iters) / (num_iters - warmup_iters)))


def adjust_learning_rate(optimizer, iteration, args):
    """Drop-in for misc.adjust_learning_rate (misc.py:306-322), incl. `lr_scale` groups."""
    lr = lr_schedule(iteration, args.lr, args.min_lr, args.warmup_iters, args.num_iters)
    for param_group in optimizer.param_groups:
        if "lr_scale" in param_group:
            param_group["lr"] = lr * param_group["lr_scale"]
        else:
            param_group["lr"] = lr
    return lr


def output_paths(save_root: str, model: str, data_root: str, filename: str) -> tuple[str, str]:
    """Output layout of main_img_denoising.py:131-139 / misc.py:326-334."""
    raw_dir = f"{save_root}/raw_features/{model}/"
    den_dir = f"{save_root}/denoised_features/{model}/"
    ext = os.path.splitext(filename)[1]
    return (filename.replace(data_root, raw_dir).replace(ext, ".npy"),
            filename.replace(data_root, den_dir).replace(ext, ".npy"))


def check_if_file_exists(args, filename: str) -> bool:
    """misc.py:325-337 -- resume by existence of BOTH output files."""
    raw_p, den_p = output_paths(args.save_root, args.model, args.data_root, filename)
    return os.path.isfile(raw_p) and os.path.isfile(den_p)


def shard_range(start_idx: int, num_imgs: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous static shard of rank r, as sample_scripts/stage1.sh:15-16 launches it:
    `--start_idx $((start_idx + i * n)) --num_imgs n` with n images per GPU.  `num_imgs` is the
    whole job here; the remainder goes to the first ranks."""
    base, rem = divmod(num_imgs, world_size)
    begin = start_idx + rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def atomic_save_npy(path: str, array: np.ndarray) -> None:
    """np.save through a temp file + rename, so that a crash never leaves a truncated file
    that the existence-based resume would accept."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as f:
        np.save(f, array)
    os.replace(tmp, path)
