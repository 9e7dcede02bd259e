"""View synthesis oracle (CPU; PIL + torch) -- TEST INFRASTRUCTURE ONLY.

Restates the image side of the reference's stage-1 input pipeline:

  base transform   main_img_denoising.py:279-286 `Compose([ToPILImage(), Resize(input_size),
                   ToTensor(), normalizer])`: torchvision's `Resize` on a PIL image is
                   `img.resize((w, h), BILINEAR)` (torchvision default interpolation; PIL's
                   resize is support-scaled = anti-aliased when shrinking) on uint8, then /255.
  set_image        dvt/dataset/single_image_dataset.py:29-38: a second `F.resize(..., BICUBIC,
                   antialias=True)` to the SAME size -- torchvision returns the input unchanged
                   when the size already matches, so it is a no-op.
  random view      dvt/dataset/transform.py:39-76: `get_params` (torchvision
                   RandomResizedCrop, third party, restated from its published algorithm),
                   `F.resized_crop(img, i, j, h, w, size, BICUBIC, antialias=True)` on a float
                   tensor = crop + `interpolate(mode="bicubic", antialias=True,
                   align_corners=False)`, optional hflip, and the per-patch coordinate lattice
                   (:55-73, crop EDGES, (x, y) order, x mirrored on flip).
torchvision is absent in this environment: PARITY UNPINNED against torchvision itself; PIL and
torch.nn.functional.interpolate (what torchvision calls) are used directly.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def base_transform(img_u8: np.ndarray, size, mean, std) -> tuple[np.ndarray, torch.Tensor]:
    """uint8 [H, W, 3] -> (resized uint8 [h, w, 3], normalised fp32 [3, h, w])."""
    from PIL import Image
    pil = Image.fromarray(np.ascontiguousarray(img_u8, dtype=np.uint8))
    if pil.size != (size[1], size[0]):
        pil = pil.resize((size[1], size[0]), Image.BILINEAR)
    u8 = np.asarray(pil, dtype=np.uint8)
    x = torch.from_numpy(u8.copy()).permute(2, 0, 1).float().div(255.0)  # ToTensor
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return u8, (x - m) / s


def get_params(height: int, width: int, rng: np.random.RandomState, scale=(0.1, 0.5),
               ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision RandomResizedCrop.get_params: (top, left, h, w)."""
    area = height * width
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(lo, hi))
        w = int(round(math.sqrt(target * aspect)))
        h = int(round(math.sqrt(target / aspect)))
        if 0 < w <= width and 0 < h <= height:
            return int(rng.randint(0, height - h + 1)), int(rng.randint(0, width - w + 1)), h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def view_coords(i, j, h, w, height, width, h_patches, w_patches, flip: bool) -> torch.Tensor:
    """transform.py:55-73."""
    ni, nj, nh, nw = i / float(height), j / float(width), h / float(height), w / float(width)
    gy, gx = torch.meshgrid(torch.linspace(ni, ni + nh, h_patches), torch.linspace(nj, nj + nw, w_patches),
                            indexing="ij")
    c = torch.stack([gx, gy], dim=-1)
    if flip:
        c[:, :, 0] = (c[:, :, 0].max() - c[:, :, 0]) + c[:, :, 0].min()
    return c


def render_view(img: torch.Tensor, box, size) -> torch.Tensor:
    """transform.py:50-52, :70 on a float [3, H, W] tensor; box = (i, j, h, w, flip)."""
    i, j, h, w, flip = (int(v) for v in box)
    out = F.interpolate(img[None, :, i:i + h, j:j + w], size=tuple(size), mode="bicubic", antialias=True,
                        align_corners=False)[0]
    return out.flip(-1) if flip else out


def make_views(img: torch.Tensor, num_views: int, size, h_patches: int, w_patches: int,
               rng: np.random.RandomState):
    """`num_views` random views + the original as the LAST sample (main_img_denoising.py:331-339):
    boxes int64 [V+1, 5], views [V+1, 3, h, w], coords [V+1, hp, wp, 2]."""
    H, W = img.shape[1:]
    boxes, views, coords = [], [], []
    for _ in range(num_views):
        i, j, h, w = get_params(H, W, rng)
        flip = bool(rng.random_sample() < 0.5)
        boxes.append((i, j, h, w, int(flip)))
        views.append(render_view(img, boxes[-1], size))
        coords.append(view_coords(i, j, h, w, H, W, h_patches, w_patches, flip))
    boxes.append((0, 0, H, W, 0))
    views.append(img.clone())
    gy, gx = torch.meshgrid(torch.linspace(0, 1, h_patches), torch.linspace(0, 1, w_patches), indexing="ij")
    coords.append(torch.stack([gx, gy], dim=-1))  # make_patch_coordinates(hp, wp, 0, 1), :337
    return np.asarray(boxes, np.int64), torch.stack(views), torch.stack(coords)
