"""View synthesis for stage 1: random resized crops (+flip) of one image and, for every
crop, the global coordinates of its patch lattice.

Reference: dvt/dataset/transform.py:9-76 (RandomResizedCropFlip, subclass of
torchvision.transforms.RandomResizedCrop -- torchvision is third party and absent here, its
`get_params` is restated from the published algorithm) and
dvt/dataset/single_image_dataset.py:12-51.

  get_params(img, scale, ratio): 10 tries of  area = H*W*U(scale),  log-uniform aspect in
  `ratio`,  w = round(sqrt(area*ar)), h = round(sqrt(area/ar)),  accept if it fits, random
  top-left; otherwise the central crop clamped to the ratio range.
  coords (transform.py:55-66): linspace(i/H, (i+h)/H, h_patches) x linspace(j/W, (j+w)/W,
  w_patches) -- the crop EDGES, (x, y) order; flip mirrors x (transform.py:69-73).
The crops themselves are resized with bicubic + antialias (transform.py:50-52) by the hand-written
HIP resampler `dvt_render_views` (csrc/dvt_views.hip): all 769 views of an image in one launch,
on the device, instead of 768 PIL resizes in 8 DataLoader worker processes.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def get_params(height: int, width: int, scale=(0.1, 0.5), ratio=(3.0 / 4.0, 4.0 / 3.0),
               rng: np.random.RandomState | None = None):
    """(top i, left j, h, w) -- torchvision RandomResizedCrop.get_params restated."""
    rng = rng or np.random
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = int(rng.randint(0, height - h + 1))
            j = int(rng.randint(0, width - w + 1))
            return i, j, h, w
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


def crop_coords(i, j, h, w, height, width, h_patches, w_patches, flip: bool) -> torch.Tensor:
    """[h_patches, w_patches, 2] (x, y) in [0, 1] -- transform.py:55-73."""
    norm_i, norm_j = i / float(height), j / float(width)
    norm_h, norm_w = h / float(height), w / float(width)
    ys = torch.linspace(norm_i, norm_i + norm_h, h_patches)
    xs = torch.linspace(norm_j, norm_j + norm_w, w_patches)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    coords = torch.stack([gx, gy], dim=-1)
    if flip:
        coords[:, :, 0] = (coords[:, :, 0].max() - coords[:, :, 0]) + coords[:, :, 0].min()
    return coords


def make_patch_coordinates(height, width, start=-1.0, end=1.0) -> torch.Tensor:
    """main_img_denoising.py:21-25."""
    py, px = torch.linspace(start, end, height), torch.linspace(start, end, width)
    py, px = torch.meshgrid(py, px, indexing="ij")
    return torch.stack([px, py], dim=-1)


def sample_view_boxes(num_views: int, size, h_patches: int, w_patches: int,
                      rng: np.random.RandomState | None = None, scale=(0.1, 0.5),
                      horizontal_flip: bool = True):
    """Boxes + coordinates of `num_views` random views plus the original image as the LAST
    sample (main_img_denoising.py:337): boxes [V+1, 5] (i, j, h, w, flip), coords [V+1, hp, wp, 2]."""
    rng = rng or np.random
    H, W = size
    boxes, coords = [], []
    for _ in range(num_views):
        i, j, h, w = get_params(H, W, scale, rng=rng)
        flip = bool(horizontal_flip and rng.random_sample() < 0.5)
        boxes.append((i, j, h, w, int(flip)))
        coords.append(crop_coords(i, j, h, w, H, W, h_patches, w_patches, flip))
    boxes.append((0, 0, H, W, 0))
    coords.append(make_patch_coordinates(h_patches, w_patches, 0.0, 1.0))
    return np.asarray(boxes, np.int64), torch.stack(coords)


@torch.no_grad()
def render_views(image: torch.Tensor, boxes: np.ndarray, out: torch.Tensor) -> None:
    """image [3, H, W] (normalised, on the device) -> out [V, 3, OH, OW]: resized crops (bicubic,
    antialias) and flips, one HIP launch; a full-image box with OH x OW == H x W is the identity."""
    from . import _lib
    _lib.require_cuda(image, out)
    image = image.contiguous().float()
    if out.dtype != torch.float32 or not out.is_contiguous() or out.shape[1] != 3:
        raise _lib.DvtError("out must be a contiguous fp32 [V, 3, OH, OW] tensor")
    H, W = image.shape[1:]
    b = np.ascontiguousarray(boxes, dtype=np.int32)
    if (b[:, 0] < 0).any() or (b[:, 1] < 0).any() or (b[:, 0] + b[:, 2] > H).any() or (b[:, 1] + b[:, 3] > W).any():
        raise _lib.DvtError("crop box outside the image")
    dbox = torch.from_numpy(b).to(image.device)
    V_, _, OH, OW = out.shape
    _lib.check(_lib.lib().dvt_render_views(image.data_ptr(), H, W, dbox.data_ptr(), out.data_ptr(),
                                           min(V_, len(b)), OH, OW, _lib.stream()), "dvt_render_views")


def base_resize_u8(img_u8: np.ndarray, size) -> np.ndarray:
    """The reference's base transform up to ToTensor (main_img_denoising.py:279-284:
    `ToPILImage() -> Resize(input_size)`): torchvision's Resize on a PIL image is PIL's own
    `resize((w, h), BILINEAR)` on uint8 (support-scaled when shrinking, rounded back to uint8).
    The later `F.resize(..., BICUBIC, antialias=True)` of single_image_dataset.py:33-38 asks for
    the size the image already has and returns it unchanged.  One image per ~1000 fit steps: this
    stays on the host, with PIL itself, so the pixels are the reference's bit for bit."""
    from PIL import Image
    pil = Image.fromarray(np.ascontiguousarray(img_u8, dtype=np.uint8))
    if pil.size != (size[1], size[0]):
        pil = pil.resize((size[1], size[0]), Image.BILINEAR)
    return np.asarray(pil, dtype=np.uint8)


def normalize_u8(img_u8: np.ndarray, mean, std, device) -> torch.Tensor:
    """ToTensor (uint8 HWC -> fp32 CHW / 255) + Normalize; [3, H, W] fp32 on `device`."""
    x = torch.from_numpy(np.array(img_u8, dtype=np.uint8, copy=True)).to(device).permute(2, 0, 1).float().div(255.0)
    m = torch.tensor(mean, device=device, dtype=torch.float32).view(3, 1, 1)
    s = torch.tensor(std, device=device, dtype=torch.float32).view(3, 1, 1)
    return ((x - m) / s).contiguous()


def load_image(path: str, size, mean, std, device) -> torch.Tensor:
    """single_image_dataset.py:29-38 + the base transform of main_img_denoising.py:279-286:
    PIL decode -> PIL bilinear resize to `size` (uint8) -> [0,1] -> normalise.
    Returns [3, H, W] fp32 on `device`."""
    from PIL import Image

    Image.MAX_IMAGE_PIXELS = None
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    return normalize_u8(base_resize_u8(img, size), mean, std, device)


def synthetic_views(num_views: int, size, h_patches: int, w_patches: int, device, seed: int = 0):
    """Benchmark input: views ~ N(0,1) generated ON THE DEVICE (already
    'normalised'), with real crop-box coordinates.  Returns (views [V+1,3,H,W], coords)."""
    rng = np.random.RandomState(seed)
    _, coords = sample_view_boxes(num_views, size, h_patches, w_patches, rng)
    g = torch.Generator(device=device).manual_seed(seed)
    views = torch.randn((num_views + 1, 3, size[0], size[1]), device=device, generator=g)
    return views, coords.to(device)
