"""PretrainedViTWrapper on MI355X -- drop-in for dvt/models/vit_wrapper.py:59-146.

Same constructor arguments, properties (`n_output_dims`, `num_blocks`, `last_layer_index`,
`patch_size`, `stride`, `transformation`) and `get_intermediate_layers` signature; the timm
model behind it is replaced by the hand-written HIP forward (csrc/dvt_vit.hip).

Differences forced by the environment (documented in DESIGN.md):
  * no network / no timm: `pretrained=True` cannot download.  Weights come from
    `checkpoint_path=` (a timm-layout state dict saved with torch.save) or the environment
    variable DVT_VIT_CHECKPOINT.  Without either the constructor RAISES, like the reference
    does when the pretrained weights cannot be loaded -- features of a random ViT written to
    disk would be skipped forever by the existence-based resume.  Random init (seed 0) is
    available only on request (`allow_random_init=True`: synthetic benchmarks and tests).
  * only the DINOv2 S/B/L backbones (with / without registers) are built; the other ids of
    the reference's MODEL_LIST raise NotImplementedError (out of scope).
  * the stride override (vit_wrapper.py:78-91) is honoured by the im2col kernel, but a
    grid other than the checkpoint's 37x37 is served by resampling pos_embed on the host (timm's
    resample_abs_pos_embed restated); the *_reg4_* models carry 4 register tokens (prefix tokens
    are stripped from the returned map like timm's `return_prefix_tokens=False`).
"""
from __future__ import annotations

import os
import re
import warnings
from typing import List, Tuple, Union

import torch
import torch.nn as nn

from .. import vit as _vit

MODEL_LIST = [
    # DINOv1
    "vit_small_patch8_224.dino", "vit_small_patch16_224.dino", "vit_base_patch8_224.dino",
    "vit_base_patch16_224.dino",
    # DINOv2
    "vit_small_patch14_dinov2.lvd142m", "vit_base_patch14_dinov2.lvd142m",
    "vit_large_patch14_dinov2.lvd142m", "vit_giant_patch14_dinov2.lvd142m",
    # DINOv2 + register
    "vit_small_patch14_reg4_dinov2.lvd142m", "vit_base_patch14_reg4_dinov2.lvd142m",
    "vit_large_patch14_reg4_dinov2.lvd142m", "vit_giant_patch14_reg4_dinov2.lvd142m",
    # MAE
    "vit_base_patch16_224.mae", "vit_large_patch16_224.mae", "vit_huge_patch14_224.mae",
    # CLIP
    "vit_base_patch16_clip_384.laion2b_ft_in12k_in1k", "vit_base_patch16_clip_224.openai",
    # EVA
    "eva02_base_patch16_clip_224.merged2b",
    # DEiT-III
    "deit3_base_patch16_224.fb_in1k",
    # Auto-auged supervised ViT:
    "vit_base_patch16_384.augreg_in21k_ft_in1k",
]  # entry for entry the reference's list (dvt/models/vit_wrapper.py:15-56; its SAM / I-JEPA ids are commented out)

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


class Normalize:
    """Minimal stand-in for torchvision.transforms.Normalize (torchvision is absent here);
    the driver only reads `.mean` / `.std` (main_img_denoising.py:250-255)."""

    def __init__(self, mean, std):
        self.mean, self.std = tuple(mean), tuple(std)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        m = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        s = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - m) / s


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class _PatchEmbedInfo:
    """Carries the attributes the reference touches on `model.patch_embed`."""

    class _Proj:
        def __init__(self, stride):
            self.stride = [stride, stride]

    def __init__(self, patch, stride):
        self.patch_size = (patch, patch)
        self.proj = self._Proj(stride)


class _ModelView(nn.Module):
    """`wrapper.model` of the reference exposes pos_embed / blocks / patch_embed; keep those."""

    def __init__(self, pos_embed, depth, patch, stride):
        super().__init__()
        self.register_buffer("pos_embed", pos_embed, persistent=False)
        self.blocks = nn.ModuleList([nn.Identity() for _ in range(depth)])
        self.patch_embed = _PatchEmbedInfo(patch, stride)


class PretrainedViTWrapper(nn.Module):
    def __init__(
        self,
        model_identifier: str = "vit_base_patch14_dinov2.lvd142m",
        stride: int = 7,
        dynamic_img_size: bool = True,
        dynamic_img_pad: bool = False,
        checkpoint_path: str | None = None,
        img_size: int | Tuple[int, int] | None = None,
        allow_random_init: bool = False,
        dtype: str = "bfloat16",
        **kwargs,
    ):
        super().__init__()
        assert model_identifier in MODEL_LIST, f"Model type {model_identifier} not tested yet."
        if model_identifier not in _vit.SPECS:
            raise NotImplementedError(
                f"{model_identifier}: only {sorted(_vit.SPECS)} are built for MI355X (BASELINE.json)")
        self.model_identifier = model_identifier
        self.stride = stride
        self.patch_size = int(re.search(r"patch(\d+)", model_identifier).group(1))
        self.dynamic_img_size = dynamic_img_size
        self.dynamic_img_pad = dynamic_img_pad
        self.spec = _vit.SPECS[model_identifier]
        size = img_size or self.spec.img_size
        self.img_size = (size, size) if isinstance(size, int) else tuple(size)
        self.allow_random_init = bool(allow_random_init)
        # arithmetic of the extractor: "bfloat16" = the reference under autocast, "float32" = its default
        self.dtype = "float32" if str(dtype) in ("float32", "torch.float32", "fp32") else "bfloat16"
        self._state_dict, self.transformation = self.create_model(model_identifier, checkpoint_path)
        # a grid other than the checkpoint's (stride override vit_wrapper.py:78-91, other input
        # sizes) is handled by resampling pos_embed once on the host (dvt_amd.vit.resample_pos_embed)
        self.model = _ModelView(self._state_dict["pos_embed"].clone(), self.spec.depth,
                                self.patch_size, stride)
        self._hip = None

    def create_model(self, model_identifier: str, checkpoint_path: str | None = None):
        path = checkpoint_path or os.environ.get("DVT_VIT_CHECKPOINT")
        n_tokens = (0 if self.spec.n_reg else 1) + (self.spec.img_size // self.spec.patch) ** 2
        if path:
            sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd.get("model", sd))
        elif not self.allow_random_init:
            raise RuntimeError(
                f"{model_identifier}: no pretrained checkpoint (timm cannot download here). Pass "
                "checkpoint_path= / --vit_checkpoint or set DVT_VIT_CHECKPOINT to a timm-layout state "
                "dict; random weights need an explicit allow_random_init=True / --synthetic.")
        else:
            warnings.warn(f"{model_identifier}: RANDOM ViT weights (seed 0) on request -- synthetic "
                          "benchmarks and tests only")
            # O(1) LayerScale / biases / norm affines: with DINOv2's LayerScale init (1e-5) twelve
            # random blocks would be a numerical no-op and neither parity nor power draw would mean much
            sd = _vit.random_state_dict(self.spec.dim, self.spec.depth, self.spec.patch, n_tokens,
                                        seed=0, well_conditioned=True, n_reg=self.spec.n_reg)
        # timm data config of the DINOv2 models: ImageNet mean/std
        return sd, Compose([Normalize(IMAGENET_MEAN, IMAGENET_STD)])

    @property
    def n_output_dims(self) -> int:
        return self.model.pos_embed.shape[-1]

    @property
    def num_blocks(self) -> int:
        return len(self.model.blocks)

    @property
    def last_layer_index(self) -> int:
        return self.num_blocks - 1

    def _engine(self, device, dtype: str | None = None, matmul: str = "highest") -> "_vit.HipViT":
        dtype = dtype or self.dtype
        if not isinstance(self._hip, dict):
            self._hip = {}
        key = (torch.device(device), dtype, matmul)
        if key not in self._hip:
            self._hip[key] = _vit.HipViT(self._state_dict, self.patch_size, self.stride, self.img_size,
                                         device, dtype=dtype, matmul=matmul)
        return self._hip[key]

    def features_nhwc(self, x: torch.Tensor, layer_index: int | None = None,
                      out: torch.Tensor | None = None, max_batch: int = 128,
                      dtype: str | None = None, matmul: str = "highest") -> torch.Tensor:
        """Fast path used by the stage-1 driver: NHWC fp32 patch-token map, optionally written
        straight into a slice of the feature store (no NCHW round trip).  `matmul` "high" (float32 only): linear
        layers through bf16x3 (dvt_amd.vit.HipViT)."""
        idx = self.last_layer_index if layer_index is None else layer_index
        return self._engine(x.device, dtype, matmul).forward_features(x.float(), n_blocks=idx + 1, out=out,
                                                                      max_batch=max_batch)

    def get_intermediate_layers(
        self,
        x: torch.Tensor,
        n: Union[int, List[int], Tuple[int]] = 1,
        reshape: bool = True,
        return_prefix_tokens: bool = False,
        norm: bool = True,
    ) -> List[torch.Tensor]:
        if return_prefix_tokens or not norm:
            raise NotImplementedError("return_prefix_tokens / norm=False are not on the DVT path")
        if isinstance(n, int):
            indices = list(range(self.num_blocks - n, self.num_blocks))
        else:
            indices = [i if i >= 0 else self.num_blocks + i for i in n]
        outs = []
        for i in indices:
            f = self.features_nhwc(x, i)  # [B, gh, gw, C]
            # timm returns NCHW (`output_fmt="NCHW"`); a permuted view keeps the driver's
            # `.permute(0, 2, 3, 1)` (main_img_denoising.py:323) free
            outs.append(f.permute(0, 3, 1, 2) if reshape else f.reshape(f.shape[0], -1, f.shape[-1]))
        return outs

    def forward(self, x: torch.Tensor):
        raise NotImplementedError("classification forward is not on the stage-1 path")
