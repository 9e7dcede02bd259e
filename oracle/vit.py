"""ViT forward oracle (plain PyTorch fp32, CPU).  PARITY UNPINNED against timm itself
(timm==1.0.7 of the reference's requirements.txt:3 is not installed and there are no
weights); cross-checked in tests against the independent `transformers` Dinov2Model.

Restates what `PretrainedViTWrapper.get_intermediate_layers(x, n=[i], reshape=True)`
(dvt/models/vit_wrapper.py:122-143) computes for a DINOv2 `vit_*_patch14_dinov2` model
through timm's `VisionTransformer.forward_intermediates(..., norm=True, output_fmt="NCHW")`:
  patch_embed (Conv2d dim x 3 x p x p, stride s [vit_wrapper.py:78-91 overrides the stride],
  NHWC flatten) -> cat(cls_token, x) + pos_embed -> blocks[0..i]:
      x = x + ls1.gamma * proj(softmax(q k^T / sqrt(64)) v),  q,k,v = split(qkv(norm1(x)))
      x = x + ls2.gamma * fc2(gelu(fc1(norm2(x))))            (LayerNorm eps 1e-6, exact GELU)
  -> norm(x) -> drop the prefix token -> [B, gh, gw, dim]
(the driver permutes NCHW back to this NHWC layout at main_img_denoising.py:323).
Weights: timm state-dict layout.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def forward_features(sd: dict, img: torch.Tensor, patch: int, stride: int,
                     n_blocks: int | None = None, eps: float = 1e-6, stream_out: list | None = None) -> torch.Tensor:
    """`stream_out`: optional list that receives the fp32 residual stream [B, tokens, dim] BEFORE the final LayerNorm
    (test instrumentation: the outlier stress measures its hot / cold channel ratio there)."""
    dim = sd["pos_embed"].shape[-1]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    n_blocks = depth if n_blocks is None else n_blocks
    heads = dim // 64
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=stride)
    B, _, gh, gw = x.shape
    x = x.permute(0, 2, 3, 1).reshape(B, gh * gw, dim)
    n_reg = sd["reg_token"].shape[1] if "reg_token" in sd else 0
    # timm VisionTransformer._pos_embed with dynamic_img_size: the checkpoint's position grid is
    # resampled to (gh, gw); DINOv2: cls is concatenated first and pos_embed covers cls + patches;
    # the reg4 models (no_embed_class=True): pos_embed covers the patches only, then [cls, reg, patches]
    pos = resample_abs_pos_embed(sd["pos_embed"], (gh, gw), num_prefix_tokens=0 if n_reg else 1)
    if n_reg:
        x = torch.cat([sd["cls_token"].expand(B, -1, -1), sd["reg_token"].expand(B, -1, -1), x + pos], dim=1)
    else:
        x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], dim=1) + pos
    for i in range(n_blocks):
        p = f"blocks.{i}."
        h = F.layer_norm(x, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(B, -1, 3, heads, 64).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        a = torch.softmax((q * 64 ** -0.5) @ k.transpose(-2, -1), dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, -1, dim)
        a = F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        x = x + sd.get(p + "ls1.gamma", 1.0) * a
        h = F.layer_norm(x, (dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])),
                     sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = x + sd.get(p + "ls2.gamma", 1.0) * h
    if stream_out is not None:
        stream_out.append(x)
    x = F.layer_norm(x, (dim,), sd["norm.weight"], sd["norm.bias"], eps)
    return x[:, 1 + n_reg:].reshape(B, gh, gw, dim)  # prefix tokens stripped (return_prefix_tokens=False)


def resample_abs_pos_embed(posemb: torch.Tensor, new_size, num_prefix_tokens: int = 1) -> torch.Tensor:
    """timm 1.0.7 layers/pos_embed.py `resample_abs_pos_embed` (third party, absent: restated from its
    published source): square source grid, bicubic + antialias interpolation in fp32."""
    n_new = new_size[0] * new_size[1] + num_prefix_tokens
    if n_new == posemb.shape[1] and new_size[0] == new_size[1]:
        return posemb
    hw = int(math.sqrt(posemb.shape[1] - num_prefix_tokens))
    prefix, grid = posemb[:, :num_prefix_tokens], posemb[:, num_prefix_tokens:]
    dim = posemb.shape[-1]
    grid = grid.float().reshape(1, hw, hw, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=tuple(new_size), mode="bicubic", antialias=True)
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim).to(posemb.dtype)
    return torch.cat([prefix, grid], dim=1)


def to_hf_dinov2(sd: dict, img_size: int, patch: int):
    """Build a transformers.Dinov2Model (Dinov2WithRegistersModel when the state dict carries
    `reg_token`) with the same weights (second opinion).  The HF register model keeps a cls row in
    its position table (added to cls), timm's reg4 checkpoints do not: that row is set to zero."""
    dim = sd["pos_embed"].shape[-1]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    n_reg = sd["reg_token"].shape[1] if "reg_token" in sd else 0
    kw = dict(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=dim // 64,
              mlp_ratio=4, image_size=img_size, patch_size=patch, layer_norm_eps=1e-6,
              hidden_act="gelu", qkv_bias=True, layerscale_value=1.0, attn_implementation="eager")
    hf = {}
    if n_reg:
        from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
        m = Dinov2WithRegistersModel(Dinov2WithRegistersConfig(num_register_tokens=n_reg, **kw)).eval()
        hf["embeddings.register_tokens"] = sd["reg_token"]
        pos = torch.cat([torch.zeros(1, 1, dim), sd["pos_embed"]], dim=1)
    else:
        from transformers import Dinov2Config, Dinov2Model
        m = Dinov2Model(Dinov2Config(**kw)).eval()
        pos = sd["pos_embed"]
    hf["embeddings.cls_token"] = sd["cls_token"]
    hf["embeddings.mask_token"] = torch.zeros(1, dim)
    hf["embeddings.position_embeddings"] = pos
    hf["embeddings.patch_embeddings.projection.weight"] = sd["patch_embed.proj.weight"]
    hf["embeddings.patch_embeddings.projection.bias"] = sd["patch_embed.proj.bias"]
    for i in range(depth):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        wq, wk, wv = sd[p + "attn.qkv.weight"].chunk(3, 0)
        bq, bk, bv = sd[p + "attn.qkv.bias"].chunk(3, 0)
        for nm, w_, b_ in (("query", wq, bq), ("key", wk, bk), ("value", wv, bv)):
            hf[q + f"attention.attention.{nm}.weight"] = w_
            hf[q + f"attention.attention.{nm}.bias"] = b_
        hf[q + "attention.output.dense.weight"] = sd[p + "attn.proj.weight"]
        hf[q + "attention.output.dense.bias"] = sd[p + "attn.proj.bias"]
        hf[q + "norm1.weight"], hf[q + "norm1.bias"] = sd[p + "norm1.weight"], sd[p + "norm1.bias"]
        hf[q + "norm2.weight"], hf[q + "norm2.bias"] = sd[p + "norm2.weight"], sd[p + "norm2.bias"]
        hf[q + "layer_scale1.lambda1"] = sd[p + "ls1.gamma"]
        hf[q + "layer_scale2.lambda1"] = sd[p + "ls2.gamma"]
        hf[q + "mlp.fc1.weight"], hf[q + "mlp.fc1.bias"] = sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]
        hf[q + "mlp.fc2.weight"], hf[q + "mlp.fc2.bias"] = sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]
    hf["layernorm.weight"], hf["layernorm.bias"] = sd["norm.weight"], sd["norm.bias"]
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected, unexpected
    assert all("mask_token" in k for k in missing), missing
    return m
