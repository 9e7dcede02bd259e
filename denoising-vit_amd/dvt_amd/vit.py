"""Host side of the HIP ViT extractor (csrc/dvt_vit.hip, C ABI in include/dvt_vit.h).

Weights are accepted in the timm `VisionTransformer` state-dict layout the reference loads
(`timm.create_model(model_identifier, pretrained=True, num_classes=0, dynamic_img_size=True)`,
dvt/models/vit_wrapper.py:105-120): patch_embed.proj.{weight,bias}, cls_token, pos_embed,
blocks.N.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,mlp.fc1,mlp.fc2,ls2.gamma}, norm.
There is no network here, so checkpoints come from a local file or are randomly initialised.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass

import torch

from . import _lib

DVT_VIT_MAX_DEPTH = 48


class VitConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "depth", "heads", "mlp_dim", "patch", "stride", "img_h", "img_w", "grid_h",
        "grid_w", "n_tokens", "s_pad", "k_patch", "n_prefix")] + [("ln_eps", C.c_float),
                                                                  ("pos_has_cls", C.c_int32)]


class VitBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1", "norm2_w", "norm2_b",
        "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2", "qkv_wf", "qkv_cs", "qkv_bf", "fc1_wf", "fc1_cs", "fc1_bf")]


class VitWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("patch_w", "patch_b", "cls_token", "pos_embed", "norm_w",
                                          "norm_b")] + [("blocks", VitBlockWeights * DVT_VIT_MAX_DEPTH)]


_P, _I = C.c_void_p, C.c_int
_lib.register_signatures({
    "dvt_vit_config": (_I, [_I, _I, _I, _I, _I, _I, C.POINTER(VitConfig)]),
    "dvt_vit_config_reg": (_I, [_I, _I, _I, _I, _I, _I, _I, C.POINTER(VitConfig)]),
    "dvt_vit_workspace_bytes": (C.c_int64, [C.POINTER(VitConfig), _I]),
    "dvt_vit_struct_sizes": (_I, [C.POINTER(C.c_int64)]),
    "dvt_vit_forward": (_I, [C.POINTER(VitConfig), C.POINTER(VitWeights), _P, _P, _I, _I, _P, _P]),
    "dvt_vit_gemm_bias": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "dvt_vit_gemm_lnfold": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "dvt_vit_layernorm": (_I, [_P, _P, _P, _P, _I, _I, C.c_float, _P]),
    "dvt_vit_attention": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dvt_vit_attention_log2q": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dvt_vit_workspace_bytes_f32": (C.c_int64, [C.POINTER(VitConfig), _I]),
    "dvt_vit_forward_f32": (_I, [C.POINTER(VitConfig), C.POINTER(VitWeights), _P, _P, _I, _I, _P, _P]),
    "dvt_vit_attention_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dvt_vit_debug_buffer": (_I, [_P]),
    "dvt_vit_is_lab_build": (_I, []),
    "dvt_vit_split3": (_I, [_P, _P, C.c_longlong, _I, _I, _I, _P]),
    "dvt_vit_linear_f32x3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dvt_vit_gemm_f32out": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "dvt_vit_workspace_bytes_f32x3": (C.c_int64, [C.POINTER(VitConfig), _I]),
    "dvt_vit_attention_x3_scratch_bytes": (C.c_int64, [_I, _I, _I]),
    "dvt_vit_attention_x3": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dvt_vit_attention_x3_presplit": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dvt_vit_gemm_gelu_x3": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "dvt_vit_gemm_qkv_x3": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dvt_vit_forward_f32x3": (_I, [C.POINTER(VitConfig), C.POINTER(VitWeights), _P, _P, _I, _I, _P, _P]),
})


@dataclass(frozen=True)
class VitSpec:
    dim: int
    depth: int
    patch: int = 14
    img_size: int = 518
    ls_init: float = 1e-5  # DINOv2 LayerScale init
    n_reg: int = 0         # register tokens (the *_reg4_* checkpoints; timm: no_embed_class=True)


# the DINOv2 ViT-S/B/L backbones of the reference's MODEL_LIST with and without register tokens
# (vit_wrapper.py:21-30; BASELINE.json names B and L).  ViT-g uses a SwiGLU MLP: not built.
SPECS = {
    "vit_small_patch14_dinov2.lvd142m": VitSpec(384, 12),
    "vit_base_patch14_dinov2.lvd142m": VitSpec(768, 12),
    "vit_large_patch14_dinov2.lvd142m": VitSpec(1024, 24),
    "vit_small_patch14_reg4_dinov2.lvd142m": VitSpec(384, 12, n_reg=4),
    "vit_base_patch14_reg4_dinov2.lvd142m": VitSpec(768, 12, n_reg=4),
    "vit_large_patch14_reg4_dinov2.lvd142m": VitSpec(1024, 24, n_reg=4),
}


def resample_pos_embed(pos_embed: torch.Tensor, new_grid: tuple[int, int], n_prefix_pos: int) -> torch.Tensor:
    """pos_embed [1, n_prefix_pos + g0*g0, dim] -> [1, n_prefix_pos + gh*gw, dim] for another token
    grid (other `--stride_size` / `--input_size`, vit_wrapper.py:78-91 + timm dynamic_img_size).
    Restates timm 1.0.7 `resample_abs_pos_embed` (layers/pos_embed.py; third party, absent here):
    the patch part is reshaped to its square grid, resized with bicubic + antialias interpolation
    (align_corners=False) in fp32 and flattened again; prefix positions are carried over."""
    import torch.nn.functional as F
    n_old = pos_embed.shape[1] - n_prefix_pos
    g0 = int(math.sqrt(n_old))
    if g0 * g0 != n_old:
        raise _lib.DvtError(f"pos_embed with {n_old} patch positions is not a square grid")
    if (g0, g0) == tuple(new_grid):
        return pos_embed
    prefix, patch = pos_embed[:, :n_prefix_pos], pos_embed[:, n_prefix_pos:]
    dim = pos_embed.shape[-1]
    patch = patch.float().reshape(1, g0, g0, dim).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=tuple(new_grid), mode="bicubic", antialias=True, align_corners=False)
    patch = patch.permute(0, 2, 3, 1).reshape(1, -1, dim).to(pos_embed.dtype)
    return torch.cat([prefix, patch], dim=1)


def vit_config(dim: int, depth: int, patch: int, stride: int, img_h: int, img_w: int,
               n_reg: int = 0, row_pad: int = 128) -> VitConfig:
    """`row_pad`: an image's tokens are padded to a multiple of it.  128 is what dvt_vit_config writes and what the fp32 /
    bf16x3 forwards need; the bf16 forward takes any multiple of 32 (round 6): 1370 tokens -> 1376 rows instead of 1408,
    2.3 % fewer rows through every GEMM and row-local kernel (DVT_VIT_ROW_PAD overrides, for A/B runs)."""
    cfg = VitConfig()
    _lib.check(_lib.lib().dvt_vit_config_reg(dim, depth, patch, stride, img_h, img_w, n_reg, C.byref(cfg)),
               "dvt_vit_config_reg")
    if row_pad != 128:
        if row_pad % 32 or row_pad <= 0:
            raise _lib.DvtError(f"row_pad must be a positive multiple of 32, not {row_pad}")
        cfg.s_pad = -(-cfg.n_tokens // row_pad) * row_pad
    sizes = (C.c_int64 * 3)()
    _lib.lib().dvt_vit_struct_sizes(sizes)
    if list(sizes) != [C.sizeof(VitConfig), C.sizeof(VitBlockWeights), C.sizeof(VitWeights)]:
        raise _lib.DvtError("ViT struct layout mismatch between ctypes and C")
    return cfg


def random_state_dict(dim: int, depth: int, patch: int, n_tokens: int, seed: int = 0,
                      ls_gamma: float | None = 1e-5, well_conditioned: bool = False,
                      n_reg: int = 0) -> dict:
    """Random-init weights in the timm layout (trunc_normal(0.02)-like matrices).  With
    `well_conditioned` biases / LayerScale / norm affine are random O(1) so that parity tests
    exercise every term (LayerScale 1e-5 would hide block errors)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std  # noqa: E731
    sd = {
        "patch_embed.proj.weight": rn(dim, 3, patch, patch, std=0.05 if well_conditioned else 0.02),
        "patch_embed.proj.bias": rn(dim, std=0.1) if well_conditioned else torch.zeros(dim),
        "cls_token": rn(1, 1, dim, std=0.5 if well_conditioned else 1e-6),
        "pos_embed": rn(1, n_tokens, dim, std=0.5 if well_conditioned else 0.02),
        "norm.weight": 1 + rn(dim, std=0.2) if well_conditioned else torch.ones(dim),
        "norm.bias": rn(dim, std=0.2) if well_conditioned else torch.zeros(dim),
    }
    if n_reg:  # timm layout of the reg4 models: reg_token [1, n_reg, dim]; pos_embed has NO cls row
        sd["reg_token"] = rn(1, n_reg, dim, std=0.5 if well_conditioned else 1e-6)
    ws = 1.0 / math.sqrt(dim) if well_conditioned else 0.02
    for i in range(depth):
        p = f"blocks.{i}."
        for nm in ("norm1", "norm2"):
            sd[p + nm + ".weight"] = 1 + rn(dim, std=0.2) if well_conditioned else torch.ones(dim)
            sd[p + nm + ".bias"] = rn(dim, std=0.2) if well_conditioned else torch.zeros(dim)
        sd[p + "attn.qkv.weight"] = rn(3 * dim, dim, std=ws * (2.0 if well_conditioned else 1.0))
        sd[p + "attn.qkv.bias"] = rn(3 * dim, std=0.2) if well_conditioned else torch.zeros(3 * dim)
        sd[p + "attn.proj.weight"] = rn(dim, dim, std=ws)
        sd[p + "attn.proj.bias"] = rn(dim, std=0.2) if well_conditioned else torch.zeros(dim)
        sd[p + "mlp.fc1.weight"] = rn(4 * dim, dim, std=ws)
        sd[p + "mlp.fc1.bias"] = rn(4 * dim, std=0.2) if well_conditioned else torch.zeros(4 * dim)
        sd[p + "mlp.fc2.weight"] = rn(dim, 4 * dim, std=ws * 0.5)
        sd[p + "mlp.fc2.bias"] = rn(dim, std=0.2) if well_conditioned else torch.zeros(dim)
        for nm in ("ls1", "ls2"):
            sd[p + nm + ".gamma"] = (0.5 + torch.rand(dim, generator=g) if well_conditioned
                                     else torch.full((dim,), float(ls_gamma)))
    return sd


_cap_warned: set = set()


def _warn_cap(msg: str) -> None:
    if msg not in _cap_warned:
        _cap_warned.add(msg)
        print(msg, flush=True)


def balanced_launch_views(n_views: int, max_batch: int) -> int:
    """Views per extractor launch, equal launches: the fewest launches of at most `max_batch` views, equally sized (769
    views at 128 -> 7 x 110 instead of 6 x 128 + 1).  DVT_VIT_BALANCE=0: the reference's plain chunks of max_batch."""
    max_batch = max(1, int(max_batch))
    if os.environ.get("DVT_VIT_BALANCE", "2") == "0":
        return max_batch
    n_launch = -(-n_views // max_batch)
    return -(-n_views // n_launch)


_PLAN_CACHE: dict = {}


def plan_launches(n_views: int, max_batch: int, s_pad: int = 1408, dim: int = 768, mlp_dim: int = 3072,
                  fp32: bool = False) -> list[int]:
    """Views per extractor launch, TILE-ROUND aware (round 4).  Every GEMM of a launch runs (M / 256) x (N / 256) tiles of
    256 x 256 on 256 CUs, so its time is ceil(tiles / 256) ROUNDS: 110 views = 605 M panels give the N = 768 GEMMs (proj,
    fc2) 7.09 -> 8 rounds, 11 % of them idle, while 124 views (682 panels) give 23.98 / 31.97 / 7.99 / 7.99 rounds for qkv /
    fc1 / proj / fc2.  The split of `n_views` into launches of at most `max_batch` views is chosen by dynamic programming
    over a cost model in microseconds per tile round (k-loop 1.68 us per 64 k + the epilogue's 5.6 / 10.5 / 20 us: the
    measured 8p figures) plus the per-view kernels (attention, im2col): 769 views at a cap of 128 -> 124 x 5 + 103 + 46, at the
    default cap of 400 -> 398 + 371, modelled 4 % below 7 x 110.  Results do not depend on the split (tests).  DVT_VIT_BALANCE=1: equal launches (round 3),
    0: plain chunks.
    fp32 = True (round 5): the exact-fp32 extractor's GEMMs run 128 x 128 tiles, two workgroups per CU (512 slots per round,
    ~0.126 us per k and round at ~130 TF/s); at the round-4 cap of 32 views the N = 768 GEMMs ran 4.1 -> 5 rounds."""
    max_batch = max(1, int(max_batch))
    mode = os.environ.get("DVT_VIT_BALANCE", "2")
    if mode != "2":
        step = balanced_launch_views(n_views, max_batch)
        return [min(step, n_views - b0) for b0 in range(0, n_views, step)]
    key = (n_views, max_batch, s_pad, dim, mlp_dim, fp32)
    if key in _PLAN_CACHE:
        return list(_PLAN_CACHE[key])
    if fp32:
        tile, slots = 128, 512
        kt = lambda k: 0.126 * k  # noqa: E731
        gemms = [(3 * dim // tile, kt(dim) + 2.0), (mlp_dim // tile, kt(dim) + 2.0), (dim // tile, kt(dim) + 2.0),
                 (dim // tile, kt(mlp_dim) + 2.0)]
        per_view = 50.0 * (s_pad / 1408.0) ** 2 * (dim / 768.0)
    else:
        tile, slots = 256, 256
        kt = lambda k: 1.68 * (k / 64.0)  # noqa: E731
        gemms = [(3 * dim // 256, kt(dim) + 5.6), (mlp_dim // 256, kt(dim) + 10.5), (dim // 256, kt(dim) + 20.0),
                 (dim // 256, kt(mlp_dim) + 10.0)]  # (N tiles, us per tile round): qkv, fc1, proj, fc2
        per_view = 8.8 * (s_pad / 1408.0) ** 2 * (dim / 768.0)  # attention + the row-local kernels

    def cost(v):
        mt = -(-v * s_pad // tile)
        return per_view * v + sum(-(-mt * nt // slots) * w for nt, w in gemms) + 9.0  # + launch boundaries

    costs = [0.0] + [cost(v) for v in range(1, min(n_views, max_batch) + 1)]
    best = [(0.0, ())] + [None] * n_views
    for n in range(1, n_views + 1):
        c_best, p_best = float("inf"), ()
        for v in range(1, min(n, max_batch) + 1):
            c = costs[v] + best[n - v][0]
            if c < c_best:
                c_best, p_best = c, best[n - v][1] + (v,)
        best[n] = (c_best, p_best)
    plan = sorted(best[n_views][1], reverse=True)
    _PLAN_CACHE[key] = tuple(plan)
    return plan


class HipViT:
    """Device-resident weights + the forward launcher."""

    def __init__(self, state_dict: dict, patch: int, stride: int, img_size: tuple[int, int],
                 device: torch.device | str = "cuda", dtype: str = "bfloat16", matmul: str = "highest"):
        """dtype "bfloat16": bf16 operands / fp32 accumulate (the reference's `--dtype bfloat16` autocast mode);
        "float32": fp32 operands everywhere (its default, autocast off) -- 16x less matrix throughput.
        `matmul` (float32 only) is torch.set_float32_matmul_precision's vocabulary: "highest" (default, what the reference
        runs with) = exact-fp32 matrix cores; "high" = every matrix product (linear layers, q.k^T, p.v) on the bf16 pipe over
        split operands ("bfloat16_3x", ~1e-5 relative per product; include/dvt_vit.h) -- an opt-in, never implied by
        `--dtype float32`."""
        if dtype not in ("bfloat16", "float32"):
            raise _lib.DvtError(f"ViT dtype must be bfloat16 or float32, not {dtype!r}")
        if matmul not in ("highest", "high") or (matmul == "high" and dtype != "float32"):
            raise _lib.DvtError(f"matmul must be 'highest' or (with dtype float32) 'high', not {matmul!r}")
        self.dtype = dtype
        self.x3 = matmul == "high"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DvtError("HipViT needs a HIP device; there is no CPU fallback")
        sd = {k: v.detach() for k, v in state_dict.items()}
        dim = sd["pos_embed"].shape[-1]
        depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        n_reg = int(sd["reg_token"].shape[1]) if "reg_token" in sd else 0
        row_pad = 128  # (the bf16x3 forward: its split kernels walk 64-token blocks of whole 128-row images)
        if not self.x3:  # bf16 (round 6) and exact fp32 (round 6, second session): any multiple of 32 -- 1370 tokens -> 1376 rows
            env = os.environ.get("DVT_VIT_ROW_PAD", "")
            row_pad = int(env) if env.isdigit() and int(env) > 0 else 32
        self.cfg = vit_config(dim, depth, patch, stride, img_size[0], img_size[1], n_reg, row_pad=row_pad)
        cfg = self.cfg
        # other strides / input sizes: the checkpoint's position grid is resampled once, on the host
        sd["pos_embed"] = resample_pos_embed(sd["pos_embed"], (cfg.grid_h, cfg.grid_w), int(cfg.pos_has_cls))
        if sd["pos_embed"].shape[1] != cfg.pos_has_cls + cfg.grid_h * cfg.grid_w:
            raise _lib.DvtError(f"pos_embed has {sd['pos_embed'].shape[1]} rows for a "
                                f"{cfg.grid_h}x{cfg.grid_w} grid")
        dev = self.device
        self._keep = []

        def f32(t):
            t = t.to(dev, torch.float32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        def bf16(t):  # the matrix operands: bf16, or fp32 as they are in the fp32 mode, or its [hi | lo | hi] split
            t = t.to(dev, torch.float32).contiguous()
            if self.x3:
                t3 = torch.empty((t.shape[0], 3 * t.shape[1]), device=dev, dtype=torch.bfloat16)
                _lib.check(_lib.lib().dvt_vit_split3(t.data_ptr(), t3.data_ptr(), t.shape[0], t.shape[1], 1, 0,
                                                     _lib.stream()), "dvt_vit_split3")
                torch.cuda.synchronize(dev)  # `t` may be freed on return
                t = t3
            elif dtype != "float32":
                t = t.to(torch.bfloat16).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        w = VitWeights()
        pw = sd["patch_embed.proj.weight"].reshape(dim, -1).float()
        pw_pad = torch.zeros(dim, cfg.k_patch)
        pw_pad[:, : pw.shape[1]] = pw
        w.patch_w, w.patch_b = bf16(pw_pad), f32(sd["patch_embed.proj.bias"])
        prefix = sd["cls_token"].reshape(1, dim)
        if n_reg:
            prefix = torch.cat([prefix.float(), sd["reg_token"].reshape(n_reg, dim).float()], 0)
        w.cls_token, w.pos_embed = f32(prefix), f32(sd["pos_embed"].reshape(-1, dim))
        w.norm_w, w.norm_b = f32(sd["norm.weight"]), f32(sd["norm.bias"])
        for i in range(depth):
            p, b = f"blocks.{i}.", w.blocks[i]
            b.norm1_w, b.norm1_b = f32(sd[p + "norm1.weight"]), f32(sd[p + "norm1.bias"])
            b.qkv_w, b.qkv_b = bf16(sd[p + "attn.qkv.weight"]), f32(sd[p + "attn.qkv.bias"])
            b.proj_w, b.proj_b = bf16(sd[p + "attn.proj.weight"]), f32(sd[p + "attn.proj.bias"])
            b.ls1 = f32(sd.get(p + "ls1.gamma", torch.ones(dim)))
            b.norm2_w, b.norm2_b = f32(sd[p + "norm2.weight"]), f32(sd[p + "norm2.bias"])
            b.fc1_w, b.fc1_b = bf16(sd[p + "mlp.fc1.weight"]), f32(sd[p + "mlp.fc1.bias"])
            b.fc2_w, b.fc2_b = bf16(sd[p + "mlp.fc2.weight"]), f32(sd[p + "mlp.fc2.bias"])
            b.ls2 = f32(sd.get(p + "ls2.gamma", torch.ones(dim)))
            if dtype == "bfloat16":
                # LayerNorm folded into the consuming GEMM (include/dvt_vit.h): W' = bf16(gamma (.) W), its fp32 column
                # sums, and b' = b + W beta -- computed once, in fp32, from the checkpoint tensors
                for norm, lin, dst in (("norm1", "attn.qkv", "qkv"), ("norm2", "mlp.fc1", "fc1")):
                    W = sd[p + lin + ".weight"].float()
                    gamma, beta = sd[p + norm + ".weight"].float(), sd[p + norm + ".bias"].float()
                    Wf = (W * gamma[None, :]).to(torch.bfloat16)
                    setattr(b, dst + "_wf", bf16(Wf))
                    setattr(b, dst + "_cs", f32(Wf.float().sum(1)))
                    setattr(b, dst + "_bf", f32(sd[p + lin + ".bias"].float() + W @ beta))
        self.weights = w
        self._ws = None
        self._ws_batch = 0

    def workspace_bytes(self, batch: int) -> int:
        """Bytes of scratch one launch of `batch` views needs (the library's own arithmetic)."""
        size_fn = (_lib.lib().dvt_vit_workspace_bytes_f32x3 if self.x3 else
                   _lib.lib().dvt_vit_workspace_bytes_f32 if self.dtype == "float32"
                   else _lib.lib().dvt_vit_workspace_bytes)
        return int(size_fn(C.byref(self.cfg), batch))

    def _workspace(self, batch: int) -> torch.Tensor:
        if self._ws is None or self._ws_batch < batch:
            nbytes = self.workspace_bytes(batch)
            self._ws = torch.zeros(nbytes, device=self.device, dtype=torch.uint8)
            self._ws_batch = batch
        return self._ws

    def forward_features(self, img: torch.Tensor, n_blocks: int | None = None,
                         out: torch.Tensor | None = None, max_batch: int = 128) -> torch.Tensor:
        """img [B,3,H,W] fp32 (normalised) -> [B, grid_h, grid_w, dim] fp32 (NHWC), the final-norm'ed
        patch tokens after `n_blocks` blocks.  `out` may be a slice of the feature store."""
        _lib.require_cuda(img)
        cfg = self.cfg
        if img.dtype != torch.float32 or tuple(img.shape[1:]) != (3, cfg.img_h, cfg.img_w):
            raise _lib.DvtError(f"expected fp32 [B,3,{cfg.img_h},{cfg.img_w}], got {tuple(img.shape)} {img.dtype}")
        img = img.contiguous()
        B = img.shape[0]
        n_blocks = cfg.depth if n_blocks is None else n_blocks
        if out is None:
            out = torch.empty((B, cfg.grid_h, cfg.grid_w, cfg.dim), device=self.device, dtype=torch.float32)
        if not out.is_contiguous() or tuple(out.shape) != (B, cfg.grid_h, cfg.grid_w, cfg.dim):
            raise _lib.DvtError("out must be a contiguous [B, grid_h, grid_w, dim] fp32 tensor")
        # launches: see plan_launches (tile-round aware).  Results do not depend on the batching (tests/test_gpu_vit.py).
        plan = self.launch_plan(B, max_batch)
        ws = self._workspace(max(plan))
        L = _lib.lib()
        fwd = L.dvt_vit_forward_f32x3 if self.x3 else L.dvt_vit_forward_f32 if self.dtype == "float32" else L.dvt_vit_forward
        b0 = 0
        for nb in plan:
            _lib.check(fwd(C.byref(cfg), C.byref(self.weights), img[b0:].data_ptr(), out[b0:].data_ptr(), nb,
                           n_blocks, ws.data_ptr(), _lib.stream()), "dvt_vit_forward")
            b0 += nb
        return out

    def _memory_cap(self, max_batch: int) -> int:
        """`max_batch` bounded by what the device can hold NOW (the fp32 extractor's launches grew from 32 to 160
        views = 8 GB of scratch per engine, zero-filled, with no look at the free memory): the largest launch whose workspace
        fits into 80 % of the free bytes (+ the workspace this engine already owns), never below 1 view.  DVT_VIT_MAX_VIEWS
        caps it by hand."""
        env = os.environ.get("DVT_VIT_MAX_VIEWS", "")
        if env.isdigit() and int(env) > 0:
            max_batch = min(max_batch, int(env))
        if self.device.type != "cuda" or max_batch <= self._ws_batch:
            return max(1, max_batch)
        free, _ = torch.cuda.mem_get_info(self.device)
        budget = int(0.8 * free) + (self._ws.numel() if self._ws is not None else 0)
        b = max_batch
        while b > 1 and self.workspace_bytes(b) > budget:
            b = max(1, min(b - 1, b * 3 // 4))
        if b < max_batch:
            _warn_cap(f"dvt_amd.vit: extractor launches capped at {b} views (asked {max_batch}): {self.workspace_bytes(max_batch) / 2**30:.1f} "
                      f"GiB of scratch do not fit into the {free / 2**30:.1f} GiB free on {self.device}")
        return b

    def launch_plan(self, n_views: int, max_batch: int = 128) -> list[int]:
        """Views of each extractor launch for `n_views` views (what forward_features will do)."""
        cfg = self.cfg
        max_batch = self._memory_cap(max_batch if not (self.dtype == "float32" and not self.x3) else min(max_batch, 160))
        if self.dtype == "float32" and self.x3:
            max_batch = min(max_batch, 64)  # bf16x3: 64 views, 4.4 GB of scratch
        elif self.dtype == "float32":
            # exact fp32: 51 MB of scratch per ViT-B view; launches of up to 160 views (8 GB) so that the 128 x 128 tiles of
            # the N = dim GEMMs fill whole rounds of 512 workgroup slots (round 4: 32 views, 4.1 -> 5 rounds, 18 % idle)
            return plan_launches(n_views, min(max_batch, 160), cfg.s_pad, cfg.dim, cfg.mlp_dim, fp32=True)
        return plan_launches(n_views, max_batch, cfg.s_pad, cfg.dim, cfg.mlp_dim)

