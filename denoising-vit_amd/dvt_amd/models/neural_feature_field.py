"""NeuralFeatureField on MI355X: 2-D multi-resolution hash grid + 2-layer MLP.

Drop-in for the reference class dvt/models/neural_feature_field.py:11-49: same constructor
arguments and defaults, same attribute names (`neural_field` with `.params` /
`.n_output_dims`, `mlp` = Sequential with Linear at indices 0 and 2), same state-dict keys
(`neural_field.params`, `mlp.0.weight`, `mlp.0.bias`, `mlp.2.weight`, `mlp.2.bias`), same
`forward(coords[..., 2]) -> [..., feat_dim]`.  The tinycudann encoding (:25-39, :48) is
replaced by the hand-written HIP kernels in csrc/dvt_grid.hip, the cuBLAS linears (:40-44,
:49) by the f32-MFMA kernels in csrc/dvt_gemm_f32.hip; both are reached through the C ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from .. import _lib


# ------------------------------------------------------------------------------ linear
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Tensor | None, relu: bool):
        _lib.require_cuda(x, w, b)
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        w = w.contiguous().float()
        bb = None if b is None else b.contiguous().float()
        m, k = x2.shape
        n = w.shape[0]
        y = torch.empty((m, n), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().dvt_linear_fwd(_lib.ptr(x2), _lib.ptr(w), _lib.ptr(bb), _lib.ptr(y),
                                             m, n, k, int(relu), _lib.stream()), "dvt_linear_fwd")
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.has_bias = b is not None
        ctx.in_shape = x.shape
        return y.reshape(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, w, y = ctx.saved_tensors
        m, k = x2.shape
        n = w.shape[0]
        dy2 = dy.reshape(m, n).contiguous().float()
        if y is not None:  # ReLU was fused into the forward: mask the incoming gradient
            dy2 = dy2 * (y > 0)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dw = torch.zeros_like(w) if need_w else None
        db = torch.zeros(n, device=w.device, dtype=torch.float32) if (ctx.has_bias and need_w) else None
        dx = torch.empty_like(x2) if need_x else None
        _lib.check(_lib.lib().dvt_linear_bwd(_lib.ptr(dy2), _lib.ptr(x2), _lib.ptr(w), _lib.ptr(dw),
                                             _lib.ptr(db), _lib.ptr(dx), None, m, n, k,
                                             _lib.stream()), "dvt_linear_bwd")
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0)
        return (None if dx is None else dx.reshape(ctx.in_shape)), dw, db, None


class HipLinear(nn.Linear):
    """nn.Linear whose forward/backward run on the f32-input MFMA kernels (exact fp32)."""

    def forward(self, x: Tensor) -> Tensor:  # noqa: D401
        return _LinearFn.apply(x, self.weight, self.bias, False)


# ------------------------------------------------------------------------------ hash grid
class _GridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xy: Tensor, params: Tensor, table):
        _lib.require_cuda(xy, params)
        xy = xy.contiguous().float()
        n = xy.shape[0]
        width = table.n_levels * table.n_features
        enc = torch.empty((n, width), device=xy.device, dtype=torch.float32)
        _lib.check(_lib.lib().dvt_grid_fwd(C.byref(table), _lib.ptr(xy), _lib.ptr(params),
                                           _lib.ptr(enc), n, _lib.stream()), "dvt_grid_fwd")
        ctx.save_for_backward(xy)
        ctx.table = table
        ctx.n_params = params.numel()
        return enc

    @staticmethod
    def backward(ctx, d_enc: Tensor):
        (xy,) = ctx.saved_tensors
        d_params = torch.zeros(ctx.n_params, device=xy.device, dtype=torch.float32)  # dense, as tcnn
        d_enc = d_enc.contiguous().float()
        _lib.check(_lib.lib().dvt_grid_bwd(C.byref(ctx.table), _lib.ptr(xy), _lib.ptr(d_enc),
                                           _lib.ptr(d_params), None, xy.shape[0], _lib.stream()),
                   "dvt_grid_bwd")
        return None, d_params, None


class HashGridEncoding(nn.Module):
    """Stand-in for `tcnn.Encoding(n_input_dims=2, {"otype": "HashGrid", ...})`.

    Attributes mirrored from the tcnn torch binding: `.params` (flat fp32 nn.Parameter,
    level-major, then entry, then feature), `.n_output_dims`, `.n_input_dims`, `.seed`.
    Initialisation is U(-1e-4, 1e-4) like tcnn's; the pcg32 stream of tcnn (seed 1337) is
    not reproduced (third-party, absent) -- a torch Philox stream with the same seed is used.
    """

    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337,
                 dtype: torch.dtype | None = torch.float32):
        super().__init__()
        if n_input_dims != 2:
            raise ValueError("only 2-D grids are on the DVT path")
        if encoding_config.get("otype", "HashGrid") != "HashGrid":
            raise ValueError("only otype=HashGrid is supported")
        if encoding_config.get("interpolation", "linear").lower() != "linear":
            raise ValueError("only linear interpolation is supported")
        if dtype not in (None, torch.float32):
            raise ValueError("grid parameters are fp32 (neural_feature_field.py:27)")
        self.n_input_dims = 2
        self.seed = seed
        self.encoding_config = dict(encoding_config)
        L = int(encoding_config["n_levels"])
        F = int(encoding_config.get("n_features_per_level", 2))
        base = int(encoding_config.get("base_resolution", 16))
        log2_T = int(encoding_config.get("log2_hashmap_size", 19))
        pls = float(encoding_config.get("per_level_scale", 2.0))
        # recover max_resolution from per_level_scale = exp((ln max - ln base)/(L-1))
        max_res = int(round(base * pls ** (L - 1))) if L > 1 else base
        self.table = _lib.grid_table(L, F, base, max_res, log2_T)
        self.n_output_dims = L * F
        n_params = int(self.table.n_entries_total) * F
        gen = torch.Generator().manual_seed(seed)
        init = (torch.rand(n_params, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * 1e-4
        self.params = nn.Parameter(init)

    def forward(self, x: Tensor) -> Tensor:
        return _GridFn.apply(x, self.params, self.table)


class NeuralFeatureField(nn.Module):
    """A neural field that maps 2D coordinates to features (reference :11-49)."""

    def __init__(
        self,
        feat_dim: int = 768,
        base_resolution: int = 16,
        max_resolution: int = 1024,
        n_levels: int = 10,
        n_features_per_level: int = 8,
        log2_hashmap_size: int = 20,
    ):
        super().__init__()
        self.neural_field = HashGridEncoding(
            n_input_dims=2,
            dtype=torch.float32,
            encoding_config={
                "otype": "HashGrid",
                "n_levels": n_levels,
                "n_features_per_level": n_features_per_level,
                "log2_hashmap_size": log2_hashmap_size,
                "base_resolution": base_resolution,
                "per_level_scale": np.exp(
                    (np.log(max_resolution) - np.log(base_resolution)) / (n_levels - 1)
                ),
                "interpolation": "linear",
            },
        )
        # exact table from the integer arguments (avoids the round trip through per_level_scale)
        self.neural_field.table = _lib.grid_table(n_levels, n_features_per_level, base_resolution,
                                                  max_resolution, log2_hashmap_size)
        self.mlp = nn.Sequential(
            HipLinear(self.neural_field.n_output_dims, feat_dim // 2),
            nn.ReLU(),
            HipLinear(feat_dim // 2, feat_dim),
        )

    def forward(self, coords: Tensor) -> Tensor:
        # Same contract as the reference (:47), including its device->host sync: this is the
        # reference-style module API.  (Without the check, out-of-range coordinates would wrap
        # silently through the grid's stride / modulo arithmetic.)  The fused loop validates the
        # whole coordinate table once per image instead (FitEngine.buffers / check_inputs).
        _lib.require_cuda(coords)
        lo, hi = torch.aminmax(coords.detach())
        if not (bool(hi <= 1) and bool(lo >= 0)):  # also catches NaN; an `assert` would vanish under python -O
            raise _lib.DvtError("coordinates should be in [0, 1] (neural_feature_field.py:47)")
        feats = self.neural_field(coords.reshape(-1, 2))
        return self.mlp(feats.view(list(coords.shape[:-1]) + [-1]))
