"""Hash-grid encoding oracle (numpy/torch, CPU).  PARITY UNPINNED (tiny-cuda-nn absent).

Restates NVlabs/tiny-cuda-nn (git master, installed unpinned by the reference's
README.md:63) `include/tiny-cuda-nn/encodings/grid.h` + `common_device.h` for the
configuration the reference uses (dvt/models/neural_feature_field.py:25-39):
2 input dims, "HashGrid", linear interpolation, CoherentPrime hash, fp32 parameters.

    scale_l = exp2f(l * log2f(per_level_scale)) * base_resolution - 1        grid_scale()
    res_l   = (uint32) ceilf(scale_l) + 1                                    grid_resolution()
    n_l     = min(next_multiple(res_l^2, 8), 2^log2_hashmap_size)            offset table
    pos = fmaf(scale_l, x, 0.5f); cell = floorf(pos); w = pos - cell         pos_fract()
    corner c (bit d set -> +1 along dim d, weight w_d; else weight 1 - w_d)
    index = (stride walk) cx + cy * res_l      if res_l^2 <= n_l
            cx ^ (cy * 2654435761)             otherwise (coherent_prime_hash)
    index %= n_l ; no clamping of cx, cy                                      grid_index()
    enc[l*F + f] = sum_c w_c * params[(offset_l + index_c) * F + f]          kernel_grid
Call site anchoring parity: neural_feature_field.py:48 `self.neural_field(coords.view(-1, 2))`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

PRIME_Y = np.uint32(2654435761)


@dataclass
class GridTable:
    n_levels: int
    n_features: int
    scale: np.ndarray       # float32 [L]
    resolution: np.ndarray  # uint32 [L]
    entries: np.ndarray     # uint32 [L]
    offset: np.ndarray      # uint32 [L]
    hashed: np.ndarray      # bool [L]

    @property
    def n_entries_total(self) -> int:
        return int(self.entries.astype(np.int64).sum())

    @property
    def n_params(self) -> int:
        return self.n_entries_total * self.n_features


def grid_table(n_levels: int, n_features: int = 8, base_resolution: int = 16,
               max_resolution: int = 1024, log2_hashmap_size: int = 20) -> GridTable:
    # neural_feature_field.py:34-36: numpy float64, then a json float -> fp32 inside tcnn
    pls64 = (np.exp((np.log(max_resolution) - np.log(base_resolution)) / (n_levels - 1))
             if n_levels > 1 else 1.0)
    pls = np.float32(pls64)
    # tcnn: std::log2(float) / exp2f on the HOST (glibc, correctly rounded); numpy's SIMD
    # float32 log2/exp2 are a few ulp off, which matters here: levels 5, 10, 15 sit within an
    # ulp of an integer scale.  Emulate the correctly rounded fp32 functions through fp64.
    log2_pls = np.float32(math.log2(float(pls)))
    scale = np.zeros(n_levels, np.float32)
    res = np.zeros(n_levels, np.uint32)
    ent = np.zeros(n_levels, np.uint32)
    off = np.zeros(n_levels, np.uint32)
    hashed = np.zeros(n_levels, bool)
    o = 0
    for l in range(n_levels):
        e = np.float32(2.0 ** float(np.float32(l) * log2_pls))  # exp2f(l * log2_pls)
        s = np.float32(e * np.float32(base_resolution) - np.float32(1.0))
        r = int(np.ceil(s)) + 1
        dense = r * r
        n = min((dense + 7) // 8 * 8, 1 << log2_hashmap_size)
        scale[l], res[l], ent[l], off[l], hashed[l] = s, r, n, o, dense > n
        o += n
    return GridTable(n_levels, n_features, scale, res, ent, off, hashed)


def corners(table: GridTable, xy: np.ndarray):
    """Entry indices (absolute) [N, L, 4] uint32 and weights [N, L, 4] float32."""
    xy = np.asarray(xy, np.float32)
    n = xy.shape[0]
    L = table.n_levels
    idx = np.zeros((n, L, 4), np.uint32)
    w = np.zeros((n, L, 4), np.float32)
    x64, y64 = xy[:, 0].astype(np.float64), xy[:, 1].astype(np.float64)
    for l in range(L):
        s = np.float64(table.scale[l])
        # fmaf: the fp32*fp32 product is exact in fp64; one rounding to fp32 at the end
        px = (s * x64 + 0.5).astype(np.float32)
        py = (s * y64 + 0.5).astype(np.float32)
        fx, fy = np.floor(px), np.floor(py)
        cx, cy = fx.astype(np.int64).astype(np.uint32), fy.astype(np.int64).astype(np.uint32)
        wx, wy = px - fx, py - fy
        res, ne = np.uint32(table.resolution[l]), np.uint32(table.entries[l])
        for c in range(4):
            ux = cx + np.uint32(c & 1)
            uy = cy + np.uint32((c >> 1) & 1)
            with np.errstate(over="ignore"):
                if table.hashed[l]:
                    index = ux ^ (uy * PRIME_Y)
                else:
                    index = ux + uy * res
            idx[:, l, c] = table.offset[l] + index % ne
            a = wx if (c & 1) else np.float32(1.0) - wx
            b = wy if (c & 2) else np.float32(1.0) - wy
            w[:, l, c] = a * b
    return idx, w


def encode(table: GridTable, params: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
    """enc [N, L*F]; differentiable w.r.t. `params` (dense gradient, like tcnn's backward)."""
    idx, w = corners(table, xy.detach().cpu().numpy())
    F = table.n_features
    n, L = idx.shape[0], table.n_levels
    entries = params.view(-1, F)
    ii = torch.from_numpy(idx.astype(np.int64)).reshape(-1)
    ww = torch.from_numpy(w).to(params.dtype).reshape(n, L, 4, 1)
    vals = entries[ii].view(n, L, 4, F)
    return (ww * vals).sum(2).reshape(n, L * F)


class HashGridOracle(torch.nn.Module):
    """Stand-in for tcnn.Encoding with a `.params` flat fp32 parameter (init U(+-1e-4))."""

    def __init__(self, n_levels=16, n_features=8, base_resolution=16, max_resolution=1024,
                 log2_hashmap_size=20, seed=1337):
        super().__init__()
        self.table = grid_table(n_levels, n_features, base_resolution, max_resolution,
                                log2_hashmap_size)
        self.n_output_dims = n_levels * n_features
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter(
            (torch.rand(self.table.n_params, generator=g) * 2 - 1) * 1e-4)

    def forward(self, xy: torch.Tensor) -> torch.Tensor:
        return encode(self.table, self.params, xy)
