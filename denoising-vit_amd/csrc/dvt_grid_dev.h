// Device-side hash-grid index / weight arithmetic shared by dvt_grid.hip and the fused fit kernel
// (dvt_fit_fused.hip).  tcnn semantics restated in the header of dvt_grid.hip.
#pragma once
#include "dvt_common.h"

// Corner c: bit0 -> +1 in x (dim 0), bit1 -> +1 in y (dim 1), as tcnn's corner loop.
__device__ __forceinline__ void corners2d(const DvtGridTable& T, int l, float x, float y,
                                          uint32_t idx[4], float w[4]) {
  const float scale = T.scale[l];
  const uint32_t res = T.resolution[l];
  const uint32_t n = T.entries[l];
  const uint32_t off = T.offset[l];
  const bool hashed = T.hashed[l] != 0;
  const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f);
  const float fx = floorf(px), fy = floorf(py);
  const uint32_t cx = (uint32_t)(int)fx, cy = (uint32_t)(int)fy;
  const float wx = px - fx, wy = py - fy;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t ux = cx + (c & 1), uy = cy + ((c >> 1) & 1);
    uint32_t index = hashed ? (ux ^ (uy * 2654435761u)) : (ux + uy * res);
    index %= n;
    idx[c] = off + index;
    const float a = (c & 1) ? wx : 1.0f - wx;
    const float b = (c & 2) ? wy : 1.0f - wy;
    w[c] = a * b;
  }
}

