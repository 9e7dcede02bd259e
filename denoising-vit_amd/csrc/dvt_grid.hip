// 2-D multi-resolution hash-grid encoding (tcnn "HashGrid", linear interpolation,
// CoherentPrime hash) for gfx950.  Replaces the tinycudann FFI used at
// dvt/models/neural_feature_field.py:25-39 (ctor) and :48 (call).
//
// Semantics restated from tiny-cuda-nn include/tiny-cuda-nn/encodings/grid.h and
// common_device.h (third party, NOT in the reference tree):
//   scale_l = exp2f(l * log2f(per_level_scale)) * base - 1           (grid_scale)
//   res_l   = (uint32)ceilf(scale_l) + 1                              (grid_resolution)
//   n_l     = min(next_multiple(res_l^2, 8), 2^log2_hashmap_size)     (offset table)
//   pos     = fmaf(scale_l, x, 0.5f); cell = floorf(pos); w = pos - cell   (pos_fract)
//   index   = res_l^2 <= n_l ? cx + cy*res_l : cx ^ (cy * 2654435761u);  index %= n_l
//   enc[l*F+f] = sum_{4 corners} w_c * params[(offset_l + index_c)*F + f]
// There is no clamping: at x == 1 the upper corner equals res_l and wraps through the
// stride/modulo arithmetic exactly as in tcnn.
//
// Memory behaviour: one lane per (sample, level[, feature]); a grid entry is F=8 floats =
// 32 B, read as two float4.  Forward touches 2048*16*4*32 B = 4.2 MB per step, all of it
// random 32-B sectors -- L2/MALL resident for the coarse levels, HBM for levels >= 11.
#include <math.h>

#include "dvt_common.h"

extern "C" int dvt_abi_version(void) { return 1; }

extern "C" int dvt_struct_sizes(int64_t* out) {
  if (!out) return DVT_E_BADARG;
  out[0] = sizeof(DvtGridTable);
  out[1] = sizeof(DvtAdamSeg);
  out[2] = sizeof(DvtAdamArgs);
  out[3] = sizeof(DvtFitConfig);
  out[4] = sizeof(DvtFitBuffers);
  return 0;
}

extern "C" int dvt_grid_table(int n_levels, int n_features, int base_resolution,
                              int max_resolution, int log2_hashmap_size, DvtGridTable* out) {
  if (!out || n_levels < 1 || n_levels > DVT_MAX_LEVELS || n_features != 8 ||
      base_resolution < 1 || max_resolution < base_resolution || log2_hashmap_size < 3 ||
      log2_hashmap_size > 30)
    return DVT_E_BADARG;
  // neural_feature_field.py:34-36 evaluates this in float64 (numpy); tcnn stores it as a
  // json float -> fp32, then takes std::log2 of the fp32 value.
  double pls64 = n_levels > 1 ? exp((log((double)max_resolution) - log((double)base_resolution)) /
                                    (double)(n_levels - 1))
                              : 1.0;
  float pls = (float)pls64;
  float log2_pls = log2f(pls);
  out->n_levels = n_levels;
  out->n_features = n_features;
  out->pad_ = 0;
  uint64_t offset = 0;
  for (int l = 0; l < DVT_MAX_LEVELS; ++l) {
    out->scale[l] = 0.f;
    out->resolution[l] = out->entries[l] = out->offset[l] = out->hashed[l] = 0;
  }
  for (int l = 0; l < n_levels; ++l) {
    float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;
    uint32_t res = (uint32_t)ceilf(scale) + 1u;
    uint64_t dense = (uint64_t)res * (uint64_t)res;
    uint64_t n = (dense + 7u) / 8u * 8u;
    uint64_t cap = 1ull << log2_hashmap_size;
    if (n > cap) n = cap;
    out->scale[l] = scale;
    out->resolution[l] = res;
    out->entries[l] = (uint32_t)n;
    out->offset[l] = (uint32_t)offset;
    out->hashed[l] = dense > n ? 1u : 0u;
    offset += n;
    if (offset > 0xffffffffull) return DVT_E_BADARG;
  }
  out->n_entries_total = (uint32_t)offset;
  return 0;
}

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

// One lane per (sample, level); level is the fast index so the 16 lanes of a sample write
// one contiguous 512-B enc row.
__global__ __launch_bounds__(256) void grid_fwd_kernel(DvtGridTable T, const float2* __restrict__ xy,
                                                       const int32_t* __restrict__ ridx,
                                                       const float4* __restrict__ params,
                                                       float4* __restrict__ enc, int n) {
  const int L = T.n_levels;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  const int b = t / L, l = t - b * L;
  const float2 p = xy[ridx != nullptr ? ridx[b] : b];
  uint32_t idx[4];
  float w[4];
  corners2d(T, l, p.x, p.y, idx, w);
  float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 a = params[(size_t)idx[c] * 2], bq = params[(size_t)idx[c] * 2 + 1];
    lo.x = fmaf(w[c], a.x, lo.x);
    lo.y = fmaf(w[c], a.y, lo.y);
    lo.z = fmaf(w[c], a.z, lo.z);
    lo.w = fmaf(w[c], a.w, lo.w);
    hi.x = fmaf(w[c], bq.x, hi.x);
    hi.y = fmaf(w[c], bq.y, hi.y);
    hi.z = fmaf(w[c], bq.z, hi.z);
    hi.w = fmaf(w[c], bq.w, hi.w);
  }
  enc[(size_t)t * 2] = lo;
  enc[(size_t)t * 2 + 1] = hi;
}

// One lane per (sample, level, feature): the 8 lanes of an entry issue one 32-B atomic
// group per corner (same sector), lane f==0 marks the entry in the touched bitmap.
__global__ __launch_bounds__(256) void grid_bwd_kernel(DvtGridTable T, const float2* __restrict__ xy,
                                                       const int32_t* __restrict__ ridx,
                                                       const float* __restrict__ d_enc,
                                                       float* __restrict__ d_params,
                                                       uint32_t* __restrict__ touched, int n) {
  const int L = T.n_levels;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * L * 8) return;
  const int f = t & 7;
  const int bl = t >> 3;
  const int b = bl / L, l = bl - b * L;
  const float2 p = xy[ridx != nullptr ? ridx[b] : b];
  uint32_t idx[4];
  float w[4];
  corners2d(T, l, p.x, p.y, idx, w);
  const float g = d_enc[t];  // [n, L*8] with column l*8+f == linear index t
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    atomic_add_f32(d_params + (size_t)idx[c] * 8 + f, w[c] * g);
    if (touched != nullptr && f == 0) {
      const uint32_t bit = 1u << (idx[c] & 31u);
      uint32_t* wp = touched + (idx[c] >> 5);
      // plain pre-check keeps the coarse levels (few hot words) from serialising on atomics
      if ((__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0u)
        __hip_atomic_fetch_or(wp, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(256) void grid_corners_kernel(DvtGridTable T,
                                                           const float2* __restrict__ xy,
                                                           uint32_t* __restrict__ oidx,
                                                           float* __restrict__ ow, int n) {
  const int L = T.n_levels;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  const int b = t / L, l = t - b * L;
  const float2 p = xy[b];
  uint32_t idx[4];
  float w[4];
  corners2d(T, l, p.x, p.y, idx, w);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    oidx[(size_t)t * 4 + c] = idx[c];
    ow[(size_t)t * 4 + c] = w[c];
  }
}

// Internal forms with a row indirection xy[ridx[b]] (used by the fused fit loop).
int dvt_grid_fwd_idx(const DvtGridTable* tbl, const float* xy, const int32_t* ridx,
                     const float* params, float* enc, int n, hipStream_t stream) {
  if (!tbl || !xy || !params || !enc || n < 0 || tbl->n_features != 8) return DVT_E_BADARG;
  if (n == 0) return 0;
  const long long threads = (long long)n * tbl->n_levels;
  hipLaunchKernelGGL(grid_fwd_kernel, dim3(dvt_cdiv(threads, 256)), dim3(256), 0, stream, *tbl,
                     (const float2*)xy, ridx, (const float4*)params, (float4*)enc, n);
  DVT_CHECK_LAUNCH();
  return 0;
}

int dvt_grid_bwd_idx(const DvtGridTable* tbl, const float* xy, const int32_t* ridx,
                     const float* d_enc, float* d_params, uint32_t* touched, int n,
                     hipStream_t stream) {
  if (!tbl || !xy || !d_enc || !d_params || n < 0 || tbl->n_features != 8) return DVT_E_BADARG;
  if (n == 0) return 0;
  const long long threads = (long long)n * tbl->n_levels * 8;
  hipLaunchKernelGGL(grid_bwd_kernel, dim3(dvt_cdiv(threads, 256)), dim3(256), 0, stream, *tbl,
                     (const float2*)xy, ridx, d_enc, d_params, touched, n);
  DVT_CHECK_LAUNCH();
  return 0;
}

extern "C" int dvt_grid_fwd(const DvtGridTable* tbl, const float* xy, const float* params,
                            float* enc, int n, void* stream) {
  return dvt_grid_fwd_idx(tbl, xy, nullptr, params, enc, n, (hipStream_t)stream);
}

extern "C" int dvt_grid_bwd(const DvtGridTable* tbl, const float* xy, const float* d_enc,
                            float* d_params, uint32_t* touched, int n, void* stream) {
  return dvt_grid_bwd_idx(tbl, xy, nullptr, d_enc, d_params, touched, n, (hipStream_t)stream);
}

extern "C" int dvt_grid_corners(const DvtGridTable* tbl, const float* xy, uint32_t* idx, float* w,
                                int n, void* stream) {
  if (!tbl || !xy || !idx || !w || n < 0) return DVT_E_BADARG;
  if (n == 0) return 0;
  const long long threads = (long long)n * tbl->n_levels;
  hipLaunchKernelGGL(grid_corners_kernel, dim3(dvt_cdiv(threads, 256)), dim3(256), 0,
                     (hipStream_t)stream, *tbl, (const float2*)xy, idx, w, n);
  DVT_CHECK_LAUNCH();
  return 0;
}
