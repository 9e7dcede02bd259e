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


// ---- backward ------------------------------------------------------------------------------------------
// (design notes: dvt_grid.hip)
struct GridBwdPlan {
  int n_lds_blocks;
  int first_direct_level;  // levels [first_direct_level, L) use global atomics
  int chunk_start[DVT_MAX_LEVELS + 1];  // first LDS block of each LDS level (prefix sum)
  int splits[DVT_MAX_LEVELS];           // sample slices per chunk of that level
};

struct GridBwdPtrs {  // per fit of a batched launch
  const float2* xy[DVT_FIT_BATCH_MAX];
  const int32_t* ridx[DVT_FIT_BATCH_MAX];
  const float* d_enc[DVT_FIT_BATCH_MAX];
  float* d_params[DVT_FIT_BATCH_MAX];
  uint32_t* touched[DVT_FIT_BATCH_MAX];
  uint32_t bitmap_end;  // entries >= this get no `touched` bit (the lazy Adam kernels own them); 0 = no limit
};

// Body of the grid backward for workgroup `bx` of fit `fy` (1024 threads); `acc` / `flags` are the caller's
// LDS ([LDS_CHUNK * 8] floats, [LDS_CHUNK / 32 + 1] words).  Shared by grid_bwd_kernel (dvt_grid.hip) and the
// merged backward launch of the fused fit step (dvt_fit_fused.hip).
template <int LDS_CHUNK>
__device__ __forceinline__ void grid_bwd_body(const DvtGridTable& T, const GridBwdPlan& plan, const GridBwdPtrs& q,
                                              int n, int bx, int fy, float* acc, uint32_t* flags) {
  const float2* __restrict__ xy = q.xy[fy];
  const int32_t* __restrict__ ridx = q.ridx[fy];
  const float* __restrict__ d_enc = q.d_enc[fy];
  float* __restrict__ d_params = q.d_params[fy];
  uint32_t* __restrict__ touched = q.touched[fy];
  const int L = T.n_levels;
  const int tid = threadIdx.x;
  if (bx < plan.n_lds_blocks) {
    int l = 0;
    while (l + 1 < plan.first_direct_level && bx >= plan.chunk_start[l + 1]) ++l;
    const int nsplit = plan.splits[l];
    const int local = bx - plan.chunk_start[l];
    const int split = local % nsplit;
    const uint32_t e0 = (uint32_t)(local / nsplit) * LDS_CHUNK;
    const int per = (n + nsplit - 1) / nsplit;  // samples of this slice
    const int b_begin = split * per, b_end = min(n, b_begin + per);
    const uint32_t abs0 = T.offset[l] + e0;  // first absolute entry
    const uint32_t cnt = min((uint32_t)LDS_CHUNK, T.entries[l] - e0);
    const uint32_t base32 = abs0 & ~31u;  // flags are kept in GLOBAL bitmap word alignment
    for (int i = tid; i < LDS_CHUNK * 8; i += 1024) acc[i] = 0.f;
    if (tid < LDS_CHUNK / 32 + 1) flags[tid] = 0u;
    __syncthreads();
    for (int b = b_begin + tid; b < b_end; b += 1024) {
      const float2 p = xy[ridx != nullptr ? ridx[b] : b];
      uint32_t idx[4];
      float w[4];
      corners2d(T, l, p.x, p.y, idx, w);
      bool any = false;
#pragma unroll
      for (int c = 0; c < 4; ++c) any |= (idx[c] - abs0) < cnt;
      if (!any) continue;
      const float4* gp = reinterpret_cast<const float4*>(d_enc + ((size_t)b * L + l) * 8);
      const float4 g0 = gp[0], g1 = gp[1];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t rel = idx[c] - abs0;
        if (rel < cnt) {
          float* a = acc + rel;  // feature-major: bank = rel % 32, no stride-8 conflicts
          atomicAdd(a + 0 * LDS_CHUNK, w[c] * g0.x);
          atomicAdd(a + 1 * LDS_CHUNK, w[c] * g0.y);
          atomicAdd(a + 2 * LDS_CHUNK, w[c] * g0.z);
          atomicAdd(a + 3 * LDS_CHUNK, w[c] * g0.w);
          atomicAdd(a + 4 * LDS_CHUNK, w[c] * g1.x);
          atomicAdd(a + 5 * LDS_CHUNK, w[c] * g1.y);
          atomicAdd(a + 6 * LDS_CHUNK, w[c] * g1.z);
          atomicAdd(a + 7 * LDS_CHUNK, w[c] * g1.w);
          atomicOr(&flags[(idx[c] - base32) >> 5], 1u << (idx[c] & 31u));
        }
      }
    }
    __syncthreads();
    for (uint32_t e = tid; e < cnt; e += 1024) {
      const uint32_t a = abs0 + e;
      if ((flags[(a - base32) >> 5] >> (a & 31u)) & 1u) {
        float4* dst = reinterpret_cast<float4*>(d_params + (size_t)a * 8);
        if (nsplit > 1) {  // several slices own this entry: combine with global atomics
          float* d = d_params + (size_t)a * 8;
#pragma unroll
          for (int f = 0; f < 8; ++f) atomic_add_f32(d + f, acc[f * LDS_CHUNK + e]);
          continue;
        }
        // single writer per entry within this launch: plain read-modify-write keeps the
        // documented "+=" semantics without atomics
        const float4 o0 = dst[0], o1 = dst[1];
        const float4 s0 = make_float4(acc[e], acc[LDS_CHUNK + e], acc[2 * LDS_CHUNK + e],
                                      acc[3 * LDS_CHUNK + e]);
        const float4 s1 = make_float4(acc[4 * LDS_CHUNK + e], acc[5 * LDS_CHUNK + e],
                                      acc[6 * LDS_CHUNK + e], acc[7 * LDS_CHUNK + e]);
        dst[0] = make_float4(o0.x + s0.x, o0.y + s0.y, o0.z + s0.z, o0.w + s0.w);
        dst[1] = make_float4(o1.x + s1.x, o1.y + s1.y, o1.z + s1.z, o1.w + s1.w);
      }
    }
    if (touched != nullptr && tid < LDS_CHUNK / 32 + 1 && flags[tid] != 0u)
      __hip_atomic_fetch_or(touched + (base32 >> 5) + tid, flags[tid], __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // ---- direct atomics for the fine levels ----
  const int nd = L - plan.first_direct_level;
  const long long t = (long long)(bx - plan.n_lds_blocks) * 1024 + tid;
  if (nd <= 0 || t >= (long long)n * nd * 8) return;
  const int f = (int)(t & 7);
  const long long bl = t >> 3;
  const int b = (int)(bl / nd), l = plan.first_direct_level + (int)(bl - (long long)b * nd);
  const float2 p = xy[ridx != nullptr ? ridx[b] : b];
  uint32_t idx[4];
  float w[4];
  corners2d(T, l, p.x, p.y, idx, w);
  const float g = d_enc[((size_t)b * L + l) * 8 + f];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    atomic_add_f32(d_params + (size_t)idx[c] * 8 + f, w[c] * g);
    if (touched != nullptr && f == 0 && (q.bitmap_end == 0u || idx[c] < q.bitmap_end)) {
      const uint32_t bit = 1u << (idx[c] & 31u);
      uint32_t* wp = touched + (idx[c] >> 5);
      if ((__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0u)
        __hip_atomic_fetch_or(wp, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// ---- sorted-list backward (fused fit step) ----------------------------------------------------------------
// The coordinates and the whole index stream of a fit are resident before its first step, so WHICH grid entries
// step t touches -- and with which interpolation weights -- is known in advance.  grid_sort_kernel (dvt_grid.hip)
// writes, per (step, level), the 4 * batch (sample, corner) pairs sorted by entry index: keys / pay / w
// [step][level][4 * batch].  The backward pass then needs no scatter: thread i takes sorted pair i, forms
// w_i * d_enc[sample_i][level] (8 features), a segmented scan over equal keys inside the wave sums each entry's
// contributions, and the LAST lane of a segment stores the 32-byte gradient -- a plain store when the segment lies
// inside one wave (always, on the fine levels), fp32 atomics only for the few segments a wave boundary cuts.
// Replaces ~1 M memory-side atomics per step (global_atomic_add_f32 / ds_add_f32) by ~60 k 32-byte stores.
struct GridSortedPtrs {
  const uint32_t* keys[DVT_FIT_BATCH_MAX];  // this step: [level][nt] entry index inside the level, ascending
  const uint16_t* pay[DVT_FIT_BATCH_MAX];   //            (sample << 2) | corner
  const float* w[DVT_FIT_BATCH_MAX];        //            bilinear weight of that corner
  int nt;                                   // 4 * batch
  uint32_t bitmap_end;                      // entries >= this are stepped lazily (dvt_adam.hip): no `touched` bit
};

// One block of `bs` threads (1024; 512 in the small-footprint backward) = `bs` consecutive sorted pairs of level `l` of fit
// `fy` (part `part` of nt / bs).  Waves are independent of each other.
__device__ __forceinline__ void grid_gather_body(const DvtGridTable& T, const GridSortedPtrs& q, int fy, int l, int part,
                                                 const float* __restrict__ d_enc, float* __restrict__ d_params,
                                                 uint32_t* __restrict__ touched, int bs = 1024) {
  const int lane = threadIdx.x & 63;
  const int u = part * bs + threadIdx.x;
  const size_t base = (size_t)l * q.nt;
  const uint32_t* __restrict__ keys = q.keys[fy] + base;
  const uint32_t key = keys[u];
  const uint32_t prev = u > 0 ? keys[u - 1] : 0xffffffffu;
  const uint32_t next = u + 1 < q.nt ? keys[u + 1] : 0xffffffffu;
  const uint32_t pay = q.pay[fy][base + u];
  const float wgt = q.w[fy][base + u];
  const float4* gp = reinterpret_cast<const float4*>(d_enc + ((size_t)(pay >> 2) * T.n_levels + l) * 8);
  const float4 g0 = gp[0], g1 = gp[1];
  float c[8] = {wgt * g0.x, wgt * g0.y, wgt * g0.z, wgt * g0.w, wgt * g1.x, wgt * g1.y, wgt * g1.z, wgt * g1.w};
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t ok = __shfl_up(key, d, 64);
    const bool take = lane >= d && ok == key;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const float o = __shfl_up(c[f], d, 64);
      if (take) c[f] += o;
    }
  }
  // all 64 lanes are still active here (nt % 1024 == 0): lane 0's key and predecessor tell whether the wave's first
  // segment began in the previous wave
  const uint32_t first_key = __builtin_amdgcn_readfirstlane(key), first_prev = __builtin_amdgcn_readfirstlane(prev);
  const bool ends = key != next;
  if (!(ends || lane == 63)) return;
  const bool began_before = key == first_key && first_prev == first_key;
  const uint32_t a = T.offset[l] + key;
  float* dst = d_params + (size_t)a * 8;
  if (ends && !began_before) {
    reinterpret_cast<float4*>(dst)[0] = make_float4(c[0], c[1], c[2], c[3]);
    reinterpret_cast<float4*>(dst)[1] = make_float4(c[4], c[5], c[6], c[7]);
  } else {
#pragma unroll
    for (int f = 0; f < 8; ++f) atomic_add_f32(dst + f, c[f]);
  }
  if (ends && touched != nullptr && a < q.bitmap_end)
    __hip_atomic_fetch_or(touched + (a >> 5), 1u << (a & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

bool dvt_grid_sorted_ok(const DvtGridTable* T, int n);
// lists of `steps` consecutive steps of k fits: ridx[f] points at the first of those steps' index rows ([steps][n])
int dvt_grid_sort_k(const DvtGridTable* T, int k, const float* const* xy, const int32_t* const* ridx, int n, int steps,
                    uint32_t* const* keys, uint16_t* const* pay, float* const* w, hipStream_t s,
                    uint32_t* const* ukeys = nullptr, int32_t* const* ucount = nullptr);
// the gather-style grid backward as its own launch (the fp32-operand fit step; the fused step runs the same body inside
// fit_backward_kernel): d_params[entry] = sum over this step's sorted (sample, corner) pairs; keys / pay / w = this step's lists
int dvt_grid_gather_k(const DvtGridTable* T, int k, const uint32_t* const* keys, const uint16_t* const* pay,
                      const float* const* w, int n, uint32_t bitmap_end, const float* const* d_enc, float* const* d_params,
                      uint32_t* const* touched, hipStream_t s);

void dvt_grid_bwd_plan(const DvtGridTable& T, int n, GridBwdPlan* plan);
extern int g_grid_lds_chunk;
