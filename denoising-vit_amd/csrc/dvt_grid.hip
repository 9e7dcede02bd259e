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
#include "dvt_grid_dev.h"

extern "C" int dvt_abi_version(void) { return DVT_ABI_VERSION; }

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

// ---- backward --------------------------------------------------------------------------
// d_params[(entry)*8 + f] += w_c * d_enc[b, l*8 + f] for the 4 corners of every (sample, level).
//
// Device-scope fp32 atomics that collide on a cache line serialise at the memory side
// (measured: 191 us/step for 1 M atomics when levels 0-5 receive 2-32 hits per entry), so
// the levels are split by expected hits per entry:
//  * "LDS levels" (entries <= LDS_LEVEL_MAX, levels 0-9 of the reference config): the entry
//    range of a level is cut into chunks of LDS_CHUNK entries, one workgroup per chunk scans
//    all samples, accumulates the hits that fall into its chunk with LDS atomics
//    (ds_add_f32) and then writes each touched entry ONCE with plain 32-B stores -- no
//    global atomics, one writer per entry;
//  * "direct levels" (fine, < 0.13 hits per entry): one lane per (sample, level, feature),
//    global_atomic_add_f32, the 8 lanes of an entry hit one 32-B sector.
// Both paths mark touched entries in the bitmap consumed by the fused Adam kernel.
constexpr int LDS_CHUNK_BIG = 1024;  // entries per workgroup (32 KB of accumulators)
constexpr int LDS_CHUNK_SMALL = 256;  // 8 KB: co-resides with the ViT extractor's 136-144 KB workgroups
int g_grid_lds_chunk = LDS_CHUNK_SMALL;  // same reason as g_f32_bk in dvt_gemm_f32.hip
int g_grid_lds_level_max = 40960;  // entries; levels above go the direct-atomic way (tunable)
// ds_add_f32 retires only ~1 lane per 2.75 cycles (measured: the single level-0 workgroup, 65 536
// lane-atomics, took 86 us), so the samples of a coarse level are split over several workgroups,
// each with a private LDS accumulator of its chunk; split chunks are flushed with global atomics
// (<= 16 adds per address, into the few-MB, cache-resident coarse region), unsplit ones with
// plain stores.
int g_grid_lds_atomics_per_block = 4096;  // target LDS lane-atomics per workgroup

template <int LDS_CHUNK>
__global__ __launch_bounds__(1024) void grid_bwd_kernel(DvtGridTable T, GridBwdPlan plan,
                                                        GridBwdPtrs q, int n) {
  __shared__ float acc[LDS_CHUNK * 8];
  __shared__ uint32_t flags[LDS_CHUNK / 32 + 1];
  grid_bwd_body<LDS_CHUNK>(T, plan, q, n, (int)blockIdx.x, (int)blockIdx.y, acc, flags);
}

// ---- sorted (entry, sample, corner) lists for the gather-style backward (dvt_grid_dev.h) ---------------------
// One 1024-thread workgroup per (level, step, fit): the 4 * n (<= 8192) corner pairs of the step's samples as
// 64-bit words (entry << 13 | sample << 2 | corner), padded with ~0, bitonic-sorted in 64 KB of LDS (91
// compare-exchange rounds), written back as keys / payloads / weights.  ~10 us per workgroup, two per CU: a chunk of
// 128 steps x 16 levels costs ~40 us, i.e. 0.3 us per step against the ~20 us per step the atomics cost.
namespace {
constexpr int GS_N = 8192;
struct GridSortArgs {
  DvtGridTable T;
  const float2* xy[DVT_FIT_BATCH_MAX];
  const int32_t* ridx[DVT_FIT_BATCH_MAX];
  uint32_t* keys[DVT_FIT_BATCH_MAX];
  uint16_t* pay[DVT_FIT_BATCH_MAX];
  float* w[DVT_FIT_BATCH_MAX];
  uint32_t* ukeys[DVT_FIT_BATCH_MAX];  // optional: [steps][L][nt] ABSOLUTE indices of the distinct entries, ascending
  int32_t* ucount[DVT_FIT_BATCH_MAX];  //           [steps][L] how many
  int n;
};

__global__ __launch_bounds__(1024) void grid_sort_kernel(GridSortArgs a) {
  __shared__ unsigned long long sk[GS_N];
  const int l = blockIdx.x, t = blockIdx.y, f = blockIdx.z, tid = threadIdx.x;
  const int n = a.n, nt = 4 * n, L = a.T.n_levels;
  const float2* __restrict__ xy = a.xy[f];
  const int32_t* __restrict__ ridx = a.ridx[f] + (size_t)t * n;
  const uint32_t off = a.T.offset[l];
  for (int smp = tid; smp < GS_N / 4; smp += 1024) {
    if (smp < n) {
      const float2 p = xy[ridx[smp]];
      uint32_t idx[4];
      float w[4];
      corners2d(a.T, l, p.x, p.y, idx, w);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        sk[4 * smp + c] = ((unsigned long long)(idx[c] - off) << 13) | (unsigned long long)(4 * smp + c);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) sk[4 * smp + c] = ~0ull;
    }
  }
  __syncthreads();
  for (int k = 2; k <= GS_N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int q = 0; q < GS_N / 2 / 1024; ++q) {
        const int p = tid + 1024 * q;
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = i | j;
        const unsigned long long x = sk[i], y = sk[ixj];
        const bool up = (i & k) == 0;
        if ((x > y) == up) {
          sk[i] = y;
          sk[ixj] = x;
        }
      }
      __syncthreads();
    }
  }
  // thread tid owns the sorted positions [8 tid, 8 tid + 8)
  const size_t base = ((size_t)t * L + l) * nt;
  unsigned long long e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = sk[8 * tid + j];
  const unsigned long long before = tid > 0 ? sk[8 * tid - 1] : ~0ull;
  __syncthreads();  // sk is free from here on
  int heads = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int u = 8 * tid + j;
    if (u < nt) {
      const uint32_t pu = (uint32_t)(e[j] & 8191u);
      const float2 p = xy[ridx[pu >> 2]];
      uint32_t idx[4];
      float w[4];
      corners2d(a.T, l, p.x, p.y, idx, w);
      a.keys[f][base + u] = (uint32_t)(e[j] >> 13);
      a.pay[f][base + u] = (uint16_t)pu;
      a.w[f][base + u] = w[pu & 3];
      const unsigned long long pk = (j ? e[j - 1] : before) >> 13;
      heads += (u == 0 || pk != (e[j] >> 13)) ? 1 : 0;
    }
  }
  if (a.ukeys[f] == nullptr) return;
  // exclusive scan of the per-thread head counts (1024 threads): wave scan + 16 wave totals through LDS
  int* scan = reinterpret_cast<int*>(sk);
  const int lane = tid & 63, wave = tid >> 6;
  int incl = heads;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) scan[wave] = incl;
  __syncthreads();
  int wave_base = 0, total = 0;
  for (int wv = 0; wv < 16; ++wv) {
    const int c = scan[wv];
    if (wv < wave) wave_base += c;
    total += c;
  }
  int pos = wave_base + incl - heads;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int u = 8 * tid + j;
    if (u < nt) {
      const unsigned long long pk = (j ? e[j - 1] : before) >> 13;
      if (u == 0 || pk != (e[j] >> 13)) a.ukeys[f][base + pos++] = off + (uint32_t)(e[j] >> 13);
    }
  }
  if (tid == 0) a.ucount[f][(size_t)t * L + l] = total;
}
}  // namespace

bool dvt_grid_sorted_ok(const DvtGridTable* T, int n) {
  if (!T || n <= 0 || n > GS_N / 4 || (n % 256)) return false;
  for (int l = 0; l < T->n_levels; ++l)
    if (T->entries[l] > (1u << 20)) return false;  // entry index must fit the 51 - 13 bits comfortably and, more to
                                                   // the point, the 32-bit key array
  return true;
}

int dvt_grid_sort_k(const DvtGridTable* T, int k, const float* const* xy, const int32_t* const* ridx, int n, int steps,
                    uint32_t* const* keys, uint16_t* const* pay, float* const* w, hipStream_t s,
                    uint32_t* const* ukeys, int32_t* const* ucount) {
  if (!dvt_grid_sorted_ok(T, n) || k < 1 || k > DVT_FIT_BATCH_MAX || steps < 1 || steps > 65535) return DVT_E_BADARG;
  GridSortArgs a{};
  a.T = *T;
  a.n = n;
  for (int f = 0; f < k; ++f) {
    if (!xy[f] || !ridx[f] || !keys[f] || !pay[f] || !w[f]) return DVT_E_BADARG;
    a.xy[f] = reinterpret_cast<const float2*>(xy[f]);
    a.ridx[f] = ridx[f];
    a.keys[f] = keys[f];
    a.pay[f] = pay[f];
    a.w[f] = w[f];
    a.ukeys[f] = ukeys ? ukeys[f] : nullptr;
    a.ucount[f] = ucount ? ucount[f] : nullptr;
    if ((a.ukeys[f] == nullptr) != (a.ucount[f] == nullptr)) return DVT_E_BADARG;
  }
  hipLaunchKernelGGL(grid_sort_kernel, dim3(T->n_levels, steps, k), dim3(1024), 0, s, a);
  DVT_CHECK_LAUNCH();
  return 0;
}

namespace {
struct GridGatherArgs {
  DvtGridTable T;
  GridSortedPtrs gs;
  const float* d_enc[DVT_FIT_BATCH_MAX];
  float* d_params[DVT_FIT_BATCH_MAX];
  uint32_t* touched[DVT_FIT_BATCH_MAX];
};
__global__ __launch_bounds__(1024) void grid_gather_kernel(GridGatherArgs a) {
  const int parts = a.gs.nt >> 10, bx = blockIdx.x, fy = blockIdx.y;
  grid_gather_body(a.T, a.gs, fy, bx / parts, bx % parts, a.d_enc[fy], a.d_params[fy], a.touched[fy]);
}
}  // namespace

int dvt_grid_gather_k(const DvtGridTable* T, int k, const uint32_t* const* keys, const uint16_t* const* pay,
                      const float* const* w, int n, uint32_t bitmap_end, const float* const* d_enc, float* const* d_params,
                      uint32_t* const* touched, hipStream_t s) {
  if (!dvt_grid_sorted_ok(T, n) || k < 1 || k > DVT_FIT_BATCH_MAX) return DVT_E_BADARG;
  GridGatherArgs a{};
  a.T = *T;
  a.gs.nt = 4 * n;
  a.gs.bitmap_end = bitmap_end;
  for (int f = 0; f < k; ++f) {
    if (!keys[f] || !pay[f] || !w[f] || !d_enc[f] || !d_params[f]) return DVT_E_BADARG;
    a.gs.keys[f] = keys[f];
    a.gs.pay[f] = pay[f];
    a.gs.w[f] = w[f];
    a.d_enc[f] = d_enc[f];
    a.d_params[f] = d_params[f];
    a.touched[f] = touched ? touched[f] : nullptr;
  }
  // algorithmic bytes: the lists (10 B per pair) + 32 B of d_enc per pair + one 32-B store per distinct entry (<= pairs)
  DvtProbeScope probe(DVT_PROBE_GRID, s, (double)k * T->n_levels * a.gs.nt * (10 + 32 + 32));
  hipLaunchKernelGGL(grid_gather_kernel, dim3(T->n_levels * (a.gs.nt >> 10), k), dim3(1024), 0, s, a);
  DVT_CHECK_LAUNCH();
  return 0;
}

void dvt_grid_bwd_plan(const DvtGridTable& T, int n, GridBwdPlan* plan) {
  plan->n_lds_blocks = 0;
  int l = 0;
  for (; l < T.n_levels; ++l) {
    if (T.entries[l] > (uint32_t)g_grid_lds_level_max) break;
    const int LDS_CHUNK = g_grid_lds_chunk;
    const int chunks = (int)((T.entries[l] + LDS_CHUNK - 1) / LDS_CHUNK);
    // expected LDS lane-atomics per chunk workgroup = n * 4 corners * 8 features / chunks
    long long per_chunk = (long long)n * 32 / chunks;
    int sp = (int)((per_chunk + g_grid_lds_atomics_per_block - 1) / g_grid_lds_atomics_per_block);
    sp = sp < 1 ? 1 : (sp > 16 ? 16 : sp);
    plan->splits[l] = sp;
    plan->chunk_start[l] = plan->n_lds_blocks;
    plan->n_lds_blocks += chunks * sp;
  }
  plan->chunk_start[l] = plan->n_lds_blocks;
  plan->first_direct_level = l;  // levels are sorted by size: the rest is fine
}

int dvt_grid_tune(int lds_level_max) {
  if (lds_level_max == -256 || lds_level_max == -1024) {  // chunk size selector
    g_grid_lds_chunk = -lds_level_max;
    return 0;
  }
  if (lds_level_max < 0) {  // other negatives: the per-block LDS atomics target
    g_grid_lds_atomics_per_block = -lds_level_max;
    return 0;
  }
  g_grid_lds_level_max = lds_level_max;
  return 0;
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
  // algorithmic bytes: 4 corners x 32 B gathered + 32 B written per (sample, level)
  DvtProbeScope probe(DVT_PROBE_GRID, stream, (double)threads * (4 * 32 + 32));
  hipLaunchKernelGGL(grid_fwd_kernel, dim3(dvt_cdiv(threads, 256)), dim3(256), 0, stream, *tbl,
                     (const float2*)xy, ridx, (const float4*)params, (float4*)enc, n);
  DVT_CHECK_LAUNCH();
  return 0;
}

int dvt_grid_bwd_idx(const DvtGridTable* tbl, const float* xy, const int32_t* ridx,
                     const float* d_enc, float* d_params, uint32_t* touched, int n,
                     hipStream_t stream) {
  return dvt_grid_bwd_k(tbl, 1, &xy, &ridx, &d_enc, &d_params, &touched, n, stream);
}

int dvt_grid_bwd_k(const DvtGridTable* tbl, int k, const float* const* xy, const int32_t* const* ridx,
                   const float* const* d_enc, float* const* d_params, uint32_t* const* touched, int n,
                   hipStream_t stream, uint32_t bitmap_end) {
  if (!tbl || k < 1 || k > DVT_FIT_BATCH_MAX || n < 0 || tbl->n_features != 8) return DVT_E_BADARG;
  GridBwdPtrs q{};
  q.bitmap_end = bitmap_end;
  for (int f = 0; f < k; ++f) {
    if (!xy[f] || !d_enc[f] || !d_params[f]) return DVT_E_BADARG;
    q.xy[f] = (const float2*)xy[f];
    q.ridx[f] = ridx[f];
    q.d_enc[f] = d_enc[f];
    q.d_params[f] = d_params[f];
    q.touched[f] = touched[f];
  }
  if (n == 0) return 0;
  GridBwdPlan plan;
  dvt_grid_bwd_plan(*tbl, n, &plan);
  const long long threads = (long long)n * (tbl->n_levels - plan.first_direct_level) * 8;
  const int blocks = plan.n_lds_blocks + dvt_cdiv(threads, 1024);
  // algorithmic bytes: 32 B read + 4 corners x 32 B read-modify-write per (sample, level)
  DvtProbeScope probe(DVT_PROBE_GRID, stream, (double)k * n * tbl->n_levels * (32 + 4 * 64));
  if (g_grid_lds_chunk == LDS_CHUNK_SMALL)
    hipLaunchKernelGGL(grid_bwd_kernel<LDS_CHUNK_SMALL>, dim3(blocks, k), dim3(1024), 0, stream, *tbl,
                       plan, q, n);
  else
    hipLaunchKernelGGL(grid_bwd_kernel<LDS_CHUNK_BIG>, dim3(blocks, k), dim3(1024), 0, stream, *tbl,
                       plan, q, n);
  DVT_CHECK_LAUNCH();
  return 0;
}

// Fused step prologue of the fit: raw-row gather (one wave per row) and hash-grid forward in ONE
// launch (blocks [0, gather_blocks) gather, the rest encode) -- two fewer dependent launches.
struct PrepPtrs {  // per fit of a batched launch (blockIdx.y)
  const float2* xy[DVT_FIT_BATCH_MAX];
  const int32_t* ridx[DVT_FIT_BATCH_MAX];
  const float4* params[DVT_FIT_BATCH_MAX];
  float4* enc[DVT_FIT_BATCH_MAX];
  const float4* feat[DVT_FIT_BATCH_MAX];
  float4* raw[DVT_FIT_BATCH_MAX];
};

__global__ __launch_bounds__(256) void fit_prep_kernel(DvtGridTable T, PrepPtrs q, int n, int cq,
                                                       int gather_blocks) {
  const float2* __restrict__ xy = q.xy[blockIdx.y];
  const int32_t* __restrict__ ridx = q.ridx[blockIdx.y];
  const float4* __restrict__ params = q.params[blockIdx.y];
  float4* __restrict__ enc = q.enc[blockIdx.y];
  const float4* __restrict__ feat = q.feat[blockIdx.y];
  float4* __restrict__ raw = q.raw[blockIdx.y];
  if ((int)blockIdx.x < gather_blocks) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const float4* s = feat + (size_t)ridx[row] * cq;
    float4* d = raw + (size_t)row * cq;
    for (int q = lane; q < cq; q += 64) d[q] = s[q];
    return;
  }
  const int L = T.n_levels;
  const int t = (blockIdx.x - gather_blocks) * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  const int b = t / L, l = t - b * L;
  const float2 p = xy[ridx[b]];
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

int dvt_fit_prep(const DvtGridTable* tbl, const float* xy, const int32_t* ridx, const float* params,
                 float* enc, const float* feat, float* raw, int n, int c, hipStream_t stream) {
  return dvt_fit_prep_k(tbl, 1, &xy, &ridx, &params, &enc, &feat, &raw, n, c, stream);
}

int dvt_fit_prep_k(const DvtGridTable* tbl, int k, const float* const* xy, const int32_t* const* ridx,
                   const float* const* params, float* const* enc, const float* const* feat,
                   float* const* raw, int n, int c, hipStream_t stream) {
  if (!tbl || k < 1 || k > DVT_FIT_BATCH_MAX || n <= 0 || (c & 3)) return DVT_E_BADARG;
  PrepPtrs q{};
  for (int f = 0; f < k; ++f) {
    if (!xy[f] || !ridx[f] || !params[f] || !enc[f] || !feat[f] || !raw[f]) return DVT_E_BADARG;
    q.xy[f] = (const float2*)xy[f];
    q.ridx[f] = ridx[f];
    q.params[f] = (const float4*)params[f];
    q.enc[f] = (float4*)enc[f];
    q.feat[f] = (const float4*)feat[f];
    q.raw[f] = (float4*)raw[f];
  }
  const int gather_blocks = dvt_cdiv(n, 4);
  const long long threads = (long long)n * tbl->n_levels;
  DvtProbeScope probe(DVT_PROBE_GRID, stream, (double)k * threads * (4 * 32 + 32));
  hipLaunchKernelGGL(fit_prep_kernel, dim3(gather_blocks + dvt_cdiv(threads, 256), k), dim3(256), 0,
                     stream, *tbl, q, n, c / 4, gather_blocks);
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
