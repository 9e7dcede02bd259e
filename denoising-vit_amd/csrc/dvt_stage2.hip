// Stage-2 generalizable denoiser: forward, loss, backward and AdamW of
// `Denoiser` = pos_embed + num_blocks x timm Block on [batch, 1369, 768] feature maps.
//
// Reference: dvt/models/online_denoiser.py:13-104 (model), main_denoiser.py:204-221 (one step),
// timm 1.0.7 vision_transformer.Block / Attention / Mlp (absent third party; restated in oracle/stage2.py).
//
// Everything is exact fp32, like the reference.  The contractions -- linear layers forward / data gradient /
// weight gradient, and the four attention products per direction -- all go through ONE GEMM kernel family
// (dvt_gemm_f32_ex, v_mfma_f32_32x32x2_f32 behind a 3-stage LDS-DMA pipeline); this file adds the row-local
// pieces (LayerNorm forward/backward fused with the residual adds, softmax forward/backward, GELU, the loss
// with its gradient, AdamW) and the launch sequence.  Attention keeps its probabilities P [batch*heads, Tp, Tp]
// in HBM: at 288 GB per GPU the 3 GB that costs at batch 32 is cheaper than recomputing QK^T in the backward
// pass, and P is exactly what dV = P^T dO and dS = P (dP - rowsum(P dP)) need.
//
// Rows: an image owns tokens_pad rows; rows t >= tokens are all-zero in every activation that feeds a
// reduction over rows (xn, dY), so weight gradients and LayerNorm parameter gradients never see them; key
// columns >= tokens get probability 0.
#include <cstdlib>
#include "dvt_common.h"
#include "../../include/dvt_stage2.h"

// dvt_gemm_f32.hip: the fp32 extractor's 128 x 128 x 32 exact-fp32 MFMA tile (x . w^T + b, shapes per dvt_linear_big_ok)
int dvt_linear_fwd_big(const float* x, const float* w, const float* b, float* y, int m, int n, int k, hipStream_t s);
bool dvt_linear_big_ok(int m, int n, int k);
bool dvt_linear_wgrad_big_ok(int rows, int n, int k);
int dvt_linear_wgrad_big(const float* dy, const float* x, float* dw, float* db, int rows, int n, int k, int accumulate,
                         hipStream_t s);
int g_s2_fork_wgrad = 1;  // DVT_S2_FORK_WGRAD=0 / dvt_tune_set(18, mask) bit 5: weight gradients on the caller's stream (A/B)
int g_s2_attn_rows = 1;  // DVT_S2_ATTN_ROWS=0: the [Tp][Tp] products on the 64 x 64 GEMM tile + separate softmax passes (A/B)
int g_s2_fuse_softmax_bwd = 1;  // DVT_S2_FUSE_SOFTMAX_BWD=0: dP written, s2_softmax_bwd_kernel over it (A/B)
int g_s2_big_wgrad = 1;  // DVT_S2_BIG_WGRAD=0: the weight-gradient GEMMs on the 64 x 64 tile (A/B)
int g_s2_big_bwd = 1;  // DVT_S2_BIG_BWD=0: the data-gradient GEMMs on the 64 x 64 tile (A/B)
int g_s2_big_fwd = 1;  // DVT_S2_BIG=0 in the environment of the process: the 64 x 64 tile for the forward layers too (A/B)

namespace {

#define S2_TRY(x)              \
  do {                         \
    const int rc__ = (x);      \
    if (rc__ != 0) return rc__; \
  } while (0)

// ---- a row of C floats held by one wave: float4 index lane + 64 j, j < NJ -------------------------------
template <int C>
struct Row {
  static constexpr int NJ = (C / 4 + 63) / 64;
  float4 v[NJ];
  __device__ __forceinline__ void load(const float* p, int lane) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = lane + 64 * j;
      v[j] = (i < C / 4) ? p4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* p, int lane) const {
    float4* p4 = reinterpret_cast<float4*>(p);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = lane + 64 * j;
      if (i < C / 4) p4[i] = v[j];
    }
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ float sum() const {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    return wave_sum(s);
  }
};
#define ROW_FOR(j, NJ) _Pragma("unroll") for (int j = 0; j < NJ; ++j)

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float f4_dot(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// ==========================================================================================================
// (a [+ b]) -> sum, LayerNorm(sum) -> xn, per-row mean / rstd.  One wave per row.
//   a: packed [batch, T, C] (a_packed) or padded [R, C];  b: nullptr, pos_embed [T, C] (b_is_pos) or padded [R, C]
// Rows t >= T: sum = xn = 0, mean = rstd = 0.
// online_denoiser.py:88-89 (x + pos_embed), Block: x + attn(norm1(x)), x + mlp(norm2(x)).
// ==========================================================================================================
template <int C>
__global__ __launch_bounds__(256) void s2_add_ln_kernel(const float* __restrict__ a, int a_packed,
                                                        const float* __restrict__ b, int b_is_pos,
                                                        float* __restrict__ sum_out, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ xn,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int T,
                                                        int Tp, int R, float eps) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int img = r / Tp, t = r - img * Tp;
  Row<C> x;
  if (t >= T) {
    x.zero();
    if (sum_out) x.store(sum_out + (size_t)r * C, lane);
    x.store(xn + (size_t)r * C, lane);
    if (lane == 0) {
      mean[r] = 0.f;
      rstd[r] = 0.f;
    }
    return;
  }
  x.load(a_packed ? a + ((size_t)img * T + t) * C : a + (size_t)r * C, lane);
  if (b) {
    Row<C> y;
    y.load(b_is_pos ? b + (size_t)t * C : b + (size_t)r * C, lane);
    ROW_FOR(j, Row<C>::NJ) x.v[j] = f4_add(x.v[j], y.v[j]);
  }
  if (sum_out) x.store(sum_out + (size_t)r * C, lane);
  const float mu = x.sum() * (1.0f / C);
  Row<C> d;
  float ss = 0.f;
  ROW_FOR(j, Row<C>::NJ) {
    const int i = lane + 64 * j;
    d.v[j] = (i < C / 4) ? f4_sub(x.v[j], make_float4(mu, mu, mu, mu)) : make_float4(0.f, 0.f, 0.f, 0.f);
    ss += f4_dot(d.v[j], d.v[j]);
  }
  const float var = wave_sum(ss) * (1.0f / C);
  const float rs = 1.0f / sqrtf(var + eps);
  Row<C> g, be;
  g.load(gamma, lane);
  be.load(beta, lane);
  ROW_FOR(j, Row<C>::NJ) d.v[j] = f4_add(f4_mul(f4_scale(d.v[j], rs), g.v[j]), be.v[j]);
  d.store(xn + (size_t)r * C, lane);
  if (lane == 0) {
    mean[r] = mu;
    rstd[r] = rs;
  }
}

// pred[batch, T, C] = a + b (padded rows in, packed rows out): the last residual add of an inference forward
template <int C>
__global__ __launch_bounds__(256) void s2_add_unpack_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ pred, int T, int Tp, int R) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int img = r / Tp, t = r - img * Tp;
  if (t >= T) return;
  Row<C> x, y;
  x.load(a + (size_t)r * C, lane);
  y.load(b + (size_t)r * C, lane);
  ROW_FOR(j, Row<C>::NJ) x.v[j] = f4_add(x.v[j], y.v[j]);
  x.store(pred + ((size_t)img * T + t) * C, lane);
}

// ==========================================================================================================
// LayerNorm backward fused with the residual-path gradient:
//   dx = dres + rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd
//   dgamma += sum_rows dy * xhat,  dbeta += sum_rows dy
// A 256-thread block walks 32 rows (8 per wave), keeps the parameter-gradient partials in registers, reduces
// the 4 waves through LDS and issues ONE atomic per column.
// ==========================================================================================================
template <int C>
__global__ __launch_bounds__(256) void s2_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ dres,
                                                        float* __restrict__ dx, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int R) {
  constexpr int NJ = Row<C>::NJ;
  __shared__ float red[2][4][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Row<C> gm, ag, ab;
  gm.load(gamma, lane);
  ag.zero();
  ab.zero();
  const int r0 = blockIdx.x * 32 + wave * 8;
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + i;
    if (r >= R) break;
    const float mu = mean[r], rs = rstd[r];
    Row<C> d, xv, o;
    d.load(dy + (size_t)r * C, lane);
    xv.load(x + (size_t)r * C, lane);
    float s1 = 0.f, s2 = 0.f;
    ROW_FOR(j, NJ) {
      const int idx = lane + 64 * j;
      const float4 xh = (idx < C / 4) ? f4_scale(f4_sub(xv.v[j], make_float4(mu, mu, mu, mu)), rs)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 g = f4_mul(d.v[j], gm.v[j]);
      ag.v[j] = f4_add(ag.v[j], f4_mul(d.v[j], xh));
      ab.v[j] = f4_add(ab.v[j], d.v[j]);
      s1 += (g.x + g.y) + (g.z + g.w);
      s2 += f4_dot(g, xh);
      xv.v[j] = xh;
      d.v[j] = g;
    }
    const float m1 = wave_sum(s1) * (1.0f / C), m2 = wave_sum(s2) * (1.0f / C);
    if (dres) o.load(dres + (size_t)r * C, lane); else o.zero();
    ROW_FOR(j, NJ) {
      const float4 t = f4_sub(f4_sub(d.v[j], make_float4(m1, m1, m1, m1)), f4_scale(xv.v[j], m2));
      o.v[j] = f4_add(o.v[j], f4_scale(t, rs));
    }
    o.store(dx + (size_t)r * C, lane);
  }
  ROW_FOR(j, NJ) {
    const int idx = lane + 64 * j;
    if (idx < C / 4) {
      *reinterpret_cast<float4*>(&red[0][wave][4 * idx]) = ag.v[j];
      *reinterpret_cast<float4*>(&red[1][wave][4 * idx]) = ab.v[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomic_add_f32(dgamma + c, (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
    atomic_add_f32(dbeta + c, (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
  }
}

// ==========================================================================================================
// GELU (nn.GELU(), exact erf) forward / backward, elementwise over float4
// ==========================================================================================================
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__global__ __launch_bounds__(256) void s2_gelu_kernel(const float4* __restrict__ h, float4* __restrict__ a, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = h[i];
  a[i] = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
}
__global__ __launch_bounds__(256) void s2_gelu_bwd_kernel(const float4* __restrict__ h, float4* __restrict__ da, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = h[i], d = da[i];
  da[i] = make_float4(d.x * gelu_grad_f(v.x), d.y * gelu_grad_f(v.y), d.z * gelu_grad_f(v.z), d.w * gelu_grad_f(v.w));
}

// ==========================================================================================================
// softmax over the valid keys of one row of S [batch*heads*Tp rows][Tp], in place:  P = softmax(scale * S).
// One wave per row, two passes (online max / sum, then normalise).  Query rows >= T and key columns >= T: 0.
// timm Attention: q * scale, attn = q @ k^T, softmax(dim=-1).
// ==========================================================================================================
__global__ __launch_bounds__(256) void s2_softmax_kernel(float* __restrict__ S, int T, int Tp, int64_t rows, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float4* row = reinterpret_cast<float4*>(S + r * Tp);
  const int q = (int)(r % Tp), n4 = Tp / 4;
  if (q >= T) {
    for (int i = lane; i < n4; i += 64) row[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  float m = -3.0e38f, l = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const float4 v = row[i];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (4 * i + c < T) {
        const float s = e[c] * scale;
        const float mn = fmaxf(m, s);
        l = l * __expf(m - mn) + __expf(s - mn);
        m = mn;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
    const float mn = fmaxf(m, m2);
    l = l * __expf(m - mn) + l2 * __expf(m2 - mn);
    m = mn;
  }
  const float inv = 1.0f / l;
  for (int i = lane; i < n4; i += 64) {
    const float4 v = row[i];
    float4 p;
    p.x = (4 * i + 0 < T) ? __expf(v.x * scale - m) * inv : 0.f;
    p.y = (4 * i + 1 < T) ? __expf(v.y * scale - m) * inv : 0.f;
    p.z = (4 * i + 2 < T) ? __expf(v.z * scale - m) * inv : 0.f;
    p.w = (4 * i + 3 < T) ? __expf(v.w * scale - m) * inv : 0.f;
    row[i] = p;
  }
}

// dS = scale * P * (dP - sum_k P dP), in place over dP (the gradient w.r.t. q k^T before the scale)
__global__ __launch_bounds__(256) void s2_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, int Tp,
                                                             int64_t rows, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float4* p = reinterpret_cast<const float4*>(P + r * Tp);
  float4* d = reinterpret_cast<float4*>(dP + r * Tp);
  const int n4 = Tp / 4;
  float dot = 0.f;
  for (int i = lane; i < n4; i += 64) dot += f4_dot(p[i], d[i]);
  dot = wave_sum(dot);
  for (int i = lane; i < n4; i += 64) {
    const float4 pv = p[i], dv = d[i];
    d[i] = make_float4(scale * pv.x * (dv.x - dot), scale * pv.y * (dv.y - dot), scale * pv.z * (dv.z - dot),
                       scale * pv.w * (dv.w - dot));
  }
}


// ==========================================================================================================
// Round 6: the two [Tp][Tp]-sized products of a head WITH the softmax arithmetic that used to run over their output.
// One workgroup = 128 query rows of one (image, head): 4 waves x 32 rows, the row operand (q, or d ao) lives in registers
// -- 32 fragment values per lane, as in the fp32 extractor's attention kernel --, the key-side operand (k, or v) streams
// through LDS in 32-key tiles (two buffers, one barrier per tile).  v_mfma_f32_32x32x2_f32 in BOTH operand orders off the
// SAME registers and the same LDS reads:
//   T-order  D = K_tile . A^T   lane holds query (lane & 31), keys kappa(r) + 4 (lane >> 5): row statistics are lane-local
//   N-order  D = A . K_tile^T   lane holds key (lane & 31), queries kappa(r) + 4 (lane >> 5): a half-wave stores 32
//                               consecutive keys of one query row = one whole 128-B line per instruction
// MODE 0, forward (main_denoiser.py:138-140 -> timm Attention.forward: softmax(q k^T / 8)): sweep 1 in T-order forms every
//   query's running max and sum (base 2, the scale folded into q), sweep 2 recomputes the logits in N-order and writes
//   P = 2^(s - m) / l ONCE.  Before: S written by a 64 x 64-tile GEMM (186 k workgroups of 32 MFMAs per wave), read and
//   rewritten by s2_softmax_kernel: 9 GB of traffic, 1.76 + 1.22 ms per step at batch 32.  Padded query rows and key columns
//   (>= T) get P = 0, as s2_softmax_kernel wrote them.
// MODE 1, backward: dS = scale P (.) (dao v^T - D) in N-order, D = rowsum(dP (.) P) = dao . ao from s2_rowdot_kernel; P is
//   read once (requested a tile ahead), dP never exists.
// ==========================================================================================================
constexpr int AR_Q = 128, AR_K = 32, AR_LD = 65;  // odd pitch: the 32 rows of a key-tile fragment read hit 32 banks
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void s2_attn_rows_kernel(const float* __restrict__ rowop, int ld_row, const float* __restrict__ keyop,
                                                           int ld_key, const float* __restrict__ Pin, const float* __restrict__ D,
                                                           float* __restrict__ out, int heads, int T, int Tp, float scale) {
  __shared__ float Ks[2][AR_K * AR_LD];
  __shared__ float stat[2][AR_Q];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h2 = lane >> 5;
  const int nqb = Tp / AR_Q;
  const int id = blockIdx.x;
  const int qb = id % nqb, hd = (id / nqb) % heads, b = id / (nqb * heads);
  const size_t row0 = (size_t)b * Tp;
  const int q0w = qb * AR_Q + wave * 32;  // first query row of this wave inside the image
  const float LOG2E = 1.4426950408889634f;
  // row-operand fragments of this lane: A[query q0w + j][d = 2 s + h2] (forward: q * scale * log2(e): softmax in base 2)
  float af[32];
  {
    const float* ap = rowop + (row0 + q0w + j) * ld_row + hd * 64 + h2;
    const float f = MODE == 0 ? scale * LOG2E : 1.0f;
#pragma unroll
    for (int s = 0; s < 32; ++s) af[s] = ap[2 * s] * f;
  }
  const float* kbase = keyop + row0 * ld_key + hd * 64;
  float* obase = out + ((size_t)(b * heads + hd) * Tp) * Tp;
  const float* pbase = MODE == 1 ? Pin + ((size_t)(b * heads + hd) * Tp) * Tp : nullptr;
  const int ntiles = Tp / AR_K;
  const int key0 = tid >> 4, dq0 = tid & 15;  // staging: 32 keys x 64 d = 512 float4, two per thread (second: key0 + 16)
  float4 kr[2];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int it = 0; it < 2; ++it)
      kr[it] = *reinterpret_cast<const float4*>(kbase + (size_t)(kt * AR_K + key0 + 16 * it) * ld_key + dq0 * 4);
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float* kd = Ks[buf] + (key0 + 16 * it) * AR_LD + dq0 * 4;
      kd[0] = kr[it].x; kd[1] = kr[it].y; kd[2] = kr[it].z; kd[3] = kr[it].w;
    }
  };
  if constexpr (MODE == 0) {
    // ---- sweep 1 (T-order): running max / sum per query, lane-local over its 16 keys of a tile
    float m_run = -1e30f, l_run = 0.f;
    fetch(0);
    park(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
      const int cur = kt & 1;
      const bool more = kt + 1 < ntiles;
      if (more) fetch(kt + 1);
      const float* K = Ks[cur];
      floatx16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 32; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(K[j * AR_LD + 2 * t + h2], af[t], s, 0, 0, 0);
      float tmax = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * AR_K + (r & 3) + 8 * (r >> 2) + 4 * h2;
        if (key >= T) s[r] = -1e30f;
        tmax = fmaxf(tmax, s[r]);
      }
      const float m_new = fmaxf(m_run, tmax);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) psum += __builtin_amdgcn_exp2f(s[r] - m_new);
      l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + psum;
      m_run = m_new;
      if (more) park(cur ^ 1);
      __syncthreads();
    }
    {  // the two lanes of a query (h2 = 0 / 1) hold disjoint keys: merge, then (m, 1 / l) of the wave's 32 queries -> LDS
      const float m2 = __shfl_xor(m_run, 32, 64), l2 = __shfl_xor(l_run, 32, 64);
      const float mt = fmaxf(m_run, m2);
      const float lt = l_run * __builtin_amdgcn_exp2f(m_run - mt) + l2 * __builtin_amdgcn_exp2f(m2 - mt);
      if (h2 == 0) {
        stat[0][wave * 32 + j] = mt;
        stat[1][wave * 32 + j] = 1.0f / lt;
      }
    }
    __syncthreads();
  }
  // ---- the N-order sweep: lane = key column (lane & 31), rows kappa(r) + 4 h2 of the wave's 32 queries
  float rs0[16], rs1[16];  // per row: forward (m, 1 / l); backward (D, -)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qi = (r & 3) + 8 * (r >> 2) + 4 * h2;
    if constexpr (MODE == 0) {
      rs0[r] = stat[0][wave * 32 + qi];
      rs1[r] = q0w + qi < T ? stat[1][wave * 32 + qi] : 0.f;  // padded query rows: P = 0
    } else {
      rs0[r] = D[(size_t)(b * heads + hd) * Tp + q0w + qi];
      rs1[r] = scale;
    }
  }
  float pr[16];
  auto fetch_p = [&](int kt) {
    if constexpr (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = (r & 3) + 8 * (r >> 2) + 4 * h2;
        pr[r] = pbase[(size_t)(q0w + qi) * Tp + kt * AR_K + j];
      }
    }
  };
  fetch(0);
  park(0);
  fetch_p(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < ntiles;
    if (more) fetch(kt + 1);
    const float* K = Ks[cur];
    floatx16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 32; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], K[j * AR_LD + 2 * t + h2], s, 0, 0, 0);
    const int key = kt * AR_K + j;
    float o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (MODE == 0) o[r] = key < T ? __builtin_amdgcn_exp2f(s[r] - rs0[r]) * rs1[r] : 0.f;
      else o[r] = rs1[r] * pr[r] * (s[r] - rs0[r]);
    }
    if (more) fetch_p(kt + 1);  // (behind the uses of this tile's P)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = (r & 3) + 8 * (r >> 2) + 4 * h2;
      obase[(size_t)(q0w + qi) * Tp + key] = o[r];
    }
    if (more) park(cur ^ 1);
    __syncthreads();
  }
}

// D[(b * heads + h) * Tp + t] = sum_d dO[b * Tp + t][64 h + d] * O[b * Tp + t][64 h + d] = rowsum(dP (.) P) of that (image, head, query)
// (O = P V, dP = dO V^T): what the softmax backward subtracts, from two [R][C] tensors instead of two [.., Tp][Tp] ones.  One
// wave per token row, 16 lanes per head pass (C / 64 heads, 4 per pass).
__global__ __launch_bounds__(256) void s2_rowdot_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                                        float* __restrict__ D, int R, int Tp, int C) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int b = r / Tp, t = r - b * Tp, heads = C >> 6;
  const float4* a = reinterpret_cast<const float4*>(dO + (size_t)r * C);
  const float4* o = reinterpret_cast<const float4*>(O + (size_t)r * C);
  for (int h0 = 0; h0 < heads; h0 += 4) {  // 64 lanes x float4 = 4 heads of 64 columns
    const int h = h0 + (lane >> 4);
    float v = h < heads ? f4_dot(a[h0 * 16 + lane], o[h0 * 16 + lane]) : 0.f;
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    if ((lane & 15) == 0 && h < heads) D[((size_t)b * heads + h) * Tp + t] = v;
  }
}

// ==========================================================================================================
// Loss and its gradient, one wave per row (main_denoiser.py:213-217):
//   o = a + b (the last residual add);  l2 = mean((o - t)^2) over batch*T*C;
//   cos_t = o.t / (max(|o|, 1e-8) max(|t|, 1e-8)) (F.cosine_similarity, eps 1e-8);  loss = l2 + 1 - mean_t cos_t
//   dout = 2 (o - t) / (batch T C) - (t / (|o| |t|) - cos o / |o|^2) / (batch T)
// acc[0] += sum (o - t)^2, acc[1] += sum cos (block partials, fp32 atomics).  Padded rows: dout = 0.
// ==========================================================================================================
template <int C>
__global__ __launch_bounds__(256) void s2_loss_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ target, float* __restrict__ pred,
                                                      float* __restrict__ dout, float* __restrict__ acc, int T, int Tp,
                                                      int R, float inv_el, float inv_tok) {
  __shared__ float part[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = blockIdx.x * 4 + wave;
  float se = 0.f, cs = 0.f;
  if (r < R) {
    const int img = r / Tp, t = r - img * Tp;
    Row<C> o;
    if (t >= T) {
      o.zero();
      o.store(dout + (size_t)r * C, lane);
    } else {
      Row<C> y, tg;
      o.load(a + (size_t)r * C, lane);
      y.load(b + (size_t)r * C, lane);
      tg.load(target + ((size_t)img * T + t) * C, lane);
      float s_d = 0.f, s_ot = 0.f, s_oo = 0.f, s_tt = 0.f;
      ROW_FOR(j, Row<C>::NJ) {
        o.v[j] = f4_add(o.v[j], y.v[j]);
        const float4 d = f4_sub(o.v[j], tg.v[j]);
        s_d += f4_dot(d, d);
        s_ot += f4_dot(o.v[j], tg.v[j]);
        s_oo += f4_dot(o.v[j], o.v[j]);
        s_tt += f4_dot(tg.v[j], tg.v[j]);
      }
      if (pred) o.store(pred + ((size_t)img * T + t) * C, lane);
      s_d = wave_sum(s_d);
      s_ot = wave_sum(s_ot);
      s_oo = wave_sum(s_oo);
      s_tt = wave_sum(s_tt);
      const float no = fmaxf(sqrtf(s_oo), 1e-8f), nt = fmaxf(sqrtf(s_tt), 1e-8f);
      const float cosv = s_ot / (no * nt);
      const float ka = 2.0f * inv_el, kt = inv_tok / (no * nt), ko = inv_tok * cosv / (no * no);
      ROW_FOR(j, Row<C>::NJ) {
        const float4 d = f4_sub(o.v[j], tg.v[j]);
        float4 g;
        g.x = ka * d.x - (kt * tg.v[j].x - ko * o.v[j].x);
        g.y = ka * d.y - (kt * tg.v[j].y - ko * o.v[j].y);
        g.z = ka * d.z - (kt * tg.v[j].z - ko * o.v[j].z);
        g.w = ka * d.w - (kt * tg.v[j].w - ko * o.v[j].w);
        o.v[j] = g;
      }
      o.store(dout + (size_t)r * C, lane);
      se = s_d;
      cs = cosv;
    }
  }
  if (lane == 0) {
    part[0][wave] = se;
    part[1][wave] = cs;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomic_add_f32(acc + 0, (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]));
    atomic_add_f32(acc + 1, (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]));
  }
}

// loss_out = {l2 + 1 - cos, l2, 1 - cos, 0} from the two accumulated sums
__global__ void s2_loss_finish_kernel(const float* __restrict__ acc, float* __restrict__ out, float inv_el, float inv_tok) {
  const float l2 = acc[0] * inv_el, cl = 1.0f - acc[1] * inv_tok;
  out[0] = l2 + cl;
  out[1] = l2;
  out[2] = cl;
  out[3] = 0.f;
}

// dpos[t, c] += sum over images of dx[img * Tp + t, c]   (pos_embed broadcasts over the batch)
__global__ __launch_bounds__(256) void s2_pos_grad_kernel(const float4* __restrict__ dx, float4* __restrict__ dpos, int batch,
                                                          int T, int Tp, int C4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)T * C4) return;
  const int t = (int)(i / C4), c = (int)(i - (int64_t)t * C4);
  float4 s = dpos[i];
  for (int b = 0; b < batch; ++b) s = f4_add(s, dx[((int64_t)b * Tp + t) * C4 + c]);
  dpos[i] = s;
}

// torch.optim.AdamW (decoupled weight decay), fused gradient scaling and zero_grad
__global__ __launch_bounds__(256) void s2_adamw_kernel(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                       float4* __restrict__ v, int64_t n4, float lr, float b1, float b2,
                                                       float eps, float wd, float step_size, float inv_sqrt_bc2,
                                                       float gscale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 gv = g[i];
  float4 pv = p[i], mv = m[i], vv = v[i];
  const float decay = 1.0f - lr * wd;
#define S2_ADAMW(c)                                                        \
  do {                                                                     \
    const float gg = gv.c * gscale;                                        \
    pv.c *= decay;                                                         \
    mv.c = b1 * mv.c + (1.0f - b1) * gg;                                   \
    vv.c = b2 * vv.c + (1.0f - b2) * gg * gg;                              \
    pv.c -= step_size * (mv.c / (sqrtf(vv.c) * inv_sqrt_bc2 + eps));       \
  } while (0)
  S2_ADAMW(x);
  S2_ADAMW(y);
  S2_ADAMW(z);
  S2_ADAMW(w);
#undef S2_ADAMW
  p[i] = pv;
  m[i] = mv;
  v[i] = vv;
  g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- host side ------------------------------------------------------------------------------------------
struct S2Offsets {
  int64_t pos;
  int64_t t[DVT_S2_MAX_BLOCKS][DVT_S2_TENSORS_PER_BLOCK];
  int64_t total;
};
enum { N1W = 0, N1B, QKVW, QKVB, PROJW, PROJB, N2W, N2B, FC1W, FC1B, FC2W, FC2B };

int check_cfg(const DvtS2Config* c) {
  if (!c) return DVT_E_BADARG;
  if (c->dim != 384 && c->dim != 768 && c->dim != 1024) return DVT_E_BADARG;
  if (c->heads * 64 != c->dim || c->mlp_dim <= 0 || c->mlp_dim % 64) return DVT_E_BADARG;
  if (c->tokens < 1 || c->tokens_pad < c->tokens || c->tokens_pad % 64) return DVT_E_BADARG;
  if (c->n_blocks < 1 || c->n_blocks > DVT_S2_MAX_BLOCKS || !(c->ln_eps > 0.f)) return DVT_E_BADARG;
  return 0;
}

void offsets(const DvtS2Config* c, S2Offsets* o) {
  const int64_t C = c->dim, F = c->mlp_dim;
  int64_t at = 0;
  o->pos = 0;
  if (c->enable_pe) at += (int64_t)c->tokens * C;
  const int64_t sz[DVT_S2_TENSORS_PER_BLOCK] = {C, C, 3 * C * C, 3 * C, C * C, C, C, C, F * C, F, C * F, C};
  for (int b = 0; b < c->n_blocks; ++b)
    for (int i = 0; i < DVT_S2_TENSORS_PER_BLOCK; ++i) {
      o->t[b][i] = at;
      at += sz[i];
    }
  o->total = at;
}

struct S2Block {  // activations a block keeps for its backward pass
  float *xin, *xn1, *mean1, *rstd1, *qkv, *P, *ao, *x1, *xn2, *mean2, *rstd2, *h, *a;
};
struct S2Work {
  S2Block blk[DVT_S2_MAX_BLOCKS];
  float *tmp, *d0, *d1, *d2, *dh, *dqkv, *dP, *acc;
  float* rowdot;  // training: [batch * heads * Tp] rowsum(dP (.) P) of the softmax backward (s2_rowdot_kernel)
  float* wT;  // training: one transposed weight matrix (max(3 C, F) x C floats), rebuilt in front of each data-gradient GEMM
};

int64_t carve(const DvtS2Config* c, int batch, int training, char* base, S2Work* w) {
  const int64_t R = (int64_t)batch * c->tokens_pad, C = c->dim, F = c->mlp_dim;
  const int64_t PP = (int64_t)batch * c->heads * c->tokens_pad * c->tokens_pad;
  int64_t o = 0;
  auto take = [&](int64_t floats) {
    float* p = base ? reinterpret_cast<float*>(base + o) : nullptr;
    o += (floats * 4 + 255) / 256 * 256;
    return p;
  };
  S2Work t{};
  const int nb = training ? c->n_blocks : 1;
  for (int b = 0; b < nb; ++b) {
    S2Block& k = t.blk[b];
    k.xin = take(R * C);
    k.xn1 = take(R * C);
    k.mean1 = take(R);
    k.rstd1 = take(R);
    k.qkv = take(R * 3 * C);
    k.P = take(PP);
    k.ao = take(R * C);
    k.x1 = take(R * C);
    k.xn2 = take(R * C);
    k.mean2 = take(R);
    k.rstd2 = take(R);
    k.h = take(R * F);
    k.a = take(R * F);
  }
  for (int b = nb; b < c->n_blocks; ++b) t.blk[b] = t.blk[0];  // inference: every block reuses one set
  if (!training && c->n_blocks > 1) t.d0 = take(R * C);        // ping-pong partner of blk[0].xin
  t.tmp = take(R * C);
  if (training) {
    t.d0 = take(R * C);
    t.d1 = take(R * C);
    t.d2 = take(R * C);
    t.dh = take(R * F);
    t.dqkv = take(R * 3 * C);
    t.dP = take(PP);
    t.acc = take(64);
    t.wT = take((3 * C > F ? 3 * C : F) * C);
    t.rowdot = take((int64_t)batch * c->heads * c->tokens_pad);
  }
  if (w) *w = t;
  return o;
}

template <typename K, typename... A>
int launch_rows(K kernel, int R, hipStream_t s, A... args) {
  hipLaunchKernelGGL(kernel, dim3(dvt_cdiv(R, 4)), dim3(256), 0, s, args...);
  DVT_CHECK_LAUNCH();
  return 0;
}

int add_ln(int C, const float* a, int a_packed, const float* b, int b_is_pos, float* sum_out, const float* g,
           const float* be, float* xn, float* mean, float* rstd, int T, int Tp, int R, float eps, hipStream_t s) {
  switch (C) {
    case 384: return launch_rows(s2_add_ln_kernel<384>, R, s, a, a_packed, b, b_is_pos, sum_out, g, be, xn, mean, rstd, T, Tp, R, eps);
    case 768: return launch_rows(s2_add_ln_kernel<768>, R, s, a, a_packed, b, b_is_pos, sum_out, g, be, xn, mean, rstd, T, Tp, R, eps);
    default: return launch_rows(s2_add_ln_kernel<1024>, R, s, a, a_packed, b, b_is_pos, sum_out, g, be, xn, mean, rstd, T, Tp, R, eps);
  }
}

int ln_bwd(int C, const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
           const float* dres, float* dx, float* dgamma, float* dbeta, int R, hipStream_t s) {
  const dim3 grid(dvt_cdiv(R, 32)), blk(256);
  switch (C) {
    case 384: hipLaunchKernelGGL(s2_ln_bwd_kernel<384>, grid, blk, 0, s, dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, R); break;
    case 768: hipLaunchKernelGGL(s2_ln_bwd_kernel<768>, grid, blk, 0, s, dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, R); break;
    default: hipLaunchKernelGGL(s2_ln_bwd_kernel<1024>, grid, blk, 0, s, dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, R); break;
  }
  DVT_CHECK_LAUNCH();
  return 0;
}

// y[R][n] = x[R][k] . w[n][k]^T + b
int lin_fwd(const float* x, const float* w, const float* b, float* y, int R, int n, int k, hipStream_t s) {
  // round 6: the forward linear layers take the fp32 extractor's 128 x 128 x 32 tile where the shape allows (R = batch x 1408
  // rows, n and k multiples of 128 / 32: every layer of the Block) -- 125 against 97 TF/s; summation order differs only
  if (g_s2_big_fwd && dvt_linear_big_ok(R, n, k)) return dvt_linear_fwd_big(x, w, b, y, R, n, k, s);
  DvtGemmEx g{};
  g.layout = 0;
  g.A = x; g.B = w; g.C = y;
  g.M = R; g.N = n; g.K = k;
  g.lda = k; g.ldb = k; g.ldc = n;
  g.bias = b;
  return dvt_gemm_f32_ex(&g, s);
}
// side stream + fork / join events of lin_bwd (created once per process, never destroyed; one trainer per process and device)
hipStream_t g_s2_side = nullptr;
hipEvent_t g_s2_ev_fork = nullptr, g_s2_ev_join = nullptr;
bool s2_side_stream(hipStream_t* out) {
  if (g_s2_side == nullptr) {
    if (hipStreamCreateWithFlags(&g_s2_side, hipStreamNonBlocking) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&g_s2_ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_s2_ev_join, hipEventDisableTiming) != hipSuccess)
      return false;
  }
  *out = g_s2_side;
  return true;
}

// out[k][n] = in[n][k] (n, k multiples of 32): 32 x 32 tiles through LDS, both sides in whole 128-B row pieces
__global__ __launch_bounds__(256) void s2_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int k) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = in[(size_t)(n0 + ty + 8 * i) * k + k0 + tx];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) out[(size_t)(k0 + ty + 8 * i) * n + n0 + tx] = tile[tx][ty + 8 * i];
}

// dx[R][k] = dy[R][n] . w[n][k];  dw[n][k] += dy^T . x;  db[n] += colsum(dy)
// wT (round 6): scratch for w^T [k][n].  With it the data gradient is a FORWARD linear layer of the transposed weight --
// dx = dy . (w^T)^T -- and takes the 128 x 128 x 32 tile (dvt_linear_fwd_big: both operands k-contiguous); the transposition is
// 2 x 9 MB of traffic at most per layer, the GEMM 0.07-0.2 TFLOP.  Summation order differs from the 64 x 64 kernel only.
int lin_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int R, int n, int k,
            hipStream_t s, float* wT = nullptr) {
  // Round 6: the weight gradient and the data gradient of a layer are independent products of the same dy.  Their grids are a few
  // rounds of the chip's 512 workgroup slots each (2112 tiles = 4.1 rounds for the 768-wide outputs: the last round is 1/8
  // full), so the weight gradient goes to a side stream and the two fill each other's tails; joined before this returns (the
  // caller's next kernels overwrite dy / x).
  hipStream_t sw = s;
  const bool fork = g_s2_fork_wgrad && dx != nullptr && s2_side_stream(&sw);
  if (fork) {
    if (hipEventRecord(g_s2_ev_fork, s) != hipSuccess || hipStreamWaitEvent(sw, g_s2_ev_fork, 0) != hipSuccess) return DVT_E_BADARG;
  } else {
    sw = s;
  }
  if (g_s2_big_wgrad && dvt_linear_wgrad_big_ok(R, n, k)) {  // round 6: the weight gradient on the 128 x 128 tile too
    S2_TRY(dvt_linear_wgrad_big(dy, x, dw, db, R, n, k, 1, sw));
  } else {
    DvtGemmEx g{};
    g.layout = 2;
    g.A = dy; g.B = x; g.C = dw;
    g.M = n; g.N = k; g.K = R;
    g.lda = n; g.ldb = k; g.ldc = k;
    g.colsum = db;
    g.accumulate = 1;
    S2_TRY(dvt_gemm_f32_ex(&g, sw));
  }
  if (!dx) return 0;
  int rc = 0;
  if (wT && g_s2_big_bwd && n % 32 == 0 && k % 32 == 0 && dvt_linear_big_ok(R, k, n)) {
    hipLaunchKernelGGL(s2_transpose_kernel, dim3(k / 32, n / 32), dim3(256), 0, s, w, wT, n, k);
    DVT_CHECK_LAUNCH();
    rc = dvt_linear_fwd_big(dy, wT, nullptr, dx, R, k, n, s);
  } else {
    DvtGemmEx d{};
    d.layout = 1;
    d.A = dy; d.B = w; d.C = dx;
    d.M = R; d.N = k; d.K = n;
    d.lda = n; d.ldb = k; d.ldc = k;
    rc = dvt_gemm_f32_ex(&d, s);
  }
  if (fork && (hipEventRecord(g_s2_ev_join, sw) != hipSuccess || hipStreamWaitEvent(s, g_s2_ev_join, 0) != hipSuccess))
    return DVT_E_BADARG;
  return rc;
}

// The six (image, head)-batched attention products.  q/k/v live in qkv [R][3C] at column offsets 0 / C / 2C
// (+ 64 head), P and dP are [batch*heads][Tp][Tp], per-head outputs are 64-column slices of [R][C] / [R][3C].
struct AttnDims {
  int batch, heads, Tp, C;
};
DvtGemmEx attn_gemm(const AttnDims& d, int layout, const float* A, int lda, long long sA0, long long sA1, const float* B,
                    int ldb, long long sB0, long long sB1, float* Cc, int ldc, long long sC0, long long sC1, int M, int N,
                    int K) {
  DvtGemmEx g{};
  g.layout = layout;
  g.A = A; g.B = B; g.C = Cc;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.nb0 = d.batch; g.nb1 = d.heads;
  g.sA0 = sA0; g.sA1 = sA1; g.sB0 = sB0; g.sB1 = sB1; g.sC0 = sC0; g.sC1 = sC1;
  return g;
}

int run(const DvtS2Config* c, const float* params, float* grads, const float* x, const float* target, float* pred,
        int batch, void* work, int64_t work_bytes, float* loss_out, hipStream_t s) {
  S2_TRY(check_cfg(c));
  const int training = grads != nullptr;
  if (!params || !x || batch < 1 || !work || (training && (!target || !loss_out)) || (!training && !pred)) return DVT_E_BADARG;
  S2Work w;
  if (carve(c, batch, training, reinterpret_cast<char*>(work), &w) > work_bytes) return DVT_E_BADARG;
  S2Offsets po;
  offsets(c, &po);
  const int C = c->dim, F = c->mlp_dim, T = c->tokens, Tp = c->tokens_pad, H = c->heads, NB = c->n_blocks;
  const int R = batch * Tp;
  const int64_t rowsP = (int64_t)batch * H * Tp;
  const float scale = 0.125f;  // head_dim^-0.5
  const bool attn_rows = g_s2_attn_rows && Tp % AR_Q == 0;  // s2_attn_rows_kernel walks whole blocks of 128 query rows
  const AttnDims ad{batch, H, Tp, C};
  const long long qs0 = (long long)Tp * 3 * C, qs1 = 64, ps0 = (long long)H * Tp * Tp, ps1 = (long long)Tp * Tp,
                  os0 = (long long)Tp * C, os1 = 64;
  auto P = [&](int b, int i) { return params + po.t[b][i]; };
  auto G = [&](int b, int i) { return grads + po.t[b][i]; };

  // ---- forward ----
  S2_TRY(add_ln(C, x, 1, c->enable_pe ? params + po.pos : nullptr, 1, w.blk[0].xin, P(0, N1W), P(0, N1B), w.blk[0].xn1,
                w.blk[0].mean1, w.blk[0].rstd1, T, Tp, R, c->ln_eps, s));
  for (int b = 0; b < NB; ++b) {
    S2Block k = w.blk[b];
    if (!training && (b & 1)) k.xin = w.d0;  // inference with several blocks: block inputs ping-pong
    S2_TRY(lin_fwd(k.xn1, P(b, QKVW), P(b, QKVB), k.qkv, R, 3 * C, C, s));
    {  // S = q k^T  ->  P = softmax(scale S)  ->  ao = P v
      DvtGemmEx g{};
      if (attn_rows) {  // round 6: q k^T and the softmax in one kernel, P written once
        hipLaunchKernelGGL(s2_attn_rows_kernel<0>, dim3((Tp / AR_Q) * H * batch), dim3(256), 0, s, (const float*)k.qkv, 3 * C,
                           (const float*)(k.qkv + C), 3 * C, (const float*)nullptr, (const float*)nullptr, k.P, H, T, Tp, scale);
        DVT_CHECK_LAUNCH();
      } else {
        g = attn_gemm(ad, 0, k.qkv, 3 * C, qs0, qs1, k.qkv + C, 3 * C, qs0, qs1, k.P, Tp, ps0, ps1, Tp, Tp, 64);
        S2_TRY(dvt_gemm_f32_ex(&g, s));
        hipLaunchKernelGGL(s2_softmax_kernel, dim3(dvt_cdiv(rowsP, 4)), dim3(256), 0, s, k.P, T, Tp, rowsP, scale);
        DVT_CHECK_LAUNCH();
      }
      g = attn_gemm(ad, 1, k.P, Tp, ps0, ps1, k.qkv + 2 * C, 3 * C, qs0, qs1, k.ao, C, os0, os1, Tp, 64, Tp);
      S2_TRY(dvt_gemm_f32_ex(&g, s));
    }
    S2_TRY(lin_fwd(k.ao, P(b, PROJW), P(b, PROJB), w.tmp, R, C, C, s));
    S2_TRY(add_ln(C, k.xin, 0, w.tmp, 0, k.x1, P(b, N2W), P(b, N2B), k.xn2, k.mean2, k.rstd2, T, Tp, R, c->ln_eps, s));
    S2_TRY(lin_fwd(k.xn2, P(b, FC1W), P(b, FC1B), k.h, R, F, C, s));
    {
      const int64_t n4 = (int64_t)R * F / 4;
      hipLaunchKernelGGL(s2_gelu_kernel, dim3(dvt_cdiv(n4, 256)), dim3(256), 0, s, (const float4*)k.h, (float4*)k.a, n4);
      DVT_CHECK_LAUNCH();
    }
    S2_TRY(lin_fwd(k.a, P(b, FC2W), P(b, FC2B), w.tmp, R, C, F, s));
    if (b + 1 < NB) {
      S2Block n = w.blk[b + 1];
      if (!training && ((b + 1) & 1)) n.xin = w.d0;
      if (!training && !((b + 1) & 1)) n.xin = w.blk[0].xin;
      S2_TRY(add_ln(C, k.x1, 0, w.tmp, 0, n.xin, P(b + 1, N1W), P(b + 1, N1B), n.xn1, n.mean1, n.rstd1, T, Tp, R,
                    c->ln_eps, s));
    }
  }
  const S2Block& last = w.blk[NB - 1];
  if (!training) {
    switch (C) {
      case 384: return launch_rows(s2_add_unpack_kernel<384>, R, s, (const float*)last.x1, (const float*)w.tmp, pred, T, Tp, R);
      case 768: return launch_rows(s2_add_unpack_kernel<768>, R, s, (const float*)last.x1, (const float*)w.tmp, pred, T, Tp, R);
      default: return launch_rows(s2_add_unpack_kernel<1024>, R, s, (const float*)last.x1, (const float*)w.tmp, pred, T, Tp, R);
    }
  }

  // ---- loss ----
  const float inv_el = 1.0f / ((float)batch * T * C), inv_tok = 1.0f / ((float)batch * T);
  {
    const hipError_t e = hipMemsetAsync(w.acc, 0, 64 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  switch (C) {
    case 384: S2_TRY(launch_rows(s2_loss_kernel<384>, R, s, (const float*)last.x1, (const float*)w.tmp, target, pred, w.d0, w.acc, T, Tp, R, inv_el, inv_tok)); break;
    case 768: S2_TRY(launch_rows(s2_loss_kernel<768>, R, s, (const float*)last.x1, (const float*)w.tmp, target, pred, w.d0, w.acc, T, Tp, R, inv_el, inv_tok)); break;
    default: S2_TRY(launch_rows(s2_loss_kernel<1024>, R, s, (const float*)last.x1, (const float*)w.tmp, target, pred, w.d0, w.acc, T, Tp, R, inv_el, inv_tok)); break;
  }
  hipLaunchKernelGGL(s2_loss_finish_kernel, dim3(1), dim3(1), 0, s, (const float*)w.acc, loss_out, inv_el, inv_tok);
  DVT_CHECK_LAUNCH();

  // ---- backward: d0 holds the gradient w.r.t. the current block's OUTPUT ----
  for (int b = NB - 1; b >= 0; --b) {
    const S2Block& k = w.blk[b];
    // mlp: out = x1 + fc2(gelu(fc1(norm2(x1))))
    S2_TRY(lin_bwd(w.d0, k.a, P(b, FC2W), w.dh, G(b, FC2W), G(b, FC2B), R, C, F, s, w.wT));
    {
      const int64_t n4 = (int64_t)R * F / 4;
      hipLaunchKernelGGL(s2_gelu_bwd_kernel, dim3(dvt_cdiv(n4, 256)), dim3(256), 0, s, (const float4*)k.h, (float4*)w.dh, n4);
      DVT_CHECK_LAUNCH();
    }
    S2_TRY(lin_bwd(w.dh, k.xn2, P(b, FC1W), w.d2, G(b, FC1W), G(b, FC1B), R, F, C, s, w.wT));
    S2_TRY(ln_bwd(C, w.d2, k.x1, k.mean2, k.rstd2, P(b, N2W), w.d0, w.d1, G(b, N2W), G(b, N2B), R, s));  // d1 = d x1
    // attention: x1 = xin + proj(attn(norm1(xin)))
    S2_TRY(lin_bwd(w.d1, k.ao, P(b, PROJW), w.d2, G(b, PROJW), G(b, PROJB), R, C, C, s, w.wT));  // d2 = d ao
    {
      // dV = P^T dao -- independent of the dS chain below (both read P and dao): on the side stream beside it (round 6, as the weight
      // gradients in lin_bwd); dq and dk, two products of the same dS, likewise.  Joined before lin_bwd reads dqkv.
      hipStream_t sv = s;
      const bool fork = g_s2_fork_wgrad && s2_side_stream(&sv);
      if (!fork) sv = s;
      if (fork && (hipEventRecord(g_s2_ev_fork, s) != hipSuccess || hipStreamWaitEvent(sv, g_s2_ev_fork, 0) != hipSuccess))
        return DVT_E_BADARG;
      DvtGemmEx g = attn_gemm(ad, 2, k.P, Tp, ps0, ps1, w.d2, C, os0, os1, w.dqkv + 2 * C, 3 * C, qs0, qs1, Tp, 64, Tp);
      S2_TRY(dvt_gemm_f32_ex(&g, sv));
      // dP = dao v^T, and the softmax backward dS = scale P (.) (dP - rowsum(dP (.) P)).  Round 6: rowsum(dP (.) P) = dao . ao per
      // (image, head, query) comes from the two [R][C] tensors (s2_rowdot_kernel) and the backward is the EPILOGUE of the dP
      // product -- dP is never written, P read once (before: 3 GB written + 9 GB read + 3 GB written by s2_softmax_bwd_kernel)
      g = attn_gemm(ad, 0, w.d2, C, os0, os1, k.qkv + 2 * C, 3 * C, qs0, qs1, w.dP, Tp, ps0, ps1, Tp, Tp, 64);
      if (g_s2_fuse_softmax_bwd && attn_rows) {
        hipLaunchKernelGGL(s2_rowdot_kernel, dim3(dvt_cdiv(R, 4)), dim3(256), 0, s, (const float*)w.d2, (const float*)k.ao, w.rowdot, R, Tp, C);
        DVT_CHECK_LAUNCH();
        hipLaunchKernelGGL(s2_attn_rows_kernel<1>, dim3((Tp / AR_Q) * H * batch), dim3(256), 0, s, (const float*)w.d2, C,
                           (const float*)(k.qkv + 2 * C), 3 * C, (const float*)k.P, (const float*)w.rowdot, w.dP, H, T, Tp, scale);
        DVT_CHECK_LAUNCH();
      } else if (g_s2_fuse_softmax_bwd) {
        hipLaunchKernelGGL(s2_rowdot_kernel, dim3(dvt_cdiv(R, 4)), dim3(256), 0, s, (const float*)w.d2, (const float*)k.ao, w.rowdot, R, Tp, C);
        DVT_CHECK_LAUNCH();
        g.smul = k.P;
        g.rowsub = w.rowdot;
        g.oscale = scale;
        S2_TRY(dvt_gemm_f32_ex(&g, s));
      } else {
        S2_TRY(dvt_gemm_f32_ex(&g, s));
        hipLaunchKernelGGL(s2_softmax_bwd_kernel, dim3(dvt_cdiv(rowsP, 4)), dim3(256), 0, s, (const float*)k.P, w.dP, Tp, rowsP, scale);
        DVT_CHECK_LAUNCH();
      }
      // dq = dS k (main stream),  dk = dS^T q (side stream: behind dV there, and behind dS here)
      if (fork && (hipEventRecord(g_s2_ev_fork, s) != hipSuccess || hipStreamWaitEvent(sv, g_s2_ev_fork, 0) != hipSuccess))
        return DVT_E_BADARG;
      g = attn_gemm(ad, 1, w.dP, Tp, ps0, ps1, k.qkv + C, 3 * C, qs0, qs1, w.dqkv, 3 * C, qs0, qs1, Tp, 64, Tp);
      S2_TRY(dvt_gemm_f32_ex(&g, s));
      g = attn_gemm(ad, 2, w.dP, Tp, ps0, ps1, k.qkv, 3 * C, qs0, qs1, w.dqkv + C, 3 * C, qs0, qs1, Tp, 64, Tp);
      S2_TRY(dvt_gemm_f32_ex(&g, sv));
      if (fork && (hipEventRecord(g_s2_ev_join, sv) != hipSuccess || hipStreamWaitEvent(s, g_s2_ev_join, 0) != hipSuccess))
        return DVT_E_BADARG;
    }
    S2_TRY(lin_bwd(w.dqkv, k.xn1, P(b, QKVW), w.d2, G(b, QKVW), G(b, QKVB), R, 3 * C, C, s, w.wT));
    S2_TRY(ln_bwd(C, w.d2, k.xin, k.mean1, k.rstd1, P(b, N1W), w.d1, w.d0, G(b, N1W), G(b, N1B), R, s));  // d0 = d xin
  }
  if (c->enable_pe) {
    const int64_t n = (int64_t)T * (C / 4);
    hipLaunchKernelGGL(s2_pos_grad_kernel, dim3(dvt_cdiv(n, 256)), dim3(256), 0, s, (const float4*)w.d0,
                       (float4*)(grads + po.pos), batch, T, Tp, C / 4);
    DVT_CHECK_LAUNCH();
  }
  return 0;
}

}  // namespace

extern "C" int dvt_s2_param_offsets(const DvtS2Config* cfg, int64_t* out) {
  if (!out) return DVT_E_BADARG;
  S2_TRY(check_cfg(cfg));
  S2Offsets o;
  offsets(cfg, &o);
  out[0] = o.pos;
  for (int b = 0; b < cfg->n_blocks; ++b)
    for (int i = 0; i < DVT_S2_TENSORS_PER_BLOCK; ++i) out[1 + DVT_S2_TENSORS_PER_BLOCK * b + i] = o.t[b][i];
  out[1 + DVT_S2_TENSORS_PER_BLOCK * cfg->n_blocks] = o.total;
  return 0;
}

extern "C" int64_t dvt_s2_workspace_bytes(const DvtS2Config* cfg, int batch, int training) {
  if (check_cfg(cfg) != 0 || batch < 1) return -1;
  return carve(cfg, batch, training, nullptr, nullptr);
}

// dvt_tune_set(18, mask): the same switches from inside a process (tests / A/B tools): bit 0 forward layers, 1 data gradients,
// 2 weight gradients on the 128 x 128 tile, 3 softmax fused into the attention products, 4 softmax backward without a dP pass,
// 5 weight gradients on a side stream beside the data gradients; 63 = default.  Results differ in summation order only.
static void s2_read_env();
int dvt_s2_tune(int mask) {
  if (mask < 0 || mask > 63) return DVT_E_BADARG;
  s2_read_env();  // (so that a later first call does not overwrite this)
  g_s2_big_fwd = mask & 1;
  g_s2_big_bwd = (mask >> 1) & 1;
  g_s2_big_wgrad = (mask >> 2) & 1;
  g_s2_attn_rows = (mask >> 3) & 1;
  g_s2_fuse_softmax_bwd = (mask >> 4) & 1;
  g_s2_fork_wgrad = (mask >> 5) & 1;
  return 0;
}

static void s2_read_env() {
  static bool done = false;
  if (done) return;
  done = true;
  const char* e = getenv("DVT_S2_BIG");
  if (e && e[0] == '0') g_s2_big_fwd = 0;
  e = getenv("DVT_S2_BIG_BWD");
  if (e && e[0] == '0') g_s2_big_bwd = 0;
  e = getenv("DVT_S2_FORK_WGRAD");
  if (e && e[0] == '0') g_s2_fork_wgrad = 0;
  e = getenv("DVT_S2_ATTN_ROWS");
  if (e && e[0] == '0') g_s2_attn_rows = 0;
  e = getenv("DVT_S2_FUSE_SOFTMAX_BWD");
  if (e && e[0] == '0') g_s2_fuse_softmax_bwd = 0;
  e = getenv("DVT_S2_BIG_WGRAD");
  if (e && e[0] == '0') g_s2_big_wgrad = 0;
}

extern "C" int dvt_s2_forward(const DvtS2Config* cfg, const float* params, const float* x, float* pred, int batch,
                              void* work, int64_t work_bytes, void* stream) {
  s2_read_env();
  return run(cfg, params, nullptr, x, nullptr, pred, batch, work, work_bytes, nullptr, (hipStream_t)stream);
}

extern "C" int dvt_s2_train_step(const DvtS2Config* cfg, const float* params, float* grads, const float* x,
                                 const float* target, float* pred, int batch, void* work, int64_t work_bytes,
                                 float* loss_out, void* stream) {
  s2_read_env();
  if (!grads) return DVT_E_BADARG;
  return run(cfg, params, grads, x, target, pred, batch, work, work_bytes, loss_out, (hipStream_t)stream);
}

extern "C" int dvt_adamw_step(float* params, float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  if (!params || !grads || !m || !v || n <= 0 || (n & 3) || step < 1) return DVT_E_BADARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(s2_adamw_kernel, dim3(dvt_cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)params,
                     (float4*)grads, (float4*)m, (float4*)v, n4, lr, beta1, beta2, eps, weight_decay, step_size,
                     inv_sqrt_bc2, grad_scale);
  DVT_CHECK_LAUNCH();
  return 0;
}
