// fp32 frozen-ViT forward for gfx950: the reference's DEFAULT precision (`--dtype float32`,
// main_img_denoising.py:173, :257, :299 -- autocast disabled, timm runs in fp32).  C ABI in include/dvt_vit.h.
//
// Same layer sequence and token layout as the bf16 extractor (dvt_vit.hip; reference
// dvt/models/vit_wrapper.py:122-143 -> timm VisionTransformer.forward_intermediates), every operand fp32:
//   GEMMs      exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32, dvt_gemm_f32.hip: bitwise an fmaf chain),
//              157 TF/s peak = 1/16 of the bf16 rate -- this mode costs seconds per image, it exists so that
//              `--dtype float32` means what the reference means, not to be fast;
//   attention  flash-style on the same instruction: S^T = K.Q^T per 32-key tile so that a query's softmax
//              statistics are lane-local and P stays in the accumulator registers as the B operand of
//              O^T = V^T.P^T (no shuffles, no LDS round trip for P);
//   LayerNorm, exact-erf GELU (erff), LayerScale + residual, position embedding: fp32 element-wise kernels.
#include <math.h>

#include "../../include/dvt_vit.h"
#include "dvt_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned short bf16_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));

int dvt_linear_fwd_big(const float* x, const float* w, const float* b, float* y, int m, int n, int k, hipStream_t s);
bool dvt_linear_big_ok(int m, int n, int k);
int dvt_linear_fwd_big_epi(const float* x, const float* w, const float* b, float* y, int m, int n, int k, int epi,
                           const float* gamma, hipStream_t s);
// bf16 GEMM kernels of dvt_vit.hip (fp32 accumulation), used by the bf16x3 mode below
extern "C" int dvt_vit_gemm_f32out(const void* a, const void* w, const float* b, float* y, int m, int n, int k, void* stream);
extern "C" int dvt_vit_gemm_residual(const void* a, const void* w, const float* b, const float* gamma, float* x, int m,
                                     int n, int k, void* stream);
extern "C" int dvt_vit_attention_x3(const float* qkv, float* out, void* scratch, int batch, int heads, int s_pad,
                                    int n_valid, void* stream);
extern "C" int dvt_vit_attention_x3_presplit(const void* scratch, void* out, int batch, int heads, int s_pad, int n_valid,
                                             int split_out, void* stream);
extern "C" int64_t dvt_vit_attention_x3_scratch_bytes(int batch, int heads, int s_pad);
extern "C" int dvt_vit_gemm_gelu_x3(const void* a, const void* w, const float* b, void* out3, int m, int n, int k, void* stream);
extern "C" int dvt_vit_gemm_qkv_x3(const void* a, const void* w, const float* b, void* scratch, int m, int dim, int heads,
                                   int s_pad, int batch, int k, void* stream);

namespace {

inline int64_t up256b(int64_t x) { return (x + 255) / 256 * 256; }

struct VitWorkF {
  float *x, *xn, *qkv, *ao, *hid, *tmp, *col;
};

// rows of the linear layers' launches: batch * s_pad rounded up to whole 128-row tiles of the 128 x 128 x 32 GEMM (round 6: s_pad
// is any multiple of 32 -- 1370 tokens -> 1376 rows instead of 1408 --, so batch * s_pad need not be one); the rows behind
// batch * s_pad are computed on whatever they hold and never read
inline int64_t rows128(const DvtVitConfig* c, int batch) { return ((int64_t)batch * c->s_pad + 127) / 128 * 128; }

int64_t carve_f32(const DvtVitConfig* c, int batch, char* base, VitWorkF* w) {
  const int64_t T = rows128(c, batch);
  int64_t o = 0;
  auto take = [&](int64_t floats) {
    char* p = base ? base + o : nullptr;
    o += up256b(floats * 4);
    return reinterpret_cast<float*>(p);
  };
  VitWorkF t;
  t.x = take(T * c->dim);
  t.xn = take(T * c->dim);
  t.qkv = take(T * 3 * c->dim);
  t.ao = take(T * c->dim);
  t.hid = take(T * c->mlp_dim);
  t.tmp = take(T * c->dim);
  t.col = take(T * c->k_patch);
  if (w) *w = t;
  return o;
}

__global__ __launch_bounds__(256) void im2col_f32_kernel(const float* __restrict__ img, float* __restrict__ col,
                                                         DvtVitConfig c) {
  const int t = blockIdx.x;  // token row in [0, batch*s_pad)
  const int b = t / c.s_pad, s = t - b * c.s_pad;
  float* dst = col + (size_t)t * c.k_patch;
  const int pp = c.patch * c.patch;
  if (s < c.n_prefix || s >= c.n_tokens) {
    for (int k = threadIdx.x; k < c.k_patch; k += 256) dst[k] = 0.f;
    return;
  }
  const int py = (s - c.n_prefix) / c.grid_w, px = (s - c.n_prefix) - py * c.grid_w;
  const float* src = img + (size_t)b * 3 * c.img_h * c.img_w;
  for (int k = threadIdx.x; k < c.k_patch; k += 256) {
    float v = 0.f;
    if (k < 3 * pp) {
      const int ch = k / pp, rem = k - ch * pp, ky = rem / c.patch, kx = rem - ky * c.patch;
      v = src[((size_t)ch * c.img_h + py * c.stride + ky) * c.img_w + px * c.stride + kx];
    }
    dst[k] = v;
  }
}

// x[t] = prefix token (+ pos_embed[0] for cls when the table has a cls row) | patch embedding + pos_embed | 0 (pad)
__global__ __launch_bounds__(256) void embed_f32_kernel(const float4* __restrict__ y, float4* __restrict__ x,
                                                        const float4* __restrict__ cls, const float4* __restrict__ pos,
                                                        DvtVitConfig c) {
  const int t = blockIdx.x, s = t % c.s_pad, dq = c.dim >> 2;
  for (int q = threadIdx.x; q < dq; q += 256) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < c.n_prefix) {
      o = cls[(size_t)s * dq + q];
      if (c.pos_has_cls && s == 0) {
        const float4 p = pos[q];
        o = make_float4(o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w);
      }
    } else if (s < c.n_tokens) {
      const float4 v = y[(size_t)t * dq + q], p = pos[(size_t)(s - c.n_prefix + c.pos_has_cls) * dq + q];
      o = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
    }
    x[(size_t)t * dq + q] = o;
  }
}

// ---- bf16x3: an fp32 value as the sum of two bf16 -------------------------------------------------------------
// x = hi + lo + e, hi = bf16_rn(x), lo = bf16_rn(x - hi), |e| <= 2^-17 |x| (x - hi is exact in fp32).  A GEMM over the
// K-concatenated operands  A3 = [hi | hi | lo]  (activations)  and  W3 = [hi | lo | hi]  (weights)  accumulates
// a_hi w_hi + a_hi w_lo + a_lo w_hi in the fp32 MFMA accumulators: everything of the fp32 product except a_lo w_lo
// (2^-16 relative) and the two e terms.  This is what torch.set_float32_matmul_precision("high") allows for fp32
// matmuls ("bfloat16_3x"); the reference never sets it, so it is an opt-in of this build, not what `--dtype float32`
// means by default (include/dvt_vit.h).
// fp32 -> bf16 round-to-nearest-even (v_cvt_pk_bf16_f32), low half = a
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const hw_bf16x2_t v = __builtin_convertvector((f32x2_t){a, b}, hw_bf16x2_t);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split2(float v, float& hi_f, uint32_t& hi_b, uint32_t& lo_b) {
  hi_b = pack2(v, 0.f) & 0xffffu;
  hi_f = __uint_as_float(hi_b << 16);
  lo_b = pack2(v - hi_f, 0.f) & 0xffffu;
}
// four consecutive values -> their places in a [.. | .. | ..] row of width 3k (ORDER 0: hi hi lo, 1: hi lo hi)
template <int ORDER>
__device__ __forceinline__ void store_split4(bf16_t* row3, int k, int col, const float4& v) {
  float hf;
  uint32_t h[4], l[4];
  split2(v.x, hf, h[0], l[0]);
  split2(v.y, hf, h[1], l[1]);
  split2(v.z, hf, h[2], l[2]);
  split2(v.w, hf, h[3], l[3]);
  const uint2 hh = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
  const uint2 ll = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
  *reinterpret_cast<uint2*>(row3 + col) = hh;
  *reinterpret_cast<uint2*>(row3 + k + col) = ORDER == 0 ? hh : ll;
  *reinterpret_cast<uint2*>(row3 + 2 * k + col) = ORDER == 0 ? ll : hh;
}

// out3[r, :] = split of (GELU ? gelu(x[r, :]) : x[r, :]); x has `k` columns, out3 3k
template <int ORDER, bool GELU>
__global__ __launch_bounds__(256) void split3_kernel(const float4* __restrict__ x, bf16_t* __restrict__ out3,
                                                     long long nq, int kq) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
    float4 v = x[q];
    if (GELU) {  // nn.GELU(): exact erf form, as gelu_f32_kernel
      v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f));
      v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
      v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f));
      v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
    }
    const long long r = q / kq;
    const int c = (int)(q - r * kq) * 4;
    store_split4<ORDER>(out3 + r * (3LL * kq * 4), kq * 4, c, v);
  }
}

// SPLIT: y is a bf16 [rows, 3 * dim] row of split values instead of fp32 [rows, dim]
template <bool FINAL, bool SPLIT = false>
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            int rows, int dim, float eps, int s_pad, int n_tokens,
                                                            int n_prefix) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  size_t in_row = row;
  if (FINAL) {  // output row = b*(n_tokens-n_prefix) + (s-n_prefix), input row = b*s_pad + s
    const int per = n_tokens - n_prefix;
    const int bb = row / per, s = row - bb * per + n_prefix;
    in_row = (size_t)bb * s_pad + s;
  }
  const float4* xr = reinterpret_cast<const float4*>(x + in_row * dim);
  const int nq = dim >> 2;
  float4 v[4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      v[i] = xr[q];
      sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  const float mean = wave_sum(sum) / (float)dim;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      const float a = v[i].x - mean, bq = v[i].y - mean, cq = v[i].z - mean, d = v[i].w - mean;
      var += a * a + bq * bq + cq * cq + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(var) / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      const float4 ww = reinterpret_cast<const float4*>(w)[q], bb = reinterpret_cast<const float4*>(b)[q];
      const float4 o = make_float4((v[i].x - mean) * rstd * ww.x + bb.x, (v[i].y - mean) * rstd * ww.y + bb.y,
                                   (v[i].z - mean) * rstd * ww.z + bb.z, (v[i].w - mean) * rstd * ww.w + bb.w);
      if constexpr (SPLIT)
        store_split4<0>(reinterpret_cast<bf16_t*>(y) + (size_t)row * 3 * dim, dim, q * 4, o);
      else
        reinterpret_cast<float4*>(y + (size_t)row * dim)[q] = o;
    }
  }
}

__global__ __launch_bounds__(256) void gelu_f32_kernel(float4* __restrict__ h, long long nq) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
    float4 v = h[q];
    v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f));  // nn.GELU(): exact erf form
    v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
    v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f));
    v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
    h[q] = v;
  }
}

// x += gamma * y  (timm Block: x = x + ls(f(norm(x))))
__global__ __launch_bounds__(256) void resid_f32_kernel(float4* __restrict__ x, const float4* __restrict__ y,
                                                        const float4* __restrict__ gamma, long long nq, int dq) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
    const float4 g = gamma[q % dq], v = y[q];
    float4 o = x[q];
    o.x += g.x * v.x;
    o.y += g.y * v.y;
    o.z += g.z * v.z;
    o.w += g.w * v.w;
    x[q] = o;
  }
}

// ---- attention, head_dim 64, fp32 ------------------------------------------------------------------
// One workgroup = 128 queries of one (image, head); 4 waves x 32 queries; 32-key tiles of K and V in LDS.
// MFMA 32x32x2 (A: lane l holds A[i = l & 31][k = l >> 5], B: B[k = l >> 5][j = l & 31], C/D: col = l & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)):
//   S^T[key][q]  = sum_d K[key][d] * Q[q][d]        A = K tile (LDS), B = Q (registers, pre-scaled by 1/8)
//   O^T[d][q]   += sum_key V[key][d] * P[key][q]    B = P = the S^T accumulator itself: MFMA step r contracts
//                                                   the two keys kappa(r) + 4h (h = l >> 5) held in register r,
//                                                   A = V^T read from the V tile at exactly those keys
constexpr int FA_Q = 128, FA_K = 32, FA_LD = 65;  // odd pitch: the 32 rows of a K-fragment read hit 32 banks

__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int heads, int s_pad, int n_valid) {
  // two K / V tile buffers: tile kt+1 is fetched into registers before tile kt is multiplied and parked in the other
  // buffer afterwards -- one barrier per tile, the global latency hides behind the 96 MFMAs (round 3; the first version
  // staged and multiplied in sequence with two barriers per tile: 4.42 ms per 64 views)
  __shared__ float Ks[2][FA_K * FA_LD];
  __shared__ float Vs[2][FA_K * FA_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h2 = lane >> 5;
  const int nqb = (s_pad + FA_Q - 1) / FA_Q;  // the last block of an image may hang over its rows (s_pad % 32 == 0: whole waves)
  const int id = blockIdx.x;
  const int qb = id % nqb, hd = (id / nqb) % heads, b = id / (nqb * heads);
  const int dim = heads * 64, ld = 3 * dim;
  const size_t row0 = (size_t)b * s_pad;
  const int qrow = qb * FA_Q + wave * 32 + j;
  const float LOG2E = 1.4426950408889634f;
  // Q fragment values of this lane: Q[q][d = 2 s + h2], scaled by head_dim^-0.5 and log2(e) (softmax in base 2)
  float qf[32];
  {
    const float* qp = qkv + (row0 + qrow) * ld + hd * 64 + h2;
#pragma unroll
    for (int s = 0; s < 32; ++s) qf[s] = qp[2 * s] * (0.125f * LOG2E);
  }
  const float* kbase = qkv + row0 * ld + dim + hd * 64;
  const float* vbase = qkv + row0 * ld + 2 * dim + hd * 64;
  floatx16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    o0[r] = 0.f;
    o1[r] = 0.f;
  }
  float m_run = -1e30f, l_run = 0.f;
  const int ntiles = (n_valid + FA_K - 1) / FA_K;
  // staging: 32 keys x 64 d per operand = 512 float4, two per thread
  const int key0 = tid >> 4, dq0 = tid & 15;  // second float4: key0 + 16
  float4 kr[2], vr[2];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const size_t g = (size_t)(kt * FA_K + key0 + 16 * it) * ld + dq0 * 4;
      kr[it] = *reinterpret_cast<const float4*>(kbase + g);
      vr[it] = *reinterpret_cast<const float4*>(vbase + g);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float* kd = Ks[buf] + (key0 + 16 * it) * FA_LD + dq0 * 4;
      float* vd = Vs[buf] + (key0 + 16 * it) * FA_LD + dq0 * 4;
      kd[0] = kr[it].x; kd[1] = kr[it].y; kd[2] = kr[it].z; kd[3] = kr[it].w;
      vd[0] = vr[it].x; vd[1] = vr[it].y; vd[2] = vr[it].z; vd[3] = vr[it].w;
    }
  };
  fetch(0);
  park(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < ntiles;
    if (more) fetch(kt + 1);
    const float* K = Ks[cur];
    const float* V = Vs[cur];
    floatx16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 32; ++t)  // d = 2 t + h2
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(K[j * FA_LD + 2 * t + h2], qf[t], s, 0, 0, 0);
    // s[r] = log2(e) * S^T[key = kappa(r) + 4 h2][query j]
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * FA_K + (r & 3) + 8 * (r >> 2) + 4 * h2;
      if (key >= n_valid) s[r] = -1e30f;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // v_exp_f32: 1 ulp
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      psum += s[r];
    }
    l_run = l_run * alpha + psum;
    if (!__all(m_new == m_run)) {  // wave-uniform: the running max rarely moves after the first tiles (alpha == 1 exactly)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] *= alpha;
        o1[r] *= alpha;
      }
    }
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // contract keys kappa(r) + 4 h2: A = V[key][d = j (+32)], B = P register r
      const int key = (r & 3) + 8 * (r >> 2) + 4 * h2;
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(V[key * FA_LD + j], s[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(V[key * FA_LD + 32 + j], s[r], o1, 0, 0, 0);
    }
    if (more) park(cur ^ 1);  // that buffer was last read in iteration kt - 1, a barrier ago
    __syncthreads();
  }
  if (qb * FA_Q + wave * 32 >= s_pad) return;  // a wave behind the image's rows (its "queries" were the next image's): nothing to store
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
  // o0[r] = O^T[d = kappa(r) + 4 h2][query j], o1: d + 32
  float* op = out + (row0 + qrow) * dim + hd * 64;
#pragma unroll
  for (int r = 0; r < 16; r += 4) {  // kappa(r .. r+3) are 4 consecutive d: one float4 per group
    const int d = 8 * (r >> 2) + 4 * h2;
    *reinterpret_cast<float4*>(op + d) = make_float4(o0[r] * inv, o0[r + 1] * inv, o0[r + 2] * inv, o0[r + 3] * inv);
    *reinterpret_cast<float4*>(op + 32 + d) = make_float4(o1[r] * inv, o1[r + 1] * inv, o1[r + 2] * inv, o1[r + 3] * inv);
  }
}

}  // namespace

extern "C" int64_t dvt_vit_workspace_bytes_f32(const DvtVitConfig* c, int batch) {
  if (!c || batch <= 0 || c->s_pad % 32 || c->dim % 64 || c->heads * 64 != c->dim) return -1;
  return carve_f32(c, batch, nullptr, nullptr);
}

int g_f32x3_exact_attention = 0;  // dvt_tune_set(1, -520 / -521): exact-fp32 attention inside the bf16x3 forward on / off
int g_f32x3_unfused = 0;          // dvt_tune_set(1, -522 / -523): split kernels instead of the split epilogues on / off

namespace {
struct VitWorkX3 {
  float *x, *qkv, *ao, *hid, *col;
  bf16_t *a3, *h3, *sc;  // split rows of the current GEMM input; split GELU(hidden) [T, 3 mlp]; attention scratch
};
inline int64_t rows256(const DvtVitConfig* c, int batch) { return ((int64_t)batch * c->s_pad + 255) / 256 * 256; }
int64_t carve_x3(const DvtVitConfig* c, int batch, char* base, VitWorkX3* w) {
  const int64_t T = rows256(c, batch);  // whole 256-row GEMM tiles; rows beyond batch * s_pad are computed and never read
  int64_t o = 0;
  auto take = [&](int64_t bytes) {
    char* p = base ? base + o : nullptr;
    o += up256b(bytes);
    return p;
  };
  VitWorkX3 t;
  t.x = (float*)take(T * c->dim * 4);
  t.qkv = (float*)take(T * 3 * c->dim * 4);
  t.ao = (float*)take(T * c->dim * 4);
  t.hid = (float*)take(T * c->mlp_dim * 4);
  t.col = (float*)take(T * c->k_patch * 4);
  const int64_t kmax = c->mlp_dim > c->k_patch ? c->mlp_dim : c->k_patch;
  t.a3 = (bf16_t*)take(T * 3 * kmax * 2);
  t.h3 = (bf16_t*)take(T * 3 * c->mlp_dim * 2);
  t.sc = (bf16_t*)take(dvt_vit_attention_x3_scratch_bytes(batch, c->heads, c->s_pad));
  if (w) *w = t;
  return o;
}
}  // namespace

extern "C" int64_t dvt_vit_workspace_bytes_f32x3(const DvtVitConfig* c, int batch) {
  if (!c || batch <= 0 || c->s_pad % 128 || c->dim % 128 || c->heads * 64 != c->dim || c->k_patch % 64) return -1;
  return carve_x3(c, batch, nullptr, nullptr);
}

extern "C" int dvt_vit_split3(const float* x, void* out3, long long rows, int k, int weights, int gelu, void* stream) {
  if (!x || !out3 || rows < 0 || k <= 0 || k % 4) return DVT_E_BADARG;
  if (rows == 0) return 0;
  const long long nq = rows * (k / 4);
  const int blocks = (int)(nq / 256 + 1 < 256 * 16 ? nq / 256 + 1 : 256 * 16);
  hipStream_t s = (hipStream_t)stream;
  if (weights)
    hipLaunchKernelGGL((split3_kernel<1, false>), dim3(blocks), dim3(256), 0, s, (const float4*)x, (bf16_t*)out3, nq, k / 4);
  else if (gelu)
    hipLaunchKernelGGL((split3_kernel<0, true>), dim3(blocks), dim3(256), 0, s, (const float4*)x, (bf16_t*)out3, nq, k / 4);
  else
    hipLaunchKernelGGL((split3_kernel<0, false>), dim3(blocks), dim3(256), 0, s, (const float4*)x, (bf16_t*)out3, nq, k / 4);
  DVT_CHECK_LAUNCH();
  return 0;
}

// fp32 in, fp32 out linear layer through the bf16x3 operands: y[m, n] = x[m, k] . W^T + b;  w3 = dvt_vit_split3(W, weights = 1)
extern "C" int dvt_vit_linear_f32x3(const float* x, const void* w3, const float* b, float* y, void* scratch3, int m, int n,
                                    int k, void* stream) {
  if (!x || !w3 || !y || !scratch3) return DVT_E_BADARG;
  int rc = dvt_vit_split3(x, scratch3, m, k, 0, 0, stream);
  if (rc) return rc;
  return dvt_vit_gemm_f32out(scratch3, w3, b, y, m, n, 3 * k, stream);
}

// The fp32 forward with every linear layer as ONE bf16 GEMM over the K-concatenated split operands (3 x the bf16 flops
// on the 2.5 PF/s pipe instead of the 157 TF/s fp32 one); LayerNorm, attention (fp32 MFMA), GELU, residual stream fp32
// as in dvt_vit_forward_f32.  `h_w`: matrices = bf16 [out, 3 * in] from dvt_vit_split3(weights = 1); vectors fp32.
extern "C" int dvt_vit_forward_f32x3(const DvtVitConfig* c, const DvtVitWeights* w, const float* img, float* feat,
                                     int batch, int n_blocks, void* workspace, void* stream) {
  if (!c || !w || !img || !feat || !workspace || batch <= 0 || n_blocks < 0 || n_blocks > c->depth)
    return DVT_E_BADARG;
  if (c->s_pad % 128 || c->dim % 128 || c->heads * 64 != c->dim || c->k_patch % 64 || c->mlp_dim % 128) return DVT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  VitWorkX3 k;
  carve_x3(c, batch, (char*)workspace, &k);
  const int T = batch * c->s_pad, D = c->dim, Tg = (int)rows256(c, batch);
#define DVT_TRY(x)         \
  do {                     \
    int rc__ = (x);        \
    if (rc__) return rc__; \
  } while (0)
  hipLaunchKernelGGL(im2col_f32_kernel, dim3(T), dim3(256), 0, s, img, k.col, *c);
  DVT_CHECK_LAUNCH();
  DVT_TRY(dvt_vit_split3(k.col, k.a3, T, c->k_patch, 0, 0, stream));
  DVT_TRY(dvt_vit_gemm_f32out(k.a3, w->patch_w, w->patch_b, k.ao, Tg, D, 3 * c->k_patch, stream));
  hipLaunchKernelGGL(embed_f32_kernel, dim3(T), dim3(256), 0, s, (const float4*)k.ao, (float4*)k.x,
                     (const float4*)w->cls_token, (const float4*)w->pos_embed, *c);
  DVT_CHECK_LAUNCH();
  // The GEMMs run over Tg = whole 256-row tiles, embed / LayerNorm write T rows: the phantom rows of the residual stream
  // and of the split A rows start every forward as zeros (they are only ever read row-locally, but the carve depends on
  // `batch`, so without this they could alias another layout's bytes, NaN patterns included -- ADVICE r3)
  if (Tg > T) {
    if (hipMemsetAsync(k.x + (size_t)T * D, 0, (size_t)(Tg - T) * D * sizeof(float), s) != hipSuccess) return DVT_E_BADARG;
    if (hipMemsetAsync((char*)k.a3 + (size_t)T * 3 * D * 2, 0, (size_t)(Tg - T) * 3 * D * 2, s) != hipSuccess)
      return DVT_E_BADARG;
  }
  for (int l = 0; l < n_blocks; ++l) {
    const DvtVitBlockWeights& bw = w->blocks[l];
    hipLaunchKernelGGL((layernorm_f32_kernel<false, true>), dim3(dvt_cdiv(T, 4)), dim3(256), 0, s, k.x, bw.norm1_w,
                       bw.norm1_b, (float*)k.a3, T, D, c->ln_eps, 0, 0, 0);
    DVT_CHECK_LAUNCH();
    // dvt_tune_set(1, -520): exact-fp32 attention; -522: unfused epilogues (fp32 qkv / hidden + split kernels), for A/B
    if (g_f32x3_exact_attention) {
      DVT_TRY(dvt_vit_gemm_f32out(k.a3, bw.qkv_w, bw.qkv_b, k.qkv, Tg, 3 * D, 3 * D, stream));
      DVT_TRY(dvt_vit_attention_f32(k.qkv, k.ao, batch, c->heads, c->s_pad, c->n_tokens, s));
    } else if (g_f32x3_unfused) {
      DVT_TRY(dvt_vit_gemm_f32out(k.a3, bw.qkv_w, bw.qkv_b, k.qkv, Tg, 3 * D, 3 * D, stream));
      DVT_TRY(dvt_vit_attention_x3(k.qkv, k.ao, k.sc, batch, c->heads, c->s_pad, c->n_tokens, stream));
    } else {  // q | k | V^T leave the qkv GEMM already split
      DVT_TRY(dvt_vit_gemm_qkv_x3(k.a3, bw.qkv_w, bw.qkv_b, k.sc, Tg, D, c->heads, c->s_pad, batch, 3 * D, stream));
      // ... and the attention output leaves as the proj GEMM's split A rows (a3 is free: the qkv GEMM has read it)
      DVT_TRY(dvt_vit_attention_x3_presplit(k.sc, k.a3, batch, c->heads, c->s_pad, c->n_tokens, 1, stream));
    }
    if (g_f32x3_exact_attention || g_f32x3_unfused) DVT_TRY(dvt_vit_split3(k.ao, k.a3, T, D, 0, 0, stream));
    DVT_TRY(dvt_vit_gemm_residual(k.a3, bw.proj_w, bw.proj_b, bw.ls1, k.x, Tg, D, 3 * D, stream));
    hipLaunchKernelGGL((layernorm_f32_kernel<false, true>), dim3(dvt_cdiv(T, 4)), dim3(256), 0, s, k.x, bw.norm2_w,
                       bw.norm2_b, (float*)k.a3, T, D, c->ln_eps, 0, 0, 0);
    DVT_CHECK_LAUNCH();
    if (g_f32x3_unfused) {
      DVT_TRY(dvt_vit_gemm_f32out(k.a3, bw.fc1_w, bw.fc1_b, k.hid, Tg, c->mlp_dim, 3 * D, stream));
      DVT_TRY(dvt_vit_split3(k.hid, k.h3, T, c->mlp_dim, 0, 1, stream));
    } else {  // GELU + split in the fc1 epilogue
      DVT_TRY(dvt_vit_gemm_gelu_x3(k.a3, bw.fc1_w, bw.fc1_b, k.h3, Tg, c->mlp_dim, 3 * D, stream));
    }
    DVT_TRY(dvt_vit_gemm_residual(k.h3, bw.fc2_w, bw.fc2_b, bw.ls2, k.x, Tg, D, 3 * c->mlp_dim, stream));
  }
#undef DVT_TRY
  const int out_rows = batch * (c->n_tokens - c->n_prefix);
  hipLaunchKernelGGL(layernorm_f32_kernel<true>, dim3(dvt_cdiv(out_rows, 4)), dim3(256), 0, s, k.x, w->norm_w,
                     w->norm_b, feat, out_rows, D, c->ln_eps, c->s_pad, c->n_tokens, c->n_prefix);
  DVT_CHECK_LAUNCH();
  return 0;
}

extern "C" int dvt_vit_attention_f32(const float* qkv, float* out, int batch, int heads, int s_pad, int n_valid,
                                     void* stream) {
  if (!qkv || !out || batch <= 0 || heads <= 0 || s_pad % 32 || n_valid <= 0 || n_valid > s_pad)
    return DVT_E_BADARG;
  hipLaunchKernelGGL(attention_f32_kernel, dim3(((s_pad + FA_Q - 1) / FA_Q) * heads * batch), dim3(256), 0, (hipStream_t)stream,
                     qkv, out, heads, s_pad, n_valid);
  DVT_CHECK_LAUNCH();
  return 0;
}

extern "C" int dvt_vit_forward_f32(const DvtVitConfig* c, const DvtVitWeights* w, const float* img, float* feat,
                                   int batch, int n_blocks, void* workspace, void* stream) {
  if (!c || !w || !img || !feat || !workspace || batch <= 0 || n_blocks < 0 || n_blocks > c->depth)
    return DVT_E_BADARG;
  if (c->s_pad % 32 || c->dim % 64 || c->heads * 64 != c->dim || c->k_patch % 4) return DVT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  VitWorkF k;
  carve_f32(c, batch, (char*)workspace, &k);
  const int T = batch * c->s_pad, D = c->dim;
  const int Tg = (int)rows128(c, batch);  // the linear layers' M (whole 128-row tiles; rows [T, Tg) are never read)
  const long long nqD = (long long)T * D / 4;
  const int ew_blocks = 256 * 8;
#define DVT_TRY(x)         \
  do {                     \
    int rc__ = (x);        \
    if (rc__) return rc__; \
  } while (0)
  hipLaunchKernelGGL(im2col_f32_kernel, dim3(T), dim3(256), 0, s, img, k.col, *c);
  DVT_CHECK_LAUNCH();
  DVT_TRY(dvt_linear_fwd_big(k.col, (const float*)w->patch_w, w->patch_b, k.tmp, Tg, D, c->k_patch, s));
  hipLaunchKernelGGL(embed_f32_kernel, dim3(T), dim3(256), 0, s, (const float4*)k.tmp, (float4*)k.x,
                     (const float4*)w->cls_token, (const float4*)w->pos_embed, *c);
  DVT_CHECK_LAUNCH();
  for (int l = 0; l < n_blocks; ++l) {
    const DvtVitBlockWeights& bw = w->blocks[l];
    hipLaunchKernelGGL(layernorm_f32_kernel<false>, dim3(dvt_cdiv(T, 4)), dim3(256), 0, s, k.x, bw.norm1_w,
                       bw.norm1_b, k.xn, T, D, c->ln_eps, 0, 0, 0);
    DVT_CHECK_LAUNCH();
    DVT_TRY(dvt_linear_fwd_big(k.xn, (const float*)bw.qkv_w, bw.qkv_b, k.qkv, Tg, 3 * D, D, s));
    DVT_TRY(dvt_vit_attention_f32(k.qkv, k.ao, batch, c->heads, c->s_pad, c->n_tokens, s));
    // round 5: where the 128 x 128 GEMM kernel takes the shapes, LayerScale + residual and the exact-erf GELU are its epilogues
    // (three passes over fp32 [T, dim] / [T, mlp_dim] tensors per block less); otherwise the separate kernels as before
    const bool fuse = dvt_linear_big_ok(Tg, D, D) && dvt_linear_big_ok(Tg, c->mlp_dim, D) && dvt_linear_big_ok(Tg, D, c->mlp_dim);
    if (fuse) {
      DVT_TRY(dvt_linear_fwd_big_epi(k.ao, (const float*)bw.proj_w, bw.proj_b, k.x, Tg, D, D, 2, bw.ls1, s));
    } else {
      DVT_TRY(dvt_linear_fwd_big(k.ao, (const float*)bw.proj_w, bw.proj_b, k.tmp, Tg, D, D, s));
      hipLaunchKernelGGL(resid_f32_kernel, dim3(ew_blocks), dim3(256), 0, s, (float4*)k.x, (const float4*)k.tmp,
                         (const float4*)bw.ls1, nqD, D / 4);
      DVT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(layernorm_f32_kernel<false>, dim3(dvt_cdiv(T, 4)), dim3(256), 0, s, k.x, bw.norm2_w,
                       bw.norm2_b, k.xn, T, D, c->ln_eps, 0, 0, 0);
    DVT_CHECK_LAUNCH();
    if (fuse) {
      DVT_TRY(dvt_linear_fwd_big_epi(k.xn, (const float*)bw.fc1_w, bw.fc1_b, k.hid, Tg, c->mlp_dim, D, 1, nullptr, s));
      DVT_TRY(dvt_linear_fwd_big_epi(k.hid, (const float*)bw.fc2_w, bw.fc2_b, k.x, Tg, D, c->mlp_dim, 2, bw.ls2, s));
    } else {
      DVT_TRY(dvt_linear_fwd_big(k.xn, (const float*)bw.fc1_w, bw.fc1_b, k.hid, Tg, c->mlp_dim, D, s));
      hipLaunchKernelGGL(gelu_f32_kernel, dim3(ew_blocks), dim3(256), 0, s, (float4*)k.hid,
                         (long long)T * c->mlp_dim / 4);
      DVT_CHECK_LAUNCH();
      DVT_TRY(dvt_linear_fwd_big(k.hid, (const float*)bw.fc2_w, bw.fc2_b, k.tmp, Tg, D, c->mlp_dim, s));
      hipLaunchKernelGGL(resid_f32_kernel, dim3(ew_blocks), dim3(256), 0, s, (float4*)k.x, (const float4*)k.tmp,
                         (const float4*)bw.ls2, nqD, D / 4);
      DVT_CHECK_LAUNCH();
    }
  }
#undef DVT_TRY
  const int out_rows = batch * (c->n_tokens - c->n_prefix);
  hipLaunchKernelGGL(layernorm_f32_kernel<true>, dim3(dvt_cdiv(out_rows, 4)), dim3(256), 0, s, k.x, w->norm_w,
                     w->norm_b, feat, out_rows, D, c->ln_eps, c->s_pad, c->n_tokens, c->n_prefix);
  DVT_CHECK_LAUNCH();
  return 0;
}
