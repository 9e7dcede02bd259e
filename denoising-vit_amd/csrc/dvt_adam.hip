// Fused dense Adam + zero_grad over the flat parameter arena (HBM-bandwidth bound).
//
// Reference: main_img_denoising.py:48-54 `torch.optim.Adam(params, lr, eps=1e-15,
// weight_decay=1e-5, betas=(0.9, 0.99))`, :87 `optimizer.zero_grad()`, :89 `optimizer.step()`.
// torch semantics restated (torch/optim/adam.py, _multi_tensor_adam, capturable=False):
//   g   = grad + weight_decay * p
//   m   = m + (g - m) * (1 - beta1)                       (lerp)
//   v   = v * beta2 + (1 - beta2) * g * g                 (mul_, addcmul_)
//   den = sqrt(v) / sqrt(1 - beta2^t) + eps
//   p   = p + (-(lr / (1 - beta1^t))) * (m / den)         (addcdiv_)
// Dense: EVERY parameter is stepped every iteration, also grid entries whose data gradient
// is zero (tcnn returns a dense dL/dparams).  Each tensor group carries
// its own step count t (quirk Q3) through DvtAdamSeg.
//
// Traffic: p, m, v are read and written once = 24 B/param.  The hash-grid gradient is sparse
// (<= 2048*16*4 of 2.47 M entries per step), so instead of streaming a dense gradient
// (+4 B read, +4 B clear per param) a 1-bit-per-entry `touched` bitmap written by
// dvt_grid_bwd gates both the gradient load and its clearing: one 32-bit word covers
// 32 entries = 256 floats = exactly the 64 float4 one wave processes per iteration.
#include "dvt_common.h"

namespace {

// Experiment knob (default off): zero-WRITE the whole sparse gradient region every step, hoping
// to keep the fine levels' gradient lines cache-resident for the next step's atomics.
// Measured: no effect on the grid backward (189.8 vs 189.4 us) and +30 us on Adam -> off.
int g_adam_zero_all = 0;

struct AdamKArgs {
  int zero_all;
  float one_m_b1, beta2, one_m_b2, eps, wd;
  long long q_sparse_end;  // float4 index
  int n_segs;
  long long q_begin[DVT_ADAM_MAX_SEGS], q_end[DVT_ADAM_MAX_SEGS];
  float neg_step[DVT_ADAM_MAX_SEGS];  // -(lr / bias_correction1)
  float bc2s[DVT_ADAM_MAX_SEGS];      // sqrt(bias_correction2)
};

__device__ __forceinline__ void adam1(float& p, float& m, float& v, float g, float wd,
                                      float one_m_b1, float b2, float one_m_b2, float bc2s,
                                      float eps, float neg_step) {
  g = g + wd * p;
  m = m + (g - m) * one_m_b1;
  v = v * b2 + (one_m_b2 * g) * g;
  const float den = sqrtf(v) / bc2s + eps;
  p = p + neg_step * (m / den);
}

// Each wave handles 64 consecutive float4 (256 floats, one bitmap word) per iteration.
__global__ __launch_bounds__(256) void adam_kernel(AdamKArgs a, float4* __restrict__ P,
                                                   float4* __restrict__ M, float4* __restrict__ V,
                                                   float4* __restrict__ G,
                                                   uint32_t* __restrict__ touched, int seg,
                                                   long long n_chunks) {
  const int lane = threadIdx.x & 63;
  const long long wave_global = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long wave_stride = (long long)gridDim.x * 4;
  const float one_m_b1 = a.one_m_b1, one_m_b2 = a.one_m_b2;
  const float neg_step = a.neg_step[seg], bc2s = a.bc2s[seg];
  const long long qb = a.q_begin[seg];
  for (long long ch = wave_global; ch < n_chunks; ch += wave_stride) {
    const long long q0 = qb + ch * 64;
    const long long q = q0 + lane;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    bool has = true;
    uint32_t word = 0;
    const bool sparse = q0 < a.q_sparse_end;
    if (sparse) {
      word = touched[q0 >> 6];
      has = (word >> (lane >> 1)) & 1u;
    }
    float4 p = P[q], m = M[q], v = V[q];
    if (has) g = G[q];
    adam1(p.x, m.x, v.x, g.x, a.wd, one_m_b1, a.beta2, one_m_b2, bc2s, a.eps, neg_step);
    adam1(p.y, m.y, v.y, g.y, a.wd, one_m_b1, a.beta2, one_m_b2, bc2s, a.eps, neg_step);
    adam1(p.z, m.z, v.z, g.z, a.wd, one_m_b1, a.beta2, one_m_b2, bc2s, a.eps, neg_step);
    adam1(p.w, m.w, v.w, g.w, a.wd, one_m_b1, a.beta2, one_m_b2, bc2s, a.eps, neg_step);
    P[q] = p;
    M[q] = m;
    V[q] = v;
    if (has || (sparse && a.zero_all)) G[q] = make_float4(0.f, 0.f, 0.f, 0.f);  // zero_grad
    if (sparse && word != 0u && lane == 0) touched[q0 >> 6] = 0u;
  }
}

// ---- exact lazy Adam for the sparse (hash-grid) region -----------------------------------
// One wave per bitmap word (32 entries = 64 float4, the same lane <-> float4 mapping as the
// dense kernel).  MODE 0 (catch-up, before the forward pass): entries marked in the bitmap are
// replayed from last_step[e] to `t` with a zero data gradient (g = wd * p), i.e. exactly the
// steps the dense sweep would have applied to them.  MODE 1 (after the backward pass): step t
// with the accumulated gradient, zero_grad, last_step = t + 1, clear the word.  MODE 2 =
// MODE 0 + clear the word (used outside the loop).  ALL != 0: ignore the bitmap, every entry.
struct LazyArgs {
  float one_m_b1, beta2, one_m_b2, eps, wd;
  int t;                    // MODE 0/2: replay through step t-1; MODE 1: apply step t
  float neg_step, bc2s;     // MODE 1 scalars of step t
  const float* table;       // [num_iters][2] = {neg_step_s, bc2s_s}
  long long n_words;
};

template <int MODE, int ALL>
__global__ __launch_bounds__(256) void adam_lazy_kernel(LazyArgs a, float4* __restrict__ P,
                                                        float4* __restrict__ M, float4* __restrict__ V,
                                                        float4* __restrict__ G,
                                                        uint32_t* __restrict__ touched,
                                                        int32_t* __restrict__ last) {
  const int lane = threadIdx.x & 63;
  const long long wave_global = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long wave_stride = (long long)gridDim.x * 4;
  for (long long w = wave_global; w < a.n_words; w += wave_stride) {
    uint32_t word = ALL ? 0xffffffffu : touched[w];
    if (word == 0u) continue;
    const bool has = (word >> (lane >> 1)) & 1u;
    const long long q = w * 64 + lane;
    const long long e = q >> 1;
    if (has) {
      float4 p = P[q], m = M[q], v = V[q];
      if (MODE == 1) {
        const float4 g = G[q];
        adam1(p.x, m.x, v.x, g.x, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, a.bc2s, a.eps, a.neg_step);
        adam1(p.y, m.y, v.y, g.y, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, a.bc2s, a.eps, a.neg_step);
        adam1(p.z, m.z, v.z, g.z, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, a.bc2s, a.eps, a.neg_step);
        adam1(p.w, m.w, v.w, g.w, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, a.bc2s, a.eps, a.neg_step);
        P[q] = p;
        M[q] = m;
        V[q] = v;
        G[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((lane & 1) == 0) last[e] = a.t + 1;
      } else {
        const int from = last[e];
        if (from < a.t) {
          for (int s = from; s < a.t; ++s) {
            const float ns = a.table[2 * s], bc = a.table[2 * s + 1];
            adam1(p.x, m.x, v.x, 0.f, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, bc, a.eps, ns);
            adam1(p.y, m.y, v.y, 0.f, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, bc, a.eps, ns);
            adam1(p.z, m.z, v.z, 0.f, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, bc, a.eps, ns);
            adam1(p.w, m.w, v.w, 0.f, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, bc, a.eps, ns);
          }
          P[q] = p;
          M[q] = m;
          V[q] = v;
          // both lanes of the entry have read `from` above (same wave instruction)
          if ((lane & 1) == 0) last[e] = a.t;
        }
      }
    }
    if (!ALL && (MODE == 1 || MODE == 2) && lane == 0) touched[w] = 0u;
  }
}

}  // namespace

// mode 0: catch-up of marked entries through step t-1; 1: apply step t to marked entries (+clear);
// 2: catch-up + clear; 3: catch-up of ALL entries.
int dvt_adam_lazy(int mode, const DvtAdamArgs* h, const DvtAdamSeg* sg, int t, const float* table,
                  float* p, float* m, float* v, float* g, uint32_t* touched, int32_t* last,
                  long long n_words, hipStream_t stream) {
  if (!h || !p || !m || !v || !g || !touched || !last || !table || n_words <= 0) return DVT_E_BADARG;
  LazyArgs a{};
  a.one_m_b1 = (float)(1.0 - h->beta1);
  a.beta2 = (float)h->beta2;
  a.one_m_b2 = (float)(1.0 - h->beta2);
  a.eps = (float)h->eps;
  a.wd = (float)h->weight_decay;
  a.t = t;
  a.table = table;
  a.n_words = n_words;
  if (sg) {
    a.neg_step = (float)(-(sg->lr / sg->bias_correction1));
    a.bc2s = (float)sg->bias_correction2_sqrt;
  }
  long long blocks = (n_words + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const dim3 grid((unsigned)blocks), block(256);
  // algorithmic bytes of the dense sweep this replaces (24 B/param) are credited to MODE 1 only
  DvtProbeScope probe(DVT_PROBE_ADAM, stream, mode == 1 ? 24.0 * 256.0 * (double)n_words : 0.0);
  switch (mode) {
    case 0: hipLaunchKernelGGL((adam_lazy_kernel<0, 0>), grid, block, 0, stream, a, (float4*)p, (float4*)m, (float4*)v, (float4*)g, touched, last); break;
    case 1: hipLaunchKernelGGL((adam_lazy_kernel<1, 0>), grid, block, 0, stream, a, (float4*)p, (float4*)m, (float4*)v, (float4*)g, touched, last); break;
    case 2: hipLaunchKernelGGL((adam_lazy_kernel<2, 0>), grid, block, 0, stream, a, (float4*)p, (float4*)m, (float4*)v, (float4*)g, touched, last); break;
    case 3: hipLaunchKernelGGL((adam_lazy_kernel<0, 1>), grid, block, 0, stream, a, (float4*)p, (float4*)m, (float4*)v, (float4*)g, touched, last); break;
    default: return DVT_E_BADARG;
  }
  DVT_CHECK_LAUNCH();
  return 0;
}

int dvt_adam_tune(int zero_all) {
  g_adam_zero_all = zero_all;
  return 0;
}

extern "C" int dvt_adam_step(const DvtAdamArgs* h, float* p, float* m, float* v, float* g,
                             uint32_t* touched, void* stream) {
  if (!h || !p || !m || !v || !g || h->n_segs < 0 || h->n_segs > DVT_ADAM_MAX_SEGS)
    return DVT_E_BADARG;
  if ((h->sparse_end & 255) || (h->sparse_end > 0 && !touched)) return DVT_E_BADARG;
  AdamKArgs a{};
  a.zero_all = g_adam_zero_all;
  // torch narrows the python doubles (1 - beta1), beta2, (1 - beta2), eps, wd to fp32 scalars
  a.one_m_b1 = (float)(1.0 - h->beta1);
  a.beta2 = (float)h->beta2;
  a.one_m_b2 = (float)(1.0 - h->beta2);
  a.eps = (float)h->eps;
  a.wd = (float)h->weight_decay;
  a.q_sparse_end = h->sparse_end / 4;
  a.n_segs = h->n_segs;
  for (int s = 0; s < h->n_segs; ++s) {
    const DvtAdamSeg& sg = h->segs[s];
    if ((sg.begin & 255) || (sg.end & 255) || sg.end < sg.begin) return DVT_E_BADARG;
    a.q_begin[s] = sg.begin / 4;
    a.q_end[s] = sg.end / 4;
    a.neg_step[s] = (float)(-(sg.lr / sg.bias_correction1));
    a.bc2s[s] = (float)sg.bias_correction2_sqrt;
  }
  for (int s = 0; s < h->n_segs; ++s) {
    const DvtAdamSeg& sg = h->segs[s];
    if (!sg.active || sg.end == sg.begin) continue;
    const long long n_chunks = (sg.end - sg.begin) / 256;
    // algorithmic bytes: p, m, v read + written once (24 B/param); dense-gradient part +8 B/param
    const double dense_floats = (double)(sg.end - (sg.begin > h->sparse_end ? sg.begin : (sg.end < h->sparse_end ? sg.end : h->sparse_end)));
    DvtProbeScope probe(DVT_PROBE_ADAM, (hipStream_t)stream,
                        24.0 * (double)(sg.end - sg.begin) + 8.0 * dense_floats);
    long long blocks = (n_chunks + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a,
                       (float4*)p, (float4*)m, (float4*)v, (float4*)g, touched, s, n_chunks);
    DVT_CHECK_LAUNCH();
  }
  return 0;
}
