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
int g_adam_nt = 0;  // dvt_tune_set(3, 10 / 11): non-temporal p / m / v streams in the dense sweep off / on

struct AdamKArgs {
  int zero_all;
  int nt;  // stream p / m / v with the non-temporal hint (dvt_tune_set(3, 10 / 11))
  int reverse;  // sweep the chunks from the end: consecutive steps alternate, see dvt_adam_step_k
  float one_m_b1, beta2, one_m_b2, eps, wd;
  long long q_sparse_end;  // float4 index
  int n_segs;
  long long q_begin[DVT_ADAM_MAX_SEGS], q_end[DVT_ADAM_MAX_SEGS];
  float neg_step[DVT_ADAM_MAX_SEGS];  // -(lr / bias_correction1)
  float bc2s[DVT_ADAM_MAX_SEGS];      // sqrt(bias_correction2)
  long long chunk0[DVT_ADAM_MAX_SEGS + 1];  // prefix sum of 256-float chunks over ACTIVE segments
  int seg_of[DVT_ADAM_MAX_SEGS];            // active segment list
  int n_active;
};

__device__ __forceinline__ void adam1(float& p, float& m, float& v, float g, float wd,
                                      float one_m_b1, float b2, float one_m_b2, float bc2s,
                                      float eps, float neg_step) {
  // explicit fused multiply-adds: with -ffp-contract=fast the compiler may contract a*b + c*d either way, and it did
  // so differently in different kernels -- the lazy replay's exact mode must reproduce this function bit for bit
  g = __builtin_fmaf(wd, p, g);
  m = __builtin_fmaf(g - m, one_m_b1, m);
  v = __builtin_fmaf(one_m_b2 * g, g, v * b2);
  const float den = sqrtf(v) / bc2s + eps;
  p = __builtin_fmaf(neg_step, m / den, p);
}

// Each wave handles 64 consecutive float4 (256 floats, one bitmap word) per iteration; all
// active tensor groups (different step counts t) are covered by ONE launch.
struct AdamGather {  // device view of DvtAdamRowGather
  long long q_begin, q_end;  // float4 range of G
  int c, lattice;
  const int32_t* offs[DVT_FIT_BATCH_MAX];
  const uint16_t* perm[DVT_FIT_BATCH_MAX];
  const float4* rows[DVT_FIT_BATCH_MAX];
};

struct AdamPtrs {  // per fit of a batched launch (blockIdx.y)
  float4* P[DVT_FIT_BATCH_MAX];
  float4* M[DVT_FIT_BATCH_MAX];
  float4* V[DVT_FIT_BATCH_MAX];
  float4* G[DVT_FIT_BATCH_MAX];
  uint32_t* touched[DVT_FIT_BATCH_MAX];
};

// <= 48 VGPRs: the extractor's 8-phase GEMM keeps two 232-register waves on every SIMD, which leaves exactly
// 48 registers -- one Adam wave -- per SIMD.  At 52 registers the HBM-bound Adam and the MFMA-bound GEMM could
// only time-slice whole CUs (the pipelined image time was t_extract + 0.96 * t_fit).
// GATHER = false is the pure streaming kernel: 48 VGPRs, which is exactly what the extractor's 8-phase GEMM
// (two 232-register waves per SIMD) leaves free -- one Adam wave per SIMD then shares the CU with it and the
// HBM-bound update overlaps the MFMA-bound GEMM instead of time-slicing whole CUs (at 52 registers, with the
// gather path compiled in, nothing fitted and the pipelined image time was t_extract + 0.96 t_fit).
struct AdamShadow {  // shadow copies (bf16 or fp32: L.f32) of the MLP weights, maintained by the dense sweep itself (SHADOW = true)
  DvtShadowLayout L;
  void* sh[DVT_FIT_BATCH_MAX];
};

typedef float adam_f4 __attribute__((ext_vector_type(4)));  // (the non-temporal builtins take clang vectors, not HIP's float4)

template <bool GATHER, bool SHADOW = false>
__device__ __forceinline__ void adam_dense_body(const AdamKArgs& a, const AdamPtrs& q, const AdamGather& gr, int bx,
                                                int nbx, int fit, const AdamShadow* shw = nullptr) {
  float4* __restrict__ P = q.P[fit];
  float4* __restrict__ M = q.M[fit];
  float4* __restrict__ V = q.V[fit];
  float4* __restrict__ G = q.G[fit];
  uint32_t* __restrict__ touched = q.touched[fit];
  const int lane = threadIdx.x & 63;
  const long long wave_global = (long long)bx * 4 + (threadIdx.x >> 6);
  const long long wave_stride = (long long)nbx * 4;
  const float one_m_b1 = a.one_m_b1, one_m_b2 = a.one_m_b2;
  const long long n_chunks = a.chunk0[a.n_active];
  for (long long ci = wave_global; ci < n_chunks; ci += wave_stride) {
    const long long ch = a.reverse ? n_chunks - 1 - ci : ci;
    int k = 0;
#pragma unroll
    for (int j = 1; j < DVT_ADAM_MAX_SEGS; ++j)
      if (j < a.n_active && ch >= a.chunk0[j]) k = j;
    const int seg = a.seg_of[k];
    const float neg_step = a.neg_step[seg], bc2s = a.bc2s[seg];
    const long long q0 = a.q_begin[seg] + (ch - a.chunk0[k]) * 64;
    const long long q = q0 + lane;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    bool has = true;
    uint32_t word = 0;
    const bool sparse = q0 < a.q_sparse_end;
    if (sparse) {
      word = touched[q0 >> 6];
      has = (word >> (lane >> 1)) & 1u;
    }
    float4 p, m, v;
    if (a.nt) {  // (wave-uniform) every element of the dense part is read once and written once per step
      p = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(&P[q])));
      m = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(&M[q])));
      v = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(&V[q])));
    } else {
      p = P[q];
      m = M[q];
      v = V[q];
    }
    const bool gathered = GATHER && q0 >= gr.q_begin && q0 < gr.q_end;  // wave-uniform: chunk inside G
    if (GATHER && gathered) {
      // dG[row] = sum of the d_pred rows of this step's samples on lattice row `row`
      const int e = (int)(q - gr.q_begin) * 4, row = e / gr.c, col4 = (e - row * gr.c) >> 2;
      const int32_t* offs = gr.offs[fit];
      const uint16_t* perm = gr.perm[fit];
      const float4* rows = gr.rows[fit];
      const int cq = gr.c >> 2;
      const bool real = row < gr.lattice;  // alignment padding behind the last row has no gradient
      for (int o = real ? offs[row] : 0, oe = real ? offs[row + 1] : 0; o < oe; ++o) {
        const float4 d = rows[(size_t)perm[o] * cq + col4];
        g.x += d.x;
        g.y += d.y;
        g.z += d.z;
        g.w += d.w;
      }
      has = false;  // nothing to clear in the dense gradient buffer
    } else if (has) {
      g = G[q];
    }
    adam1(p.x, m.x, v.x, g.x, a.wd, one_m_b1, a.beta2, one_m_b2, bc2s, a.eps, neg_step);
    adam1(p.y, m.y, v.y, g.y, a.wd, one_m_b1, a.beta2, one_m_b2, bc2s, a.eps, neg_step);
    adam1(p.z, m.z, v.z, g.z, a.wd, one_m_b1, a.beta2, one_m_b2, bc2s, a.eps, neg_step);
    adam1(p.w, m.w, v.w, g.w, a.wd, one_m_b1, a.beta2, one_m_b2, bc2s, a.eps, neg_step);
    if (a.nt) {
      __builtin_nontemporal_store(__builtin_bit_cast(adam_f4, p), reinterpret_cast<adam_f4*>(&P[q]));
      __builtin_nontemporal_store(__builtin_bit_cast(adam_f4, m), reinterpret_cast<adam_f4*>(&M[q]));
      __builtin_nontemporal_store(__builtin_bit_cast(adam_f4, v), reinterpret_cast<adam_f4*>(&V[q]));
    } else {
      P[q] = p;
      M[q] = m;
      V[q] = v;
    }
    if constexpr (SHADOW) {  // wave-uniform: the chunk lies inside the shadowed matrices or not
      if (q0 * 4 >= shw->L.lo && q0 * 4 < shw->L.hi) dvt_shadow_store(shw->L, shw->sh[fit], q * 4, p);
    }
    if (has || (sparse && a.zero_all)) G[q] = make_float4(0.f, 0.f, 0.f, 0.f);  // zero_grad
    if (sparse && word != 0u && lane == 0) touched[q0 >> 6] = 0u;
  }
}

template <bool GATHER>
__global__ __launch_bounds__(256) void adam_kernel(AdamKArgs a, AdamPtrs q, AdamGather gr) {
  adam_dense_body<GATHER>(a, q, gr, blockIdx.x, gridDim.x, blockIdx.y);
}

// ==========================================================================================================
// Lazy-exact Adam for the fine hash-grid levels.
//
// 98 % of the arena are grid entries of the levels with >= 64 k entries, and a step touches at most 4 * batch of
// them: an entry of the finest levels is sampled once in ~128 steps.  The reference nevertheless steps EVERY
// parameter EVERY iteration (dense Adam with coupled weight decay: g = wd * p even where the data gradient is
// zero, quirk Q2) -- 515 MB of HBM traffic per step, 55 % of the fit.  But an untouched entry's update depends on
// nothing except its own (p, m, v) and the step's learning rate, so it can be applied LATER, as long as it is
// applied before anyone reads p: per entry a 16-bit `done` counter says how many steps are in; the gradient a
// step wrote stays in the gradient arena as that entry's PENDING step.  Before the row kernel of step t the
// catch-up kernel visits exactly the entries step t will read (their sorted list is known in advance,
// dvt_grid_dev.h): apply the pending step `done` with its gradient, then steps done+1 .. t-1 with g = 0 -- the
// same arithmetic in the same order as the dense sweep -- store, clear the gradient, done = t.  At the end of a
// dvt_fit_run call one sweep brings every entry to the last step, so the arena is exact at every call boundary.
// Work is conserved (every entry-step is still executed once), HBM traffic is not: ~12 MB instead of ~470 MB per
// step; the kernel is VALU-bound.  v_rcp_f32 / v_sqrt_f32 (1 ulp) replace the IEEE division / square root of the
// dense kernel in the replay loop: 2 instead of ~25 instructions per element-step.
// 4 lanes per entry (a feature pair each, packed fp32 arithmetic): the lanes of an entry share the trip count.
// ==========================================================================================================
constexpr int LAZY_TAB = 256;  // most recent steps whose (neg_step, 1 / bc2s) sit in LDS (2 KB); older ones come from L2
                               // (never needed while the refresh interval is <= LAZY_TAB)
constexpr int LAZY_BLOCK = 256;  // one wave per SIMD, < 48 VGPRs, 2 KB of LDS: like the dense Adam kernel a lazy block
                                 // fits beside the extractor's 8-phase GEMM workgroups (456 of 512 VGPRs, 136 of 160 KB)
struct LazyArgs {
  float* P[DVT_FIT_BATCH_MAX];
  float* M[DVT_FIT_BATCH_MAX];
  float* V[DVT_FIT_BATCH_MAX];
  float* G[DVT_FIT_BATCH_MAX];
  uint16_t* done[DVT_FIT_BATCH_MAX];       // [n_entries - e0], steps already applied (relative to the call's first step)
  const uint32_t* ukeys[DVT_FIT_BATCH_MAX];  // catch-up: this step's distinct entries [L][nt] (absolute, ascending)
  const int32_t* ucount[DVT_FIT_BATCH_MAX];  //           [L]
  const float* neg_step;   // [steps of the call]  -(lr / bias_correction1)
  const float* inv_bc2s;   //                      1 / sqrt(bias_correction2)
  const float* bc2s;       //                      sqrt(bias_correction2) (exact mode)
  uint32_t e0, n_entries;  // lazy entries: [e0, n_entries)
  int nt, l0;              // catch-up: list pitch, first level that has lazy entries
  int target;              // bring entries to `target` applied steps
  float one_m_b1, beta2, one_m_b2, eps, wd;
};

// EXACT: the dense kernel's own adam1() (IEEE division and square root): bit-identical to the dense sweep, ~3x the
// instructions.  Default: v_rcp_f32 / v_sqrt_f32 (1 ulp each) and PACKED fp32 arithmetic -- a lane steps two
// features at once (v_pk_mul_f32 / v_pk_fma_f32: one instruction per pair for everything but the two transcendentals).
typedef float lazy_f2 __attribute__((ext_vector_type(2)));
template <bool EXACT>
__device__ __forceinline__ void lazy_replay(const LazyArgs& a, const float* tab_ns, const float* tab_ib, int tab0,
                                            lazy_f2& p, lazy_f2& m, lazy_f2& v, lazy_f2 g, int from, int to) {
  const lazy_f2 wd2 = {a.wd, a.wd}, c1 = {a.one_m_b1, a.one_m_b1}, b2 = {a.beta2, a.beta2}, c2 = {a.one_m_b2, a.one_m_b2},
                eps2 = {a.eps, a.eps};
  for (int s = from; s < to; ++s) {
    const float ns = s >= tab0 ? tab_ns[s - tab0] : a.neg_step[s];
    if (EXACT) {
      const float bc = a.bc2s[s];
      float p0 = p.x, p1 = p.y, m0 = m.x, m1 = m.y, v0 = v.x, v1 = v.y;
      adam1(p0, m0, v0, g.x, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, bc, a.eps, ns);
      adam1(p1, m1, v1, g.y, a.wd, a.one_m_b1, a.beta2, a.one_m_b2, bc, a.eps, ns);
      p = (lazy_f2){p0, p1};
      m = (lazy_f2){m0, m1};
      v = (lazy_f2){v0, v1};
    } else {
      const float ib = s >= tab0 ? tab_ib[s - tab0] : a.inv_bc2s[s];
      const lazy_f2 gg = __builtin_elementwise_fma(wd2, p, g);
      m = __builtin_elementwise_fma(gg - m, c1, m);
      v = __builtin_elementwise_fma(c2 * gg, gg, v * b2);
      const lazy_f2 sq = {__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)};
      const lazy_f2 den = __builtin_elementwise_fma(sq, (lazy_f2){ib, ib}, eps2);
      const lazy_f2 rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
      p = __builtin_elementwise_fma(m * (lazy_f2){ns, ns}, rc, p);
    }
    g = (lazy_f2){0.f, 0.f};
  }
}

// FINAL = false: grid (nt * 4 / LAZY_BLOCK, lazy levels, fits), one entry of this step's list per 4 lanes.
// FINAL = true:  grid (ceil((n_entries - e0) * 4 / LAZY_BLOCK), 1, fits), every lazy entry.
template <bool FINAL, bool EXACT>
__device__ __forceinline__ void adam_lazy_body(const LazyArgs& a, int bx, int by, int fit, float* tab_ns, float* tab_ib) {
  const int tab0 = a.target > LAZY_TAB ? a.target - LAZY_TAB : 0;
  for (int i = threadIdx.x; i < a.target - tab0; i += LAZY_BLOCK) {
    tab_ns[i] = a.neg_step[tab0 + i];
    tab_ib[i] = a.inv_bc2s[tab0 + i];
  }
  __syncthreads();
  const long long i = (long long)bx * LAZY_BLOCK + threadIdx.x;
  const int f = (int)(i & 3);  // feature pair
  uint32_t e;
  if (FINAL) {
    if (i >> 2 >= (long long)(a.n_entries - a.e0)) return;
    e = a.e0 + (uint32_t)(i >> 2);
  } else {
    const int l = a.l0 + by;
    if ((i >> 2) >= a.ucount[fit][l]) return;
    e = a.ukeys[fit][(size_t)l * a.nt + (i >> 2)];
    if (e < a.e0) return;
  }
  const int from = a.done[fit][e - a.e0];
  if (from >= a.target) return;
  const size_t q = (size_t)e * 4 + f;  // float2 index
  lazy_f2* P = reinterpret_cast<lazy_f2*>(a.P[fit]);
  lazy_f2* M = reinterpret_cast<lazy_f2*>(a.M[fit]);
  lazy_f2* V = reinterpret_cast<lazy_f2*>(a.V[fit]);
  lazy_f2* G = reinterpret_cast<lazy_f2*>(a.G[fit]);
  lazy_f2 p = P[q], m = M[q], v = V[q];
  const lazy_f2 g = G[q];
  lazy_replay<EXACT>(a, tab_ns, tab_ib, tab0, p, m, v, g, from, a.target);
  P[q] = p;
  M[q] = m;
  V[q] = v;
  if (g.x != 0.f || g.y != 0.f) G[q] = (lazy_f2){0.f, 0.f};
  if (f == 0) a.done[fit][e - a.e0] = (uint16_t)a.target;
}

template <bool FINAL, bool EXACT>
__global__ __launch_bounds__(LAZY_BLOCK) void adam_lazy_kernel(LazyArgs a) {
  __shared__ float tab_ns[LAZY_TAB], tab_ib[LAZY_TAB];
  adam_lazy_body<FINAL, EXACT>(a, blockIdx.x, blockIdx.y, blockIdx.z, tab_ns, tab_ib);
}

// ONE launch for the step's dense Adam (HBM-bound) and the catch-up of the NEXT step's entries (VALU-bound): they touch
// disjoint parameters, both depend only on the backward pass that just finished, and side by side they take the longer
// of the two instead of the sum.  The first blocks are the (level, list chunk) blocks of the catch-up, the last
// dense_blocks sweep the dense segments.
template <bool EXACT, bool SHADOW>
__device__ __forceinline__ void adam_dense_lazy_body(const AdamKArgs& a, const AdamPtrs& q, const LazyArgs& z, int dense_blocks,
                                                     int lazy_bx, const AdamShadow* shw, float* tab_ns, float* tab_ib) {
  // catch-up blocks FIRST: their replay chains are the long pole, the streaming blocks fill in around them
  const int lazy_blocks = (int)gridDim.x - dense_blocks;
  if ((int)blockIdx.x >= lazy_blocks) {
    AdamGather none{};
    none.q_begin = none.q_end = -1;
    adam_dense_body<false, SHADOW>(a, q, none, (int)blockIdx.x - lazy_blocks, dense_blocks, blockIdx.y, shw);
    return;
  }
  const int b = (int)blockIdx.x;
  adam_lazy_body<false, EXACT>(z, b % lazy_bx, b / lazy_bx, blockIdx.y, tab_ns, tab_ib);
}

// (two kernels, not one with a flag: the shadow layout among the arguments costs registers, and the plain variant must
// stay at 48 VGPRs -- what the extractor's GEMM waves leave free on a SIMD)
template <bool EXACT>
__global__ __launch_bounds__(256) void adam_dense_lazy_kernel(AdamKArgs a, AdamPtrs q, LazyArgs z, int dense_blocks,
                                                              int lazy_bx) {
  __shared__ float tab_ns[LAZY_TAB], tab_ib[LAZY_TAB];
  const int lazy_blocks = (int)gridDim.x - dense_blocks;
  if ((int)blockIdx.x >= lazy_blocks) {
    AdamGather none{};
    none.q_begin = none.q_end = -1;
    adam_dense_body<false>(a, q, none, (int)blockIdx.x - lazy_blocks, dense_blocks, blockIdx.y);
    return;
  }
  const int b = (int)blockIdx.x;
  adam_lazy_body<false, EXACT>(z, b % lazy_bx, b / lazy_bx, blockIdx.y, tab_ns, tab_ib);
}
// ... and with the dense blocks gathering the gradient of G from the row lists (the fp32-operand step of round 4: no fused
// weight-gradient launch writes dG there)
template <bool EXACT>
__global__ __launch_bounds__(256) void adam_dense_lazy_gather_kernel(AdamKArgs a, AdamPtrs q, LazyArgs z, int dense_blocks,
                                                                     int lazy_bx, AdamGather gr) {
  __shared__ float tab_ns[LAZY_TAB], tab_ib[LAZY_TAB];
  const int lazy_blocks = (int)gridDim.x - dense_blocks;
  if ((int)blockIdx.x >= lazy_blocks) {
    adam_dense_body<true>(a, q, gr, (int)blockIdx.x - lazy_blocks, dense_blocks, blockIdx.y);
    return;
  }
  const int b = (int)blockIdx.x;
  adam_lazy_body<false, EXACT>(z, b % lazy_bx, b / lazy_bx, blockIdx.y, tab_ns, tab_ib);
}
template <bool EXACT>
__global__ __launch_bounds__(256) void adam_dense_lazy_shadow_kernel(AdamKArgs a, AdamPtrs q, LazyArgs z, int dense_blocks,
                                                                     int lazy_bx, AdamShadow shw) {
  __shared__ float tab_ns[LAZY_TAB], tab_ib[LAZY_TAB];
  adam_dense_lazy_body<EXACT, true>(a, q, z, dense_blocks, lazy_bx, &shw, tab_ns, tab_ib);
}

}  // namespace

int dvt_adam_tune(int v) {
  if (v == 10 || v == 11) g_adam_nt = v == 11;
  else g_adam_zero_all = v;
  return 0;
}

extern "C" int dvt_adam_step(const DvtAdamArgs* h, float* p, float* m, float* v, float* g,
                             uint32_t* touched, void* stream) {
  return dvt_adam_step_k(h, 1, &p, &m, &v, &g, &touched, (hipStream_t)stream);
}

namespace {
int make_lazy_args(const DvtAdamLazy* z, int k, bool final_sweep, int target, const uint32_t* const* ukeys,
                   const int32_t* const* ucount, LazyArgs* out);
}

int dvt_adam_step_k(const DvtAdamArgs* h, int k, float* const* p, float* const* m, float* const* v,
                    float* const* g, uint32_t* const* touched, hipStream_t stream,
                    const DvtAdamRowGather* gather, int reverse, const DvtAdamLazy* lazy_next, int lazy_target,
                    const uint32_t* const* lazy_ukeys, const int32_t* const* lazy_ucount, const DvtShadowLayout* shadow_L,
                    void* const* shadow) {
  if (!h || k < 1 || k > DVT_FIT_BATCH_MAX || h->n_segs < 0 || h->n_segs > DVT_ADAM_MAX_SEGS)
    return DVT_E_BADARG;
  if (h->sparse_end & 255) return DVT_E_BADARG;
  AdamPtrs q{};
  for (int f = 0; f < k; ++f) {
    if (!p[f] || !m[f] || !v[f] || !g[f] || (h->sparse_end > 0 && !touched[f])) return DVT_E_BADARG;
    q.P[f] = (float4*)p[f];
    q.M[f] = (float4*)m[f];
    q.V[f] = (float4*)v[f];
    q.G[f] = (float4*)g[f];
    q.touched[f] = touched[f];
  }
  AdamGather gr{};
  gr.q_begin = gr.q_end = -1;
  if (gather != nullptr && gather->rows[0] != nullptr) {
    if ((gather->begin & 255) || (gather->end & 255) || gather->c <= 0 || (gather->c & 3))
      return DVT_E_BADARG;
    gr.q_begin = gather->begin / 4;
    gr.q_end = gather->end / 4;
    gr.c = gather->c;
    gr.lattice = gather->lattice;
    for (int f = 0; f < k; ++f) {
      if (!gather->offs[f] || !gather->perm[f] || !gather->rows[f]) return DVT_E_BADARG;
      gr.offs[f] = gather->offs[f];
      gr.perm[f] = gather->perm[f];
      gr.rows[f] = (const float4*)gather->rows[f];
    }
  }
  AdamKArgs a{};
  a.zero_all = g_adam_zero_all;
  a.nt = g_adam_nt;
  // Direction of the sweep over the arena.  p + m + v (258 MB) is a hair larger than the 256-MB memory-side
  // cache: a forward sweep every step evicts each line just before it is needed again (LRU streaming
  // pathology, 0 % hits); alternating the direction lets a step START on the lines the previous step touched
  // LAST.  The update of a chunk does not depend on the order, so results are unchanged.
  a.reverse = reverse ? 1 : 0;
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
  double work = 0.0;
  a.n_active = 0;
  a.chunk0[0] = 0;
  for (int s = 0; s < h->n_segs; ++s) {
    const DvtAdamSeg& sg = h->segs[s];
    if (!sg.active || sg.end == sg.begin) continue;
    a.seg_of[a.n_active] = s;
    a.chunk0[a.n_active + 1] = a.chunk0[a.n_active] + (sg.end - sg.begin) / 256;
    ++a.n_active;
    // algorithmic bytes: p, m, v read + written once (24 B/param); dense-gradient part +8 B/param
    const double dense_floats = (double)(sg.end - (sg.begin > h->sparse_end ? sg.begin : (sg.end < h->sparse_end ? sg.end : h->sparse_end)));
    work += 24.0 * (double)(sg.end - sg.begin) + 8.0 * dense_floats;
  }
  if (a.n_active == 0) return 0;
  const long long n_chunks = a.chunk0[a.n_active];
  long long blocks = (n_chunks + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  {
    DvtProbeScope probe(DVT_PROBE_ADAM, stream, work * k);
    if (lazy_next != nullptr && gr.q_end > gr.q_begin) {  // catch-up of the next step's entries + the G row gather
      if (shadow_L != nullptr && shadow != nullptr) return DVT_E_BADARG;  // (the fused step never gathers)
      LazyArgs z{};
      const int rc = make_lazy_args(lazy_next, k, false, lazy_target, lazy_ukeys, lazy_ucount, &z);
      if (rc) return rc;
      const int lazy_bx = dvt_cdiv((long long)lazy_next->nt * 4, LAZY_BLOCK);
      const unsigned total = (unsigned)blocks + (unsigned)(lazy_bx * (lazy_next->n_levels - lazy_next->l0));
      const dim3 grid(total, k), blk(256);
      if (lazy_next->exact)
        hipLaunchKernelGGL((adam_dense_lazy_gather_kernel<true>), grid, blk, 0, stream, a, q, z, (int)blocks, lazy_bx, gr);
      else
        hipLaunchKernelGGL((adam_dense_lazy_gather_kernel<false>), grid, blk, 0, stream, a, q, z, (int)blocks, lazy_bx, gr);
    } else if (lazy_next != nullptr) {  // + the catch-up of the next step's entries, same launch
      LazyArgs z{};
      const int rc = make_lazy_args(lazy_next, k, false, lazy_target, lazy_ukeys, lazy_ucount, &z);
      if (rc) return rc;
      const int lazy_bx = dvt_cdiv((long long)lazy_next->nt * 4, LAZY_BLOCK);
      const unsigned total = (unsigned)blocks + (unsigned)(lazy_bx * (lazy_next->n_levels - lazy_next->l0));
      AdamShadow shw{};
      const bool with_shadow = shadow_L != nullptr && shadow != nullptr;
      if (with_shadow) {
        shw.L = *shadow_L;
        for (int f = 0; f < k; ++f) shw.sh[f] = shadow[f];
      }
      const dim3 grid(total, k), blk(256);
      if (lazy_next->exact && with_shadow)
        hipLaunchKernelGGL((adam_dense_lazy_shadow_kernel<true>), grid, blk, 0, stream, a, q, z, (int)blocks, lazy_bx, shw);
      else if (lazy_next->exact)
        hipLaunchKernelGGL((adam_dense_lazy_kernel<true>), grid, blk, 0, stream, a, q, z, (int)blocks, lazy_bx);
      else if (with_shadow)
        hipLaunchKernelGGL((adam_dense_lazy_shadow_kernel<false>), grid, blk, 0, stream, a, q, z, (int)blocks, lazy_bx, shw);
      else
        hipLaunchKernelGGL((adam_dense_lazy_kernel<false>), grid, blk, 0, stream, a, q, z, (int)blocks, lazy_bx);
    } else if (gr.q_end > gr.q_begin)
      hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks, k), dim3(256), 0, stream, a, q, gr);
    else
      hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks, k), dim3(256), 0, stream, a, q, gr);
    DVT_CHECK_LAUNCH();
  }
  return 0;
}

// ---- lazy-exact Adam: host side (see the kernel comment above) ----------------------------------------------
namespace {
int make_lazy_args(const DvtAdamLazy* z, int k, bool final_sweep, int target, const uint32_t* const* ukeys,
                   const int32_t* const* ucount, LazyArgs* out) {
  if (!z || k < 1 || k > DVT_FIT_BATCH_MAX || target < 1 || target > 65535 || z->n_entries <= z->e0) return DVT_E_BADARG;
  LazyArgs a{};
  for (int f = 0; f < k; ++f) {
    a.P[f] = z->p[f];
    a.M[f] = z->m[f];
    a.V[f] = z->v[f];
    a.G[f] = z->g[f];
    a.done[f] = z->done[f];
    if (!a.P[f] || !a.M[f] || !a.V[f] || !a.G[f] || !a.done[f]) return DVT_E_BADARG;
    if (!final_sweep) {
      if (!ukeys || !ucount || !ukeys[f] || !ucount[f]) return DVT_E_BADARG;
      a.ukeys[f] = ukeys[f];
      a.ucount[f] = ucount[f];
    }
  }
  a.neg_step = z->neg_step;
  a.inv_bc2s = z->inv_bc2s;
  a.bc2s = z->bc2s;
  a.e0 = z->e0;
  a.n_entries = z->n_entries;
  a.nt = z->nt;
  a.l0 = z->l0;
  a.target = target;
  a.one_m_b1 = (float)(1.0 - z->beta1);
  a.beta2 = (float)z->beta2;
  a.one_m_b2 = (float)(1.0 - z->beta2);
  a.eps = (float)z->eps;
  a.wd = (float)z->weight_decay;
  *out = a;
  return 0;
}
}  // namespace

int dvt_adam_lazy_k(const DvtAdamLazy* z, int k, bool final_sweep, int target, const uint32_t* const* ukeys,
                    const int32_t* const* ucount, hipStream_t s) {
  LazyArgs a{};
  {
    const int rc = make_lazy_args(z, k, final_sweep, target, ukeys, ucount, &a);
    if (rc) return rc;
  }
  const dim3 blk(LAZY_BLOCK);
  if (final_sweep) {
    const dim3 grid((unsigned)dvt_cdiv((long long)(z->n_entries - z->e0) * 4, LAZY_BLOCK), 1, k);
    if (z->exact)
      hipLaunchKernelGGL((adam_lazy_kernel<true, true>), grid, blk, 0, s, a);
    else
      hipLaunchKernelGGL((adam_lazy_kernel<true, false>), grid, blk, 0, s, a);
  } else {
    const dim3 grid((unsigned)dvt_cdiv((long long)z->nt * 4, LAZY_BLOCK), z->n_levels - z->l0, k);
    if (z->exact)
      hipLaunchKernelGGL((adam_lazy_kernel<false, true>), grid, blk, 0, s, a);
    else
      hipLaunchKernelGGL((adam_lazy_kernel<false, false>), grid, blk, 0, s, a);
  }
  DVT_CHECK_LAUNCH();
  return 0;
}

// Device tables of the per-step scalars of the lazy kernels, written from kernel arguments (stream-ordered, no
// host buffer has to outlive the call): relative step i of the call <-> Adam step count t0 + i + 1.
namespace {
struct TabChunk {
  int n, off;
  float ns[320], ib[320], bc[320];
};
__global__ void lazy_tab_kernel(TabChunk c, float* neg_step, float* inv_bc2s, float* bc2s) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < c.n) {
    neg_step[c.off + i] = c.ns[i];
    inv_bc2s[c.off + i] = c.ib[i];
    bc2s[c.off + i] = c.bc[i];
  }
}
}  // namespace

int dvt_adam_lazy_tables(const double* h_lr, int step_begin, int step_end, double beta1, double beta2, float* neg_step,
                         float* inv_bc2s, float* bc2s, hipStream_t s) {
  if (!h_lr || !neg_step || !inv_bc2s || !bc2s || step_end <= step_begin) return DVT_E_BADARG;
  for (int o = step_begin; o < step_end; o += 320) {
    TabChunk c{};
    c.off = o - step_begin;
    c.n = step_end - o < 320 ? step_end - o : 320;
    for (int i = 0; i < c.n; ++i) {
      const double t = (double)(o + i + 1);
      c.ns[i] = (float)(-(h_lr[o + i] / (1.0 - pow(beta1, t))));
      c.ib[i] = (float)(1.0 / sqrt(1.0 - pow(beta2, t)));
      c.bc[i] = (float)sqrt(1.0 - pow(beta2, t));
    }
    hipLaunchKernelGGL(lazy_tab_kernel, dim3(5), dim3(64), 0, s, c, neg_step, inv_bc2s, bc2s);
    DVT_CHECK_LAUNCH();
  }
  return 0;
}
