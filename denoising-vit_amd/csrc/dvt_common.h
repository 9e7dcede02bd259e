// Shared host/device helpers for libdvt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dvt_hip.h"

#define DVT_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int dvt_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Wave = 64 lanes on CDNA4; all reductions below are written for exactly that.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Fire-and-forget fp32 global atomic add (global_atomic_add_f32, no return value).
__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- internal cross-TU entry points (not part of the C ABI) ----
int dvt_grid_fwd_idx(const DvtGridTable* tbl, const float* xy, const int32_t* ridx,
                     const float* params, float* enc, int n, hipStream_t stream);
int dvt_grid_bwd_idx(const DvtGridTable* tbl, const float* xy, const int32_t* ridx,
                     const float* d_enc, float* d_params, uint32_t* touched, int n,
                     hipStream_t stream);
int dvt_loss_launch(const float* F, const float* G, const int32_t* g_idx, int lattice,
                    const float* Hres, const float* raw_rows, float* d_pred, float* d_hres,
                    float* d_G, float* row_sums, int n, int c, float grad_scale, hipStream_t s);

// Lazy-exact Adam over the fine hash-grid levels (dvt_adam.hip).  Entries [e0, n_entries) are stepped on demand:
// `done[e - e0]` steps (relative to the first step of the dvt_fit_run call) are applied, the gradient arena holds
// the pending step's gradient.  neg_step / inv_bc2s: device tables indexed by relative step.
struct DvtAdamLazy {
  float* p[DVT_FIT_BATCH_MAX];
  float* m[DVT_FIT_BATCH_MAX];
  float* v[DVT_FIT_BATCH_MAX];
  float* g[DVT_FIT_BATCH_MAX];
  uint16_t* done[DVT_FIT_BATCH_MAX];
  const float* neg_step;
  const float* inv_bc2s;
  const float* bc2s;
  int exact;  // replay with the dense kernel's IEEE division / square root (bit-identical to the dense sweep)
  uint32_t e0, n_entries;
  int nt, l0, n_levels;
  double beta1, beta2, eps, weight_decay;
};
// final_sweep: every lazy entry is brought to `target` applied steps; otherwise only the distinct entries listed
// in ukeys / ucount (this step's lists, [L][nt] / [L]).
int dvt_adam_lazy_k(const DvtAdamLazy* z, int k, bool final_sweep, int target, const uint32_t* const* ukeys,
                    const int32_t* const* ucount, hipStream_t s);

int dvt_adam_lazy_tables(const double* h_lr, int step_begin, int step_end, double beta1, double beta2, float* neg_step,
                         float* inv_bc2s, float* bc2s, hipStream_t s);

// ---- batched fits: k images advanced by the SAME launches (blockIdx.y = fit) ----
// The per-image fit is a chain of ~10 small dependent launches per Adam step; each launch pays a
// fixed dependent-launch latency (and, next to the extractor, a wait for free CU slots) that does
// not grow when its grid covers several images.  The k fits share one configuration.
int dvt_fit_prep_k(const DvtGridTable* tbl, int k, const float* const* xy, const int32_t* const* ridx,
                   const float* const* params, float* const* enc, const float* const* feat,
                   float* const* raw, int n, int c, hipStream_t stream);
int dvt_grid_bwd_k(const DvtGridTable* tbl, int k, const float* const* xy, const int32_t* const* ridx,
                   const float* const* d_enc, float* const* d_params, uint32_t* const* touched, int n,
                   hipStream_t stream, uint32_t bitmap_end = 0u);
int dvt_loss_launch_k(int k, const float* const* F, const float* const* G, const int32_t* const* g_idx,
                      int lattice, const float* const* Hres, const float* const* raw_rows,
                      float* const* d_pred, float* const* d_hres, float* const* d_G,
                      float* const* row_sums, int n, int c, float grad_scale, hipStream_t s);
// Gradient of the shared-artifact map G gathered inside Adam instead of scattered with atomics by the
// loss kernel: G row r of step t receives the d_pred rows of the samples listed in
// perm[offs[r] .. offs[r + 1]) (built once per run from the resident index stream, lists sorted, so
// the sums are deterministic).  rows == nullptr: read the dense gradient buffer as usual.
struct DvtAdamRowGather {
  int64_t begin, end;           // arena range [begin, end) of G (floats, 256-aligned)
  int c, lattice;               // floats per G row, rows
  const int32_t* offs[DVT_FIT_BATCH_MAX];   // [lattice + 1] of this step, per fit
  const uint16_t* perm[DVT_FIT_BATCH_MAX];  // [batch] of this step, per fit
  const float* rows[DVT_FIT_BATCH_MAX];     // d_pred [batch, c], per fit
};
// lazy_next != nullptr: the same launch also runs the lazy-Adam catch-up of the NEXT step's distinct entries
// (dvt_adam_lazy_k(lazy_next, k, false, lazy_target, lazy_ukeys, lazy_ucount) side by side with the dense sweep)
struct DvtAdamLazy;
int dvt_adam_step_k(const DvtAdamArgs* h, int k, float* const* p, float* const* m, float* const* v,
                    float* const* g, uint32_t* const* touched, hipStream_t stream,
                    const DvtAdamRowGather* gather = nullptr, int reverse = 0, const DvtAdamLazy* lazy_next = nullptr,
                    int lazy_target = 0, const uint32_t* const* lazy_ukeys = nullptr,
                    const int32_t* const* lazy_ucount = nullptr, const struct DvtShadowLayout* shadow_L = nullptr,
                    void* const* shadow = nullptr);  // + (with lazy_next) the weight shadow stored by the sweep
// offs [steps, lattice + 1] / perm [steps, batch] for steps [0, steps) of idx [steps, batch]; lattice <= 8192,
// batch <= 65535
int dvt_build_row_lists(const int32_t* idx, int steps, int batch, int lattice, int32_t* offs,
                        uint16_t* perm, hipStream_t stream);

// ---- fused row kernel of the fit (dvt_fit_fused.hip) ----
// SHADOW copies of the five MLP weight matrices (W1, W2, Wh1, Wh2, Wh3), rebuilt from the fp32
// master weights after every Adam step: [N][K] (forward operand) and, where a data gradient flows back
// through the layer, [K][N] (dgrad operand), both in MFMA-fragment-major order (dvt_frag_off / dvt_frag_off32) so that
// every B fragment of the fused kernel is one fully coalesced 16-byte-per-lane global load.  Element type: bf16 in the
// bf16-operand mode (v_mfma_f32_16x16x32_bf16), fp32 in the fp32-operand mode (round 5: v_mfma_f32_16x16x4_f32; f32 = 1).
#define DVT_SHADOW_MATS 5
struct DvtShadowLayout {
  int n;                         // matrices present (0: no shadow maintained)
  int N[DVT_SHADOW_MATS];        // rows of the fp32 matrix (output features)
  int K[DVT_SHADOW_MATS];        // columns (input features), K % 8 == 0
  long long begin[DVT_SHADOW_MATS];   // arena float offset of the matrix
  long long direct[DVT_SHADOW_MATS];  // shadow element offset of the [N][K] copy
  long long transp[DVT_SHADOW_MATS];  // shadow element offset of the [K][N] copy, -1: none
  long long lo, hi;              // arena float range covering all matrices (quick wave-uniform reject)
  long long total;               // elements (2 bytes each, or 4 with f32)
  int f32;                       // element type of the copies: 0 bf16, 1 fp32
};
int dvt_shadow_layout(const DvtFitConfig* c, DvtShadowLayout* out);

// (re)build the shadow copies of the matrices inside arena floats [lo, hi) from the fp32 master weights: the
// whole range at the start of a run, the ranges Adam just stepped after every step (a 2-3 us launch; inside
// the Adam kernel the extra scalar registers cost its streaming loop a wave per SIMD, ~10 % bandwidth)
int dvt_shadow_build_k(const DvtShadowLayout* L, int k, const float* const* params, void* const* shadow,
                       long long lo, long long hi, hipStream_t s);
// Operands of the weight-gradient GEMMs, written by the row kernel as bf16 (fp32 with f32 operands) [cols][batch]
// matrices in fragment-major order (dvt_frag_off / dvt_frag_off32 with K = batch): the batch index is the contraction index.
enum { DVT_T_DF = 0, DVT_T_H1, DVT_T_DH1, DVT_T_ENC, DVT_T_RAW, DVT_T_R1, DVT_T_R2, DVT_T_DH, DVT_T_DR2, DVT_T_DR1, DVT_T_COUNT };
struct DvtTLayout {
  long long off[DVT_T_COUNT];  // element offsets
  int cols[DVT_T_COUNT];
  long long total;
};
void dvt_t_layout(const DvtFitConfig* c, DvtTLayout* out);
struct DvtFusedFit {  // per fit: inputs, arena, shadow weights and what the row kernel hands to the rest of the step
  const float* xy;
  const int32_t* ridx;
  const float* feat;
  const float* params;
  const void* shadow;  // bf16 or fp32 elements (DvtShadowLayout.f32)
  void* T;            // transposed operand copies (DvtTLayout; element type as the shadow's) -> weight-gradient kernel
  float *F, *Hres;    // fp32 rows the loss stage reads back
  float *dF;          // fp32 d(pred) rows -> Adam gathers the gradient of G from them
  float *denc;        // fp32 d(enc) rows -> grid backward
  float *rows;        // per-row loss sums -> loss logging
  float* grads;       // gradient arena (weight gradients accumulate here with fp32 atomics)
  const int32_t* g_offs;   // this step's row lists of the G gradient (DvtAdamRowGather), phase 1 only
  const uint16_t* g_perm;
  uint32_t* touched;       // bitmap of grid entries with a gradient (grid backward -> Adam)
  const uint32_t* gs_keys; // this step's sorted grid-corner lists (dvt_grid_dev.h: GridSortedPtrs), or nullptr
  const uint16_t* gs_pay;
  const float* gs_w;
  uint32_t gs_bitmap_end;  // grid entries >= this get no `touched` bit (lazy Adam owns them); 0xffffffff: all do
};
#if defined(__HIPCC__)
typedef __bf16 dvt_hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float dvt_f32x2 __attribute__((ext_vector_type(2)));
// fp32 pair -> packed bf16 pair, round-to-nearest-even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t dvt_pack_bf16x2(float a, float b) {
  const dvt_hwbf16x2 v = __builtin_convertvector((dvt_f32x2){a, b}, dvt_hwbf16x2);
  return __builtin_bit_cast(uint32_t, v);
}
// Element offset of W[n][k] (k-contiguous matrix with K columns, N % 16 == 0, K % 32 == 0) inside its
// FRAGMENT-MAJOR shadow copy: the 16 x 32 block (tile n / 16, k-step k / 32) is one contiguous 1-KB piece
// holding, lane by lane, exactly the B fragment of v_mfma_f32_16x16x32_bf16 (lane = 16 * ((k % 32) / 8) +
// n % 16, 8 consecutive k per lane).  One wave-wide 16-byte load then touches 8 full 128-B lines; with a
// plain row-major copy the four lanes that share a 64-B segment are 16 lanes apart, the address coalescer
// only merges neighbours, and every load cost 64 L1 tag look-ups (measured: the row kernel was bound by
// exactly that, 14.4 M TCP accesses per launch = 55 us).
__host__ __device__ __forceinline__ long long dvt_frag_off(int n, int k, int K) {
  return ((long long)(n >> 4) * (K >> 5) + (k >> 5)) * 512 + (((((k >> 3) & 3) << 4) + (n & 15)) << 3) + (k & 7);
}
// The fp32 twin for v_mfma_f32_16x16x4_f32, read four MFMA steps at a time: the 16 x 16 block (tile n / 16, super-step
// k / 16) is one contiguous 1-KB piece, lane = 16 * ((k % 16) / 4) + n % 16 holds the 4 consecutive k of its k-group --
// component j of a lane's float4 is its B (or A) value of sub-step j, whose MFMA contracts k = 16 S + 4 g + j over the four
// lane groups g.  (Any partition of k into MFMA steps is valid as long as both operands use the same one.)
__host__ __device__ __forceinline__ long long dvt_frag_off32(int n, int k, int K) {
  return ((long long)(n >> 4) * (K >> 4) + (k >> 4)) * 256 + (((((k >> 2) & 3) << 4) + (n & 15)) << 2) + (k & 3);
}
// The four consecutive arena floats v at float offset e (e % 4 == 0) -> their shadow copies, when e
// lies inside one of the shadowed matrices (row-major [N][K], K % 4 == 0: the four share a row).
__device__ __forceinline__ void dvt_shadow_store(const DvtShadowLayout& L, void* __restrict__ shv, long long e,
                                                 float4 v) {
  uint16_t* __restrict__ sh = static_cast<uint16_t*>(shv);
#pragma unroll
  for (int i = 0; i < DVT_SHADOW_MATS; ++i) {
    const long long rel64 = e - L.begin[i];
    const int N = L.N[i], K = L.K[i];
    if (rel64 >= 0 && rel64 < (long long)N * K) {
      const int rel = (int)rel64, n = rel / K, k = rel - n * K;
      if (L.f32) {  // (wave-uniform) fp32 copies: the four k share a lane's float4 in the direct copy
        float* __restrict__ sf = static_cast<float*>(shv);
        *reinterpret_cast<float4*>(sf + L.direct[i] + dvt_frag_off32(n, k, K)) = v;
        if (L.transp[i] >= 0) {
          float* t = sf + L.transp[i];
          t[dvt_frag_off32(k + 0, n, N)] = v.x;
          t[dvt_frag_off32(k + 1, n, N)] = v.y;
          t[dvt_frag_off32(k + 2, n, N)] = v.z;
          t[dvt_frag_off32(k + 3, n, N)] = v.w;
        }
        continue;
      }
      const uint32_t lo = dvt_pack_bf16x2(v.x, v.y), hi = dvt_pack_bf16x2(v.z, v.w);
      // forward operand W[n][k..k+3]: four consecutive k stay inside one lane's 8-element run
      *reinterpret_cast<uint2*>(sh + L.direct[i] + dvt_frag_off(n, k, K)) = make_uint2(lo, hi);
      if (L.transp[i] >= 0) {  // dgrad operand Wt[k][n] = W[n][k]: an [K][N] matrix, n is its contiguous index
        uint16_t* t = sh + L.transp[i];
        t[dvt_frag_off(k + 0, n, N)] = (uint16_t)(lo & 0xffffu);
        t[dvt_frag_off(k + 1, n, N)] = (uint16_t)(lo >> 16);
        t[dvt_frag_off(k + 2, n, N)] = (uint16_t)(hi & 0xffffu);
        t[dvt_frag_off(k + 3, n, N)] = (uint16_t)(hi >> 16);
      }
    }
  }
}
#endif
bool dvt_fit_fused_ok(const DvtFitConfig* c);         // shapes fit AND the bf16-operand mode is selected
bool dvt_fit_fused_shapes_ok(const DvtFitConfig* c);  // shapes only (workspace carving must not depend on the mode)
int dvt_fit_rows_k(const DvtFitConfig* c, const DvtShadowLayout* L, int k, const DvtFusedFit* fits, bool phase2,
                   hipStream_t s);
// everything of the backward pass that reduces over rows, one launch: hash-grid backward || all weight (and
// bias) gradients from the transposed operand copies (+ the gradient of G)
int dvt_fit_backward_k(const DvtFitConfig* c, int k, const DvtFusedFit* fits, bool phase2, hipStream_t s);

// ---- profiling probes (dvt_prof.hip) ----
extern unsigned g_dvt_prof_mask;
long dvt_prof_begin(int probe, hipStream_t s, unsigned long long* gen);  // -> this scope's sample index + pool generation (thread-safe, dvt_prof.hip)
void dvt_prof_end(int probe, long idx, unsigned long long gen, hipStream_t s, double work);
struct DvtProbeScope {
  int probe;
  hipStream_t s;
  double work;
  long idx;
  unsigned long long gen;
  DvtProbeScope(int p, hipStream_t st, double w) : probe(p), s(st), work(w), idx(-1), gen(0) {
    if ((g_dvt_prof_mask >> p) & 1u) idx = dvt_prof_begin(probe, s, &gen);
  }
  ~DvtProbeScope() {
    if (idx >= 0) dvt_prof_end(probe, idx, gen, s, work);
  }
};
int dvt_vit_tune(int gemm_variant);
int dvt_grid_tune(int lds_level_max);
int dvt_adam_tune(int zero_all);
int dvt_s2_tune(int mask);

// One linear-layer contraction for the grouped launch (dvt_gemm_f32.hip).
struct DvtLinearOp {
  int kind;  // 0: y = act(x.w^T + b); 1: dw += dy^T.x (+ db); 2: dx = dy.w (* relu_mask > 0)
  const float* x;
  const float* w;
  const float* b;
  const float* dy;
  const float* relu_mask;
  float* y;
  float* dw;
  float* db;
  float* dx;
  int m, n, k, relu;
};
// General exact-fp32 GEMM (dvt_gemm_f32.hip), the building block of the stage-2 trainer.
//   layout 0: C[M][N] = A[M][K] . B[N][K]^T   (both k-contiguous: a forward linear layer)
//   layout 1: C[M][N] = A[M][K] . B[K][N]     (a data gradient; N % 64 == 0)
//   layout 2: C[M][N] = A[K][M]^T . B[K][N]   (a weight gradient; M % 64 == 0, N % 64 == 0)
// K % 64 == 0.  `accumulate`: fp32 atomic adds into C (the reduction is split over workgroups);
// `colsum` (layout 2): colsum[m] += sum_k A[k][m] (bias gradient).  nb0 x nb1 > 1: batched over
// (b0, b1) with element strides s?0 / s?1, no split.
struct DvtGemmEx {
  int layout;
  const float* A;
  const float* B;
  float* C;
  int M, N, K, lda, ldb, ldc;
  const float* bias;
  float* colsum;
  int accumulate;
  int nb0, nb1;
  long long sA0, sA1, sB0, sB1, sC0, sC1;
  // softmax-backward epilogue (batched layout 0 with K <= 64 only; round 6): C = oscale * smul (.) (A B^T - rowsub), smul
  // indexed like C (batch strides sC0 / sC1, leading dimension ldc), rowsub[(b0 * nb1 + b1) * M + m]
  const float* smul;
  const float* rowsub;
  float oscale;
};
int dvt_gemm_f32_ex(const DvtGemmEx* g, hipStream_t s);
// bf16_operands: round operands to bf16 while staging (autocast semantics), fp32 otherwise
int dvt_linear_group(const DvtLinearOp* ops, int n_ops, hipStream_t s, int bf16_operands = 0);
int dvt_fit_prep(const DvtGridTable* tbl, const float* xy, const int32_t* ridx, const float* params,
                 float* enc, const float* feat, float* raw, int n, int c, hipStream_t stream);
