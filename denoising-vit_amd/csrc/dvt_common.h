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

// ---- batched fits: k images advanced by the SAME launches (blockIdx.y = fit) ----
// The per-image fit is a chain of ~10 small dependent launches per Adam step; each launch pays a
// fixed dependent-launch latency (and, next to the extractor, a wait for free CU slots) that does
// not grow when its grid covers several images.  The k fits share one configuration.
int dvt_fit_prep_k(const DvtGridTable* tbl, int k, const float* const* xy, const int32_t* const* ridx,
                   const float* const* params, float* const* enc, const float* const* feat,
                   float* const* raw, int n, int c, hipStream_t stream);
int dvt_grid_bwd_k(const DvtGridTable* tbl, int k, const float* const* xy, const int32_t* const* ridx,
                   const float* const* d_enc, float* const* d_params, uint32_t* const* touched, int n,
                   hipStream_t stream);
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
int dvt_adam_step_k(const DvtAdamArgs* h, int k, float* const* p, float* const* m, float* const* v,
                    float* const* g, uint32_t* const* touched, hipStream_t stream,
                    const DvtAdamRowGather* gather = nullptr);
// offs [steps, lattice + 1] / perm [steps, batch] for steps [0, steps) of idx [steps, batch]; lattice <= 8192,
// batch <= 65535
int dvt_build_row_lists(const int32_t* idx, int steps, int batch, int lattice, int32_t* offs,
                        uint16_t* perm, hipStream_t stream);

// ---- profiling probes (dvt_prof.hip) ----
extern unsigned g_dvt_prof_mask;
void dvt_prof_begin(int probe, hipStream_t s);
void dvt_prof_end(int probe, hipStream_t s, double work);
struct DvtProbeScope {
  int probe;
  hipStream_t s;
  double work;
  bool on;
  DvtProbeScope(int p, hipStream_t st, double w) : probe(p), s(st), work(w), on((g_dvt_prof_mask >> p) & 1u) {
    if (on) dvt_prof_begin(probe, s);
  }
  ~DvtProbeScope() {
    if (on) dvt_prof_end(probe, s, work);
  }
};
int dvt_vit_tune(int gemm_variant);
int dvt_grid_tune(int lds_level_max);
int dvt_adam_tune(int zero_all);

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
// bf16_operands: round operands to bf16 while staging (autocast semantics), fp32 otherwise
int dvt_linear_group(const DvtLinearOp* ops, int n_ops, hipStream_t s, int bf16_operands = 0);
int dvt_fit_prep(const DvtGridTable* tbl, const float* xy, const int32_t* ridx, const float* params,
                 float* enc, const float* feat, float* raw, int n, int c, hipStream_t stream);
