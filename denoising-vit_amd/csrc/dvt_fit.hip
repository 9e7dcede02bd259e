// The fused per-image fit loop: every launch of one Adam step is enqueued from native code,
// with no host<->device traffic inside the loop (the index stream is resident, there is no
// range-assert sync, losses are reduced on the device only on logging steps).
//
// Reference: main_img_denoising.py:67-89 (denoise_an_image inner loop), with the models of
// neural_feature_field.py:40-49 and offline_denoiser.py:62-140.  Quirks reproduced
// (SURVEY.md Q1-Q5): gradients reach Adam scaled by 1024; dense Adam incl. weight decay on
// untouched grid entries; per-tensor step counts (G stops, h starts at the switch);
// phase 2 iff step > int(freeze_after * num_iters).
//
// Arena layout (float offsets, each tensor padded to a multiple of 256 floats so that one
// Adam wave-iteration never straddles tensors and the touched bitmap words line up):
//   [ grid | W1 | b1 | W2 | b2 | G | Wh1 | bh1 | Wh2 | bh2 | Wh3 | bh3 ]
// phase 1 steps the contiguous prefix [grid .. G], phase 2 steps [grid .. b2] and [Wh1 .. bh3].
#include <math.h>

#include "dvt_common.h"
#include "dvt_grid_dev.h"

extern int g_fit_sorted_grid;
#ifdef DVT_LAB
// developer library only, dvt_tune_set(15, mask): TIMING-ONLY ablation of the fused step -- bit 0 skips the row kernel, bit 1 the
// backward kernel, bit 2 the Adam launch (results are wrong by construction; what each launch costs the PIPELINE: tools/r06)
int g_fit_skip_mask = 0;
#endif
int g_fit_lazy_adam = 1;  // dvt_tune_set(9, 0): dense Adam over the whole arena
int g_fit_shadow_in_adam = 1;  // dvt_tune_set(12, 0): shadow_build_kernel after every Adam launch
int g_fit_lazy_merge = 1;  // dvt_tune_set(11, 0): catch-up as its own launch
// Lazy-Adam replay arithmetic.  0 (default): v_rcp_f32 / v_sqrt_f32, 1 ulp each -- a tolerance-tested APPROXIMATION of the
// dense sweep; dvt_tune_set(10, 1): IEEE division / sqrt, bit-identical to the dense sweep, +18 us per step.  Same-box A/B
// of the bench (profiles/r03/r03a_bench_ab_adam_replay_and_gemm8q.txt): 2.70 vs 2.56 images/s (5.4 %), identical oracle
// parity of both modes over the 1000-step schedule (tests/test_gpu_parity_full.py fixture test, both replays).
int g_fit_lazy_exact = 0;
int g_fit_lazy_refresh = 32;  // dvt_tune_set(9, n >= 2): steps between full sweeps of the lazy region

int g_adam_pingpong = 1;  // dvt_tune_set(8, 0): always sweep forward (A/B timing)

namespace {

inline int64_t up256(int64_t x) { return (x + 255) / 256 * 256; }

struct Work {
  float *enc, *h1, *F, *raw, *dF, *dh1, *denc, *rows, *r1, *r2, *Hres, *dH, *dr2, *dr1;
  int32_t* g_offs;   // [num_iters, lattice + 1] row lists of the G gradient (nullptr: atomics path)
  uint16_t* g_perm;  // [num_iters, batch]
  void* shadow;      // shadow copies (bf16 or fp32) of the MLP weights for the fused row kernel (nullptr: shapes not eligible)
  void* T;           // transposed operand copies for the weight-gradient kernel (with `shadow`)
  // sorted grid-corner lists of the current chunk of GS_CHUNK steps (dvt_grid_dev.h; nullptr: atomics path)
  uint32_t* gs_keys;
  uint16_t* gs_pay;
  float* gs_w;
  // lazy-exact Adam over the fine grid levels (dvt_adam.hip): distinct-entry lists of the chunk, per-entry step
  // counters, per-step scalar tables (nullptr: dense Adam everywhere)
  uint32_t* gs_ukeys;
  int32_t* gs_ucount;
  uint16_t* lazy_done;
  float *lazy_ns, *lazy_ib, *lazy_bc;
};
constexpr int GS_CHUNK = 128;  // steps per sort launch: 168 MB of lists per fit at batch 2048, 16 levels

bool row_lists_ok(const DvtFitConfig* c) { return c->lattice <= 8192 && c->batch <= 65535; }

// Grid entries [e0, n_entries_total) are stepped lazily: everything from the first level with >= 64 k entries on
// (a step touches <= 4 * batch = 8 k of them), e0 rounded up to a whole `touched` word / Adam chunk.
// sorted lists / lazy Adam are available where their buffers were carved (dvt_fit_fused_shapes_ok), in both operand
// precisions; dvt_tune_set(6, 0) (fused step off) leaves them to the bf16 layer-by-layer A/B as before
bool g_fit_fused_enable_lists(const DvtFitConfig* c) {
  return dvt_fit_fused_shapes_ok(c) && (!c->mlp_bf16 || dvt_fit_fused_ok(c));  // (fp32 operands: fused or not)
}

bool lazy_range(const DvtFitConfig* c, uint32_t* e0, int* l0) {
  if (c->num_iters > 65535) return false;  // 16-bit step counters
  for (int l = 0; l < c->grid.n_levels; ++l)
    if (c->grid.entries[l] >= 65536u) {
      const uint32_t e = (c->grid.offset[l] + 31u) & ~31u;
      if (e >= c->grid.n_entries_total) return false;
      *e0 = e;
      *l0 = l;
      return true;
    }
  return false;
}

int64_t carve(const DvtFitConfig* c, float* base, Work* w) {
  const int64_t B = c->batch, C = c->feat_dim, H = c->hidden, R = c->res_hidden;
  const int64_t E = (int64_t)c->grid.n_levels * c->grid.n_features;
  int64_t o = 0;
  auto take = [&](int64_t n) {
    float* p = base ? base + o : nullptr;
    o += up256(n);
    return p;
  };
  Work t;
  t.enc = take(B * E);
  t.h1 = take(B * H);
  t.F = take(B * C);
  t.raw = take(B * C);
  t.dF = take(B * C);
  t.dh1 = take(B * H);
  t.denc = take(B * E);
  t.rows = take(B * 8);
  t.r1 = take(B * R);
  t.r2 = take(B * R);
  t.Hres = take(B * C);
  t.dH = take(B * C);
  t.dr2 = take(B * R);
  t.dr1 = take(B * R);
  t.g_offs = nullptr;
  t.g_perm = nullptr;
  if (row_lists_ok(c)) {
    t.g_offs = reinterpret_cast<int32_t*>(take((int64_t)c->num_iters * (c->lattice + 1)));
    t.g_perm = reinterpret_cast<uint16_t*>(take(((int64_t)c->num_iters * B + 1) / 2));
  }
  t.shadow = nullptr;
  t.T = nullptr;
  bool fused_bufs = false, lazy_bufs = false;  // (pointers are all null in the size query: never test them here)
  if (dvt_fit_fused_shapes_ok(c)) {
    DvtShadowLayout L;
    if (dvt_shadow_layout(c, &L) == 0) {
      fused_bufs = true;
      // sized for fp32 elements whatever the mode of this run: the operand precision may change between runs on one
      // workspace (bench.py switches it), and the carving must not depend on it
      t.shadow = take(L.total);
      DvtTLayout TL;
      dvt_t_layout(c, &TL);
      t.T = take(TL.total);
    }
  }
  t.gs_keys = nullptr;
  t.gs_pay = nullptr;
  t.gs_w = nullptr;
  if (fused_bufs && dvt_grid_sorted_ok(&c->grid, c->batch)) {
    const int64_t per_step = (int64_t)c->grid.n_levels * 4 * B;
    t.gs_keys = reinterpret_cast<uint32_t*>(take(per_step * GS_CHUNK));
    t.gs_w = take(per_step * GS_CHUNK);
    t.gs_pay = reinterpret_cast<uint16_t*>(take((per_step * GS_CHUNK + 1) / 2));
    uint32_t e0;
    int l0;
    if (lazy_range(c, &e0, &l0)) {
      t.gs_ukeys = reinterpret_cast<uint32_t*>(take(per_step * GS_CHUNK));
      t.gs_ucount = reinterpret_cast<int32_t*>(take((int64_t)c->grid.n_levels * GS_CHUNK));
      t.lazy_done = reinterpret_cast<uint16_t*>(take(((int64_t)(c->grid.n_entries_total - e0) + 1) / 2));
      t.lazy_ns = take(c->num_iters);
      t.lazy_ib = take(c->num_iters);
      t.lazy_bc = take(c->num_iters);
      lazy_bufs = true;
    }
  }
  if (!lazy_bufs) {
    t.gs_ukeys = nullptr;
    t.gs_ucount = nullptr;
    t.lazy_done = nullptr;
    t.lazy_ns = t.lazy_ib = t.lazy_bc = nullptr;
  }
  if (w) *w = t;
  return o;
}

int check_cfg(const DvtFitConfig* c) {
  if (!c) return DVT_E_BADARG;
  if (c->feat_dim <= 0 || (c->feat_dim & 3) || c->feat_dim > 1024) return DVT_E_BADARG;
  if (c->hidden <= 0 || (c->hidden & 3) || c->res_hidden <= 0 || (c->res_hidden & 3))
    return DVT_E_BADARG;
  if (c->lattice <= 0 || c->n_rows <= 0 || c->batch <= 0 || c->num_iters <= 0) return DVT_E_BADARG;
  if (c->grid.n_features != 8 || c->grid.n_levels < 1 || c->grid.n_levels > DVT_MAX_LEVELS)
    return DVT_E_BADARG;
  if (c->off_grid != 0) return DVT_E_BADARG;  // bitmap word <-> arena chunk correspondence
  return 0;
}

}  // namespace

extern "C" int dvt_fit_layout(DvtFitConfig* c) {
  if (!c) return DVT_E_BADARG;
  c->off_grid = 0;
  int rc = check_cfg(c);
  if (rc) return rc;
  const int64_t C = c->feat_dim, H = c->hidden, R = c->res_hidden;
  const int64_t E = (int64_t)c->grid.n_levels * c->grid.n_features;
  int64_t o = 0;
  c->off_grid = o; o += up256((int64_t)c->grid.n_entries_total * c->grid.n_features);
  c->off_w1 = o;   o += up256(H * E);
  c->off_b1 = o;   o += up256(H);
  c->off_w2 = o;   o += up256(C * H);
  c->off_b2 = o;   o += up256(C);
  c->off_G = o;    o += up256((int64_t)c->lattice * C);
  c->off_wh1 = o;  o += up256(R * C);
  c->off_bh1 = o;  o += up256(R);
  c->off_wh2 = o;  o += up256(R * R);
  c->off_bh2 = o;  o += up256(R);
  c->off_wh3 = o;  o += up256(C * R);
  c->off_bh3 = o;  o += up256(C);
  c->arena_floats = o;
  return 0;
}

extern "C" int64_t dvt_fit_workspace_floats(const DvtFitConfig* c) {
  if (check_cfg(c)) return -1;
  return carve(c, nullptr, nullptr);
}

extern "C" int dvt_field_infer(const DvtFitConfig* c, const float* params, const float* xy,
                               float* out, float* workspace, int n, void* stream) {
  int rc = check_cfg(c);
  if (rc) return rc;
  if (!params || !xy || !out || !workspace || n < 0) return DVT_E_BADARG;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int E = c->grid.n_levels * c->grid.n_features;
  float* enc = workspace;
  float* h1 = workspace + up256((int64_t)n * E);
  rc = dvt_grid_fwd_idx(&c->grid, xy, nullptr, params + c->off_grid, enc, n, s);
  if (rc) return rc;
  rc = dvt_linear_fwd(enc, params + c->off_w1, params + c->off_b1, h1, n, c->hidden, E, 1, s);
  if (rc) return rc;
  return dvt_linear_fwd(h1, params + c->off_w2, params + c->off_b2, out, n, c->feat_dim,
                        c->hidden, 0, s);
}

namespace {

#define DVT_TRY(x)         \
  do {                     \
    int rc__ = (x);        \
    if (rc__) return rc__; \
  } while (0)

int check_bufs(const DvtFitConfig* c, const DvtFitBuffers* b, int step_begin, int step_end) {
  int rc = check_cfg(c);
  if (rc) return rc;
  if (!b || !b->feat || !b->xy || !b->idx || !b->params || !b->adam_m || !b->adam_v ||
      !b->grads || !b->touched || !b->workspace || !b->h_lr)
    return DVT_E_BADARG;
  if (step_begin < 0 || step_end > c->num_iters || step_begin > step_end) return DVT_E_BADARG;
  return 0;
}

// One Adam step of k images' fits (same configuration, same step), enqueued on stream s: every
// launch covers all k fits (blockIdx.y = fit; the grouped GEMM launch simply carries k x the
// problems).  k = 1 is the reference's per-image loop.
int fit_step(const DvtFitConfig* c, int k, const DvtFitBuffers* const* bs, const Work* ws, int step,
             int gs_local, uint32_t lazy_e0, const DvtAdamLazy* lazy_next, int lazy_target, bool fused, hipStream_t s) {
  // fused: the row-kernel step -- decided ONCE per dvt_fit_run_batched call (which builds the weight shadow the row kernel
  // reads), never re-read from the process-global knob between steps
  // gs_local: index of `step` inside the current chunk of sorted grid lists, or -1;  lazy_e0: first grid entry the lazy
  // Adam kernels own (0xffffffff: none, dense Adam steps everything);  lazy_next: run the catch-up of step + 1 (same
  // chunk of lists, relative step count lazy_target) inside this step's Adam launch
  const int B = c->batch, C = c->feat_dim, H = c->hidden, R = c->res_hidden;
  const int E = c->grid.n_levels * c->grid.n_features;
  const bool phase2 = step > c->switch_step;
  const bool use_res = phase2 && c->enable_residual;
  constexpr int KM = DVT_FIT_BATCH_MAX;
  const float* xy[KM];
  const int32_t* ridx[KM];
  const float* feat[KM];
  float *P[KM], *Gd[KM], *M[KM], *V[KM];
  uint32_t* touched[KM];
  for (int f = 0; f < k; ++f) {
    xy[f] = bs[f]->xy;
    ridx[f] = bs[f]->idx + (size_t)step * B;
    feat[f] = bs[f]->feat;
    P[f] = bs[f]->params;
    Gd[f] = bs[f]->grads;
    M[f] = bs[f]->adam_m;
    V[f] = bs[f]->adam_v;
    touched[f] = bs[f]->touched;
  }
  // per-fit pointer tables: arena tensor at offset `off`, or a workspace member
  auto at = [&](float* const* base, int64_t off, float** out) {
    for (int f = 0; f < k; ++f) out[f] = base[f] + off;
  };
#define WS_TAB(name, member) \
  float* name[KM];           \
  for (int f = 0; f < k; ++f) name[f] = ws[f].member;
  WS_TAB(enc, enc) WS_TAB(raw, raw) WS_TAB(Fp, F) WS_TAB(Hres, Hres) WS_TAB(dF, dF) WS_TAB(dH, dH)
  WS_TAB(rows, rows) WS_TAB(denc, denc)
#undef WS_TAB

  // Linear layers go out as GROUPED launches: independent GEMMs of the step (field branch and
  // residual branch, weight- and data-gradient of one layer, all k fits) share one grid.
  auto fwd_op = [&](int f, const float* x, int64_t ow, int64_t ob, float* y, int n, int kk, int relu) {
    DvtLinearOp o{};
    o.kind = 0; o.x = x; o.w = P[f] + ow; o.b = P[f] + ob; o.y = y; o.m = B; o.n = n; o.k = kk; o.relu = relu;
    return o;
  };
  auto wgrad_op = [&](int f, const float* dy, const float* x, int64_t ow, int64_t ob, int n, int kk) {
    DvtLinearOp o{};
    o.kind = 1; o.dy = dy; o.x = x; o.dw = Gd[f] + ow; o.db = Gd[f] + ob; o.m = B; o.n = n; o.k = kk;
    return o;
  };
  auto dgrad_op = [&](int f, const float* dy, int64_t ow, float* dx, const float* mask, int n, int kk) {
    DvtLinearOp o{};
    o.kind = 2; o.dy = dy; o.w = P[f] + ow; o.dx = dx; o.relu_mask = mask; o.m = B; o.n = n; o.k = kk;
    return o;
  };
  DvtLinearOp ops[4 * KM];  // <= 16 problems per grouped launch (MULTI_MAX)
  int n_ops;
  auto launch = [&]() { return dvt_linear_group(ops, n_ops, s, c->mlp_bf16); };
  DvtShadowLayout shl{};
  void* shadow[KM] = {nullptr, nullptr, nullptr, nullptr};
  DvtFusedFit ff[KM];
  if (fused) {
    // ---- ONE launch for everything row-local: gather, grid forward, MLP forward, loss, dgrad chain ----
    DVT_TRY(dvt_shadow_layout(c, &shl));
    for (int f = 0; f < k; ++f) {
      const Work& w = ws[f];
      shadow[f] = w.shadow;
      const size_t gso = (size_t)gs_local * c->grid.n_levels * 4 * B;
      const bool gs = w.gs_keys != nullptr && gs_local >= 0;
      ff[f] = DvtFusedFit{xy[f], ridx[f], feat[f], P[f], w.shadow, w.T, w.F, w.Hres, w.dF, w.denc, w.rows, Gd[f],
                          w.g_offs + (size_t)step * (c->lattice + 1), w.g_perm + (size_t)step * B, touched[f],
                          gs ? w.gs_keys + gso : nullptr, gs ? w.gs_pay + gso : nullptr, gs ? w.gs_w + gso : nullptr,
                          lazy_e0};
    }
#ifdef DVT_LAB
    if (!(g_fit_skip_mask & 1))
#endif
    DVT_TRY(dvt_fit_rows_k(c, &shl, k, ff, use_res, s));
  } else {
  // ---- forward ----
  {
    float* gridp[KM];
    at(P, c->off_grid, gridp);
    DVT_TRY(dvt_fit_prep_k(&c->grid, k, xy, ridx, gridp, enc, feat, raw, B, C, s));
  }
  n_ops = 0;
  for (int f = 0; f < k; ++f) {
    const Work& w = ws[f];
    ops[n_ops++] = fwd_op(f, w.enc, c->off_w1, c->off_b1, w.h1, H, E, 1);
    if (use_res) ops[n_ops++] = fwd_op(f, w.raw, c->off_wh1, c->off_bh1, w.r1, R, C, 1);
  }
  DVT_TRY(launch());
  n_ops = 0;
  for (int f = 0; f < k; ++f) {
    const Work& w = ws[f];
    ops[n_ops++] = fwd_op(f, w.h1, c->off_w2, c->off_b2, w.F, C, H, 0);
    if (use_res) ops[n_ops++] = fwd_op(f, w.r1, c->off_wh2, c->off_bh2, w.r2, R, R, 1);
  }
  DVT_TRY(launch());
  if (use_res) {
    n_ops = 0;
    for (int f = 0; f < k; ++f) ops[n_ops++] = fwd_op(f, ws[f].r2, c->off_wh3, c->off_bh3, ws[f].Hres, C, R, 0);
    DVT_TRY(launch());
  }
  // ---- loss + d(pred), G gradient scattered in the same pass while G still trains ----
  {
    float *Gp[KM], *dG[KM];
    at(P, c->off_G, Gp);
    at(Gd, c->off_G, dG);
    if (phase2 || ws[0].g_offs != nullptr)  // G frozen, or its gradient is gathered inside Adam
      for (int f = 0; f < k; ++f) dG[f] = nullptr;
    DVT_TRY(dvt_loss_launch_k(k, Fp, Gp, ridx, c->lattice, use_res ? Hres : nullptr, raw, dF, dH, dG,
                              rows, B, C, (float)c->grad_scale, s));
  }
  }  // !fused
  for (int f = 0; f < k; ++f) {
    const DvtFitBuffers* b = bs[f];
    const bool log = b->losses != nullptr &&
                     ((b->log_every > 0 && step % b->log_every == 0) || step == c->num_iters - 1);
    if (log) DVT_TRY(dvt_loss_reduce(ws[f].rows, b->losses + (size_t)step * 8, B, C, use_res, s));
  }
  if (fused) {
    // ---- everything that reduces over rows in ONE launch: grid backward || weight gradients (+ dG)
#ifdef DVT_LAB
    if (!(g_fit_skip_mask & 2))
#endif
    DVT_TRY(dvt_fit_backward_k(c, k, ff, use_res, s));
  } else {
  // ---- backward: {field layer 2, h layer 3}, {field layer 1, h layer 2}, hash grid, {h layer 1}
  n_ops = 0;
  for (int f = 0; f < k; ++f) {
    const Work& w = ws[f];
    ops[n_ops++] = dgrad_op(f, w.dF, c->off_w2, w.dh1, w.h1, C, H);
    ops[n_ops++] = wgrad_op(f, w.dF, w.h1, c->off_w2, c->off_b2, C, H);
    if (use_res) {
      ops[n_ops++] = dgrad_op(f, w.dH, c->off_wh3, w.dr2, w.r2, C, R);
      ops[n_ops++] = wgrad_op(f, w.dH, w.r2, c->off_wh3, c->off_bh3, C, R);
    }
  }
  DVT_TRY(launch());
  n_ops = 0;
  for (int f = 0; f < k; ++f) {
    const Work& w = ws[f];
    ops[n_ops++] = dgrad_op(f, w.dh1, c->off_w1, w.denc, nullptr, H, E);
    ops[n_ops++] = wgrad_op(f, w.dh1, w.enc, c->off_w1, c->off_b1, H, E);
    if (use_res) {
      ops[n_ops++] = dgrad_op(f, w.dr2, c->off_wh2, w.dr1, w.r1, R, R);
      ops[n_ops++] = wgrad_op(f, w.dr2, w.r1, c->off_wh2, c->off_bh2, R, R);
    }
  }
  DVT_TRY(launch());
  {
    float* dgrid[KM];
    at(Gd, c->off_grid, dgrid);
    bool lists = gs_local >= 0;
    for (int f = 0; f < k; ++f) lists = lists && ws[f].gs_keys != nullptr;
    if (lists) {
      // this step's sorted (entry, sample, corner) lists exist (round 4: also with fp32 operands -- they depend on the
      // coordinates and the index stream only): gather, deterministic sums, plain stores; entries from lazy_e0 on belong
      // to the lazy Adam kernels and get no `touched` bit
      const uint32_t* gk[KM];
      const uint16_t* gp[KM];
      const float* gw[KM];
      const size_t gso = (size_t)gs_local * c->grid.n_levels * 4 * B;
      for (int f = 0; f < k; ++f) {
        gk[f] = ws[f].gs_keys + gso;
        gp[f] = ws[f].gs_pay + gso;
        gw[f] = ws[f].gs_w + gso;
      }
      DVT_TRY(dvt_grid_gather_k(&c->grid, k, gk, gp, gw, B, lazy_e0, denc, dgrid, touched, s));
    } else {
      DVT_TRY(dvt_grid_bwd_k(&c->grid, k, xy, ridx, denc, dgrid, touched, B, s,
                             lazy_e0 != 0xffffffffu ? lazy_e0 : 0u));
    }
  }
  if (use_res) {
    n_ops = 0;
    for (int f = 0; f < k; ++f)
      ops[n_ops++] = wgrad_op(f, ws[f].dr1, ws[f].raw, c->off_wh1, c->off_bh1, R, C);
    DVT_TRY(launch());
  }
  }  // !fused
  // ---- Adam (dense) + zero_grad ----
  DvtAdamArgs a{};
  a.beta1 = c->beta1;
  a.beta2 = c->beta2;
  a.eps = c->eps;
  a.weight_decay = c->weight_decay;
  a.sparse_end = c->off_w1;
  const double lr = bs[0]->h_lr[step];
  auto seg = [&](int64_t beg, int64_t end, int t) {
    DvtAdamSeg sg{};
    sg.begin = beg;
    sg.end = end;
    sg.lr = lr;
    sg.bias_correction1 = 1.0 - pow(c->beta1, (double)t);
    sg.bias_correction2_sqrt = sqrt(1.0 - pow(c->beta2, (double)t));
    sg.active = 1;
    a.segs[a.n_segs++] = sg;
  };
  const bool lazy = lazy_e0 != 0xffffffffu;
  const int64_t dense_grid_end = lazy ? (int64_t)lazy_e0 * 8 : c->off_w1;  // the lazy kernels own [lazy_e0 * 8, off_w1)
  if (lazy) {
    a.sparse_end = dense_grid_end;
    seg(c->off_grid, dense_grid_end, step + 1);  // coarse grid levels
  }
  if (!phase2) {
    seg(lazy ? c->off_w1 : c->off_grid, c->off_wh1, step + 1);  // (grid +) field MLP + G
  } else {
    seg(lazy ? c->off_w1 : c->off_grid, c->off_G, step + 1);  // (grid +) field MLP (G frozen: grad None)
    if (use_res) seg(c->off_wh1, c->arena_floats, step - c->switch_step);  // h: own step count
  }
  DvtAdamRowGather gather{};
  if (!phase2 && ws[0].g_offs != nullptr && !fused) {  // (fused step: the weight-gradient launch wrote dG densely)
    gather.begin = c->off_G;
    gather.end = c->off_wh1;
    gather.c = C;
    gather.lattice = c->lattice;
    for (int f = 0; f < k; ++f) {
      gather.offs[f] = ws[f].g_offs + (size_t)step * (c->lattice + 1);
      gather.perm[f] = ws[f].g_perm + (size_t)step * B;
      gather.rows[f] = ws[f].dF;
    }
  }
  {
    const uint32_t* uk[KM];
    const int32_t* uc[KM];
    if (lazy_next != nullptr)
      for (int f = 0; f < k; ++f) {
        uk[f] = ws[f].gs_ukeys + (size_t)(gs_local + 1) * c->grid.n_levels * 4 * B;
        uc[f] = ws[f].gs_ucount + (size_t)(gs_local + 1) * c->grid.n_levels;
      }
    const bool shadow_in_adam = fused && lazy_next != nullptr && g_fit_shadow_in_adam;
#ifdef DVT_LAB
    if (g_fit_skip_mask & 4) return 0;
#endif
    DVT_TRY(dvt_adam_step_k(&a, k, P, M, V, Gd, touched, s, &gather, g_adam_pingpong ? (step & 1) : 0, lazy_next,
                            lazy_target, uk, uc, shadow_in_adam ? &shl : nullptr, shadow_in_adam ? shadow : nullptr));
    if (shadow_in_adam) return 0;
  }
  if (fused) {  // bf16 shadow copies of the weights Adam just stepped, for the next step's row kernel
    const float* pp[KM];
    for (int f = 0; f < k; ++f) pp[f] = P[f];
    DVT_TRY(dvt_shadow_build_k(&shl, k, pp, shadow, c->off_w1, use_res ? c->arena_floats : c->off_b2, s));
  }
  return 0;
}
#undef DVT_TRY

}  // namespace

extern "C" int dvt_fit_run(const DvtFitConfig* c, const DvtFitBuffers* b, int step_begin,
                           int step_end, void* stream) {
  return dvt_fit_run_batched(c, 1, &b, step_begin, step_end, stream);
}

// k independent fits (k images, ONE configuration and learning-rate schedule) advanced in lock
// step by shared launches (BASELINE.json configs[2]: many concurrent neural fields per GPU).
// Results are those of k separate dvt_fit_run calls: every fit keeps its own arena, Adam state,
// index stream and workspace; only the grids are fused.
extern "C" int dvt_fit_run_batched(const DvtFitConfig* c, int k, const DvtFitBuffers* const* bufs,
                                   int step_begin, int step_end, void* stream) {
  if (k <= 0 || k > DVT_FIT_BATCH_MAX || !bufs) return DVT_E_BADARG;
  Work w[DVT_FIT_BATCH_MAX];
  for (int j = 0; j < k; ++j) {
    int rc = check_bufs(c, bufs[j], step_begin, step_end);
    if (rc) return rc;
    for (int i = 0; i < j; ++i)
      if (bufs[i]->params == bufs[j]->params || bufs[i]->workspace == bufs[j]->workspace)
        return DVT_E_BADARG;  // fits must not share state
    carve(c, bufs[j]->workspace, &w[j]);
    if (w[j].g_offs != nullptr && step_begin < c->switch_step + 1) {  // lists of the phase-1 steps of this call
      const int last = step_end < c->switch_step + 1 ? step_end : c->switch_step + 1;
      rc = dvt_build_row_lists(bufs[j]->idx + (size_t)step_begin * c->batch, last - step_begin, c->batch,
                               c->lattice, w[j].g_offs + (size_t)step_begin * (c->lattice + 1),
                               w[j].g_perm + (size_t)step_begin * c->batch, (hipStream_t)stream);
      if (rc) return rc;
    }
  }
  // Latched for the whole call: dvt_tune_set(6, .) is process-global and another thread / engine may flip it
  // while these steps are being enqueued; a step that found it newly "on" would run the row kernel on a shadow that was
  // never built (or is stale, since Adam keeps it current only while the step is fused).
  const bool fused = dvt_fit_fused_ok(c) && w[0].shadow != nullptr && w[0].g_offs != nullptr;
  if (fused && step_end > step_begin) {
    // the fp32 arena may have been (re)initialised by the caller: rebuild the bf16 shadow of the MLP
    // weights once; from here on the Adam kernel keeps it current
    DvtShadowLayout L;
    int rc = dvt_shadow_layout(c, &L);
    if (rc) return rc;
    const float* pp[DVT_FIT_BATCH_MAX];
    void* ss[DVT_FIT_BATCH_MAX];
    for (int j = 0; j < k; ++j) {
      pp[j] = bufs[j]->params;
      ss[j] = w[j].shadow;
    }
    rc = dvt_shadow_build_k(&L, k, pp, ss, 0, c->arena_floats, (hipStream_t)stream);
    if (rc) return rc;
  }
  // Sorted grid lists and the lazy Adam depend on the coordinates and the index stream only, not on the operand
  // precision of the MLP: since round 4 the fp32-operand step (the reference's default --dtype float32) uses them too --
  // gather instead of atomics for the grid gradient, and the IEEE replay (bit-identical to the dense sweep) for the
  // fine grid levels instead of streaming 515 MB per step.
  const bool sorted_lists = g_fit_fused_enable_lists(c) && w[0].g_offs != nullptr && w[0].gs_keys != nullptr && g_fit_sorted_grid;
  uint32_t lazy_e0 = 0xffffffffu;
  int lazy_l0 = 0;
  DvtAdamLazy lz{};
  const int n_call = step_end - step_begin;
  const bool lazy = sorted_lists && g_fit_lazy_adam && w[0].lazy_done != nullptr && n_call >= 1 && n_call <= 65535 &&
                    lazy_range(c, &lazy_e0, &lazy_l0);
  if (!lazy) lazy_e0 = 0xffffffffu;
  if (lazy) {
    lz.e0 = lazy_e0;
    lz.n_entries = c->grid.n_entries_total;
    lz.nt = 4 * c->batch;
    lz.l0 = lazy_l0;
    lz.n_levels = c->grid.n_levels;
    lz.beta1 = c->beta1;
    lz.beta2 = c->beta2;
    lz.eps = c->eps;
    lz.weight_decay = c->weight_decay;
    lz.neg_step = w[0].lazy_ns;
    lz.inv_bc2s = w[0].lazy_ib;
    lz.bc2s = w[0].lazy_bc;
    lz.exact = g_fit_lazy_exact || !c->mlp_bf16;  // fp32 operands: the reference's arithmetic, to the bit
    int rc = dvt_adam_lazy_tables(bufs[0]->h_lr, step_begin, step_end, c->beta1, c->beta2, w[0].lazy_ns, w[0].lazy_ib,
                                  w[0].lazy_bc, (hipStream_t)stream);
    if (rc) return rc;
    for (int j = 0; j < k; ++j) {
      lz.p[j] = bufs[j]->params;
      lz.m[j] = bufs[j]->adam_m;
      lz.v[j] = bufs[j]->adam_v;
      lz.g[j] = bufs[j]->grads;
      lz.done[j] = w[j].lazy_done;
      const hipError_t e = hipMemsetAsync(w[j].lazy_done, 0, (size_t)(lz.n_entries - lz.e0) * sizeof(uint16_t),
                                          (hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
    }
  }
  bool merged_catchup = false;  // the previous step's Adam launch already brought this step's entries up to date
  for (int step = step_begin; step < step_end; ++step) {
    int gs_local = -1;
    if (sorted_lists) {
      gs_local = (step - step_begin) % GS_CHUNK;
      if (gs_local == 0) {  // sorted grid-corner lists of the next GS_CHUNK steps, from the resident index stream
        const int steps = step_end - step < GS_CHUNK ? step_end - step : GS_CHUNK;
        const float* xy[DVT_FIT_BATCH_MAX];
        const int32_t* ridx[DVT_FIT_BATCH_MAX];
        uint32_t* keys[DVT_FIT_BATCH_MAX];
        uint16_t* pay[DVT_FIT_BATCH_MAX];
        float* ww[DVT_FIT_BATCH_MAX];
        uint32_t* uk[DVT_FIT_BATCH_MAX];
        int32_t* uc[DVT_FIT_BATCH_MAX];
        for (int j = 0; j < k; ++j) {
          xy[j] = bufs[j]->xy;
          ridx[j] = bufs[j]->idx + (size_t)step * c->batch;
          keys[j] = w[j].gs_keys;
          pay[j] = w[j].gs_pay;
          ww[j] = w[j].gs_w;
          uk[j] = w[j].gs_ukeys;
          uc[j] = w[j].gs_ucount;
        }
        int rc = dvt_grid_sort_k(&c->grid, k, xy, ridx, c->batch, steps, keys, pay, ww, (hipStream_t)stream,
                                 lazy ? uk : nullptr, lazy ? uc : nullptr);
        if (rc) return rc;
      }
    }
    if (lazy && step > step_begin && (step - step_begin) % g_fit_lazy_refresh == 0) {
      // REFRESH every g_fit_lazy_refresh steps: all lazy entries through step - 1.  A replay is a sequential recurrence
      // (~70 cycles per step), and without a bound the longest gap of a step's entries (~128 ln 8192 steps on the
      // finest level) made the catch-up kernel 74 us long whatever the throughput; with the refresh no chain exceeds
      // GS_CHUNK steps, and the sweep itself runs at full lane utilisation (nearly every lane replays the same
      // GS_CHUNK steps) -- work is only moved, not added.
      int rc = dvt_adam_lazy_k(&lz, k, true, step - step_begin, nullptr, nullptr, (hipStream_t)stream);
      if (rc) return rc;
    } else if (lazy && step > step_begin && !merged_catchup) {  // bring the entries this step reads up to date
      const uint32_t* uk[DVT_FIT_BATCH_MAX];
      const int32_t* uc[DVT_FIT_BATCH_MAX];
      for (int j = 0; j < k; ++j) {
        uk[j] = w[j].gs_ukeys + (size_t)gs_local * c->grid.n_levels * 4 * c->batch;
        uc[j] = w[j].gs_ucount + (size_t)gs_local * c->grid.n_levels;
      }
      int rc = dvt_adam_lazy_k(&lz, k, false, step - step_begin, uk, uc, (hipStream_t)stream);
      if (rc) return rc;
    }
    // the catch-up of step + 1 rides in this step's Adam launch when its lists exist already (same chunk) and step + 1
    // is not a refresh step
    const int rel_next = step + 1 - step_begin;
    merged_catchup = lazy && g_fit_lazy_merge && step + 1 < step_end && gs_local + 1 < GS_CHUNK &&
                     rel_next % g_fit_lazy_refresh != 0;
    int rc = fit_step(c, k, bufs, w, step, gs_local, lazy_e0, merged_catchup ? &lz : nullptr, rel_next, fused, (hipStream_t)stream);
    if (rc) return rc;
  }
  if (lazy) {  // the arena is exact at every call boundary: every lazy entry through the last step of this call
    int rc = dvt_adam_lazy_k(&lz, k, true, n_call, nullptr, nullptr, (hipStream_t)stream);
    if (rc) return rc;
  }
  return 0;
}
