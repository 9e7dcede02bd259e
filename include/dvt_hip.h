/*
 * dvt_hip.h -- C ABI of libdvt_hip.so: the MI355X (gfx950) implementation of the
 * DVT stage-1 per-image denoising hot path.
 *
 * Every entry point replaces one native/GPU boundary of the reference
 * (Jiawei-Yang/Denoising-ViT, paths relative to the reference root):
 *
 *   dvt_grid_*      tinycudann `tcnn.Encoding(HashGrid)` fwd/bwd, the only native FFI on the
 *                   reference's path -- dvt/models/neural_feature_field.py:25-39 (ctor), :48 (call)
 *   dvt_linear_*    the `nn.Sequential(Linear, ReLU, Linear)` field MLP and the residual
 *                   predictor -- neural_feature_field.py:40-44,:49; offline_denoiser.py:40-46,:107
 *   dvt_gather_rows / dvt_bilinear_rows_*   `F.grid_sample(shared_artifacts, coords)` -- offline_denoiser.py:96-102
 *   dvt_loss_*      reconstruction composition + 4 loss terms -- offline_denoiser.py:113-140
 *   dvt_adam_step   `torch.optim.Adam(lr, eps=1e-15, wd, betas=(0.9,0.99))` + zero_grad
 *                   -- main_img_denoising.py:48-54, :87-89
 *   dvt_fit_run     the whole inner loop of denoise_an_image -- main_img_denoising.py:67-89
 *   dvt_field_infer final full-image inference F(lattice) -- main_img_denoising.py:121-130
 *   dvt_vit_*       frozen ViT forward_intermediates (timm) -- dvt/models/vit_wrapper.py:122-143,
 *                   main_img_denoising.py:317-323
 *
 * Conventions
 *   - all functions return int: 0 = ok, >0 = hipError_t, <0 = DVT_E_* argument error;
 *     they never throw and never allocate device memory;
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch) unless the name
 *     starts with `h_` / the doc says host;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - the caller selects the device (hipSetDevice); functions are re-entrant on
 *     distinct streams; row-major fp32 unless stated.
 */
#ifndef DVT_HIP_H
#define DVT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVT_MAX_LEVELS 32

#define DVT_E_BADARG (-1) /* shape/alignment precondition violated */
#define DVT_E_NOTIMPL (-2)

/* ABI version; bumped on any struct or entry-point change (2: dvt_fit_run_batched replaces
 * dvt_fit_run_multi; dvt_render_views; dvt_vit_gemm_residual). */
#define DVT_ABI_VERSION 2
int dvt_abi_version(void);
/* HOST: sizeof() of {DvtGridTable, DvtAdamSeg, DvtAdamArgs, DvtFitConfig, DvtFitBuffers} so that
 * a foreign-language binding (ctypes) can verify its struct mirrors. */
int dvt_struct_sizes(int64_t* h_out5);

/* ------------------------------------------------------------------------------------
 * Hash grid (tcnn HashGrid, 2-D, linear interpolation, CoherentPrime hash)
 * ---------------------------------------------------------------------------------- */

/* Per-level table, computed ONCE on the host in fp32 (tcnn grid.h: grid_scale /
 * grid_resolution / offset table) and passed by value to every kernel.            */
typedef struct DvtGridTable {
  int32_t n_levels;
  int32_t n_features; /* features per level; this build supports 8 (tcnn config :30) */
  uint32_t n_entries_total; /* sum of entries over levels; params = n_entries_total * n_features */
  uint32_t pad_;
  float scale[DVT_MAX_LEVELS];        /* exp2f(l*log2f(pls))*base - 1 */
  uint32_t resolution[DVT_MAX_LEVELS]; /* ceilf(scale)+1 */
  uint32_t entries[DVT_MAX_LEVELS];    /* min(round_up(res^2,8), 2^log2_T) */
  uint32_t offset[DVT_MAX_LEVELS];     /* cumulative entries */
  uint32_t hashed[DVT_MAX_LEVELS];     /* 1 if res^2 > entries (only then the hash is used) */
} DvtGridTable;

/* HOST function: fill `out` for (n_levels, base_resolution, max_resolution, log2_hashmap_size).
 * per_level_scale = exp((ln max - ln base)/(L-1)) evaluated in float64 then narrowed to
 * fp32 exactly as neural_feature_field.py:34-36 -> tcnn json float.                  */
int dvt_grid_table(int n_levels, int n_features, int base_resolution, int max_resolution,
                   int log2_hashmap_size, DvtGridTable* h_out);

/* enc[n, L*F] (column = level*F + f) from xy[n,2] in [0,1].  Replaces tcnn fwd. */
int dvt_grid_fwd(const DvtGridTable* h_tbl, const float* xy, const float* params, float* enc,
                 int n, void* stream);

/* d_params[idx*F+f] += w * d_enc[level*F+f] (fp32 atomics) into a buffer the caller keeps
 * zeroed between steps; when `touched` != NULL also sets bit (entry) of the bitmap
 * touched[entry>>5] so that dvt_adam_step can skip reading/clearing untouched gradients.
 * Replaces tcnn bwd (dL/dparams only; coords never require grad on this path).        */
int dvt_grid_bwd(const DvtGridTable* h_tbl, const float* xy, const float* d_enc, float* d_params,
                 uint32_t* touched, int n, void* stream);

/* Debug/parity: the 4 corner entry indices (absolute, incl. level offset) and weights for
 * every (sample, level): idx[n, L, 4] uint32, w[n, L, 4] fp32. Integer path is bit-exact
 * against the oracle.                                                                   */
int dvt_grid_corners(const DvtGridTable* h_tbl, const float* xy, uint32_t* idx, float* w, int n,
                     void* stream);

/* ------------------------------------------------------------------------------------
 * fp32 linear layers on f32-input MFMA (exact fp32: v_mfma_f32_32x32x2_f32)
 * ---------------------------------------------------------------------------------- */

/* y[m,n] = act(x[m,k] . w[n,k]^T + b[n]);  relu != 0 applies ReLU. b may be NULL.
 * Preconditions: k % 4 == 0, n % 4 == 0 (else DVT_E_BADARG).                       */
int dvt_linear_fwd(const float* x, const float* w, const float* b, float* y, int m, int n, int k,
                   int relu, void* stream);

/* Backward of the above for upstream grad dy[m,n]:
 *   dw[n,k] += dy^T . x   (atomic accumulation into a zeroed buffer; split over m)
 *   db[n]   += colsum(dy) (db may be NULL)
 *   dx[m,k]  = dy . w, multiplied by (relu_mask[m,k] > 0) when relu_mask != NULL
 *              (relu_mask = the ReLU output that produced x); dx may be NULL.       */
int dvt_linear_bwd(const float* dy, const float* x, const float* w, float* dw, float* db,
                   float* dx, const float* relu_mask, int m, int n, int k, void* stream);

/* ------------------------------------------------------------------------------------
 * Row gathers / lattice artifact map G
 * ---------------------------------------------------------------------------------- */

/* dst[i, :] = src[idx[i] % modulo, :] (modulo <= 0: no modulo). c % 4 == 0.          */
int dvt_gather_rows(const float* src, const int32_t* idx, float* dst, int n, int c, int modulo,
                    void* stream);

/* dst[idx[i] % modulo, :] += src[i, :] (fp32 atomics).                              */
int dvt_scatter_add_rows(const float* src, const int32_t* idx, float* dst, int n, int c,
                         int modulo, void* stream);

/* General form of offline_denoiser.py:96-102 for arbitrary coords (module API):
 * out[i,:] = grid_sample(G, coords[i], bilinear, zeros padding, align_corners=True) with the
 * map stored row-major G_rows[H*W, c]; coords[n,2] = (x, y) in [-1,1]. c % 4 == 0.   */
int dvt_bilinear_rows_fwd(const float* G_rows, const float* coords, float* out, int n, int c,
                          int H, int W, void* stream);
/* d_G_rows[H*W, c] += scatter of d_out[n, c] with the same 4 weights (fp32 atomics). */
int dvt_bilinear_rows_bwd(const float* d_out, const float* coords, float* d_G_rows, int n, int c,
                          int H, int W, void* stream);

/* ------------------------------------------------------------------------------------
 * Loss (offline_denoiser.py:113-140) forward + backward in one pass
 * ---------------------------------------------------------------------------------- */

/* Per row i (n rows, c channels):
 *   raw = raw_rows[i]            (already gathered, [n,c])
 *   g   = G[g_idx[i] % lattice]  (G stored [lattice, c]; g_idx may be NULL -> row i)
 *   pred = F[i] + g (+ Hres[i] when Hres != NULL, treated as detached)
 *   loss = mse(pred, raw) + 1 - mean_i cos(pred_i, raw_i)
 *          (+ 0.1*mse(Hres, raw - F - g) + 0.02*mean|Hres| when Hres != NULL)
 * Writes d_pred[n,c] = grad_scale * dloss/dpred, d_hres[n,c] (when Hres != NULL) and the
 * per-row partial sums row_sums[n,8] = {sse, cos, res_sse, res_abs, 0...}.
 * d_pred / d_hres may be NULL (inference: only row_sums).                           */
int dvt_loss_fwd_bwd(const float* F, const float* G, const int32_t* g_idx, int lattice,
                     const float* Hres, const float* raw_rows, float* d_pred, float* d_hres,
                     float* row_sums, int n, int c, float grad_scale, void* stream);

/* out[0..4] = {loss, patch_l2_loss, cosine_similarity_loss, residual_loss,
 * residual_sparsity_loss} from row_sums (one small reduction launch).               */
int dvt_loss_reduce(const float* row_sums, float* out5, int n, int c, int with_residual,
                    void* stream);

/* ------------------------------------------------------------------------------------
 * Fused dense Adam + zero_grad over a flat parameter arena
 * ---------------------------------------------------------------------------------- */

#define DVT_ADAM_MAX_SEGS 8
typedef struct DvtAdamSeg {
  int64_t begin; /* float offset in the arena, multiple of 256 */
  int64_t end;   /* float offset (exclusive), multiple of 256 (pad with zeros) */
  double lr;     /* learning rate of this step (python float in the reference) */
  double bias_correction1;      /* 1 - beta1^t, t = this tensor group's own step count */
  double bias_correction2_sqrt; /* sqrt(1 - beta2^t) */
  int32_t active;               /* 0: skipped entirely (grad None in torch) */
  int32_t pad_;
} DvtAdamSeg;

typedef struct DvtAdamArgs {
  double beta1, beta2, eps, weight_decay; /* python floats in the reference; narrowed like torch does */
  int32_t n_segs;
  int32_t pad_;
  /* floats [0, sparse_end) carry a SPARSE gradient gated by `touched` (1 bit per 8 floats,
   * one 32-bit word per 256 floats); beyond it gradients are dense.  sparse_end % 256 == 0. */
  int64_t sparse_end;
  DvtAdamSeg segs[DVT_ADAM_MAX_SEGS];
} DvtAdamArgs;

/* torch.optim.Adam semantics (L2 decay folded into the gradient, lerp first moment,
 * eps added after the bias-corrected sqrt), g is consumed and cleared (zero_grad).
 * 24 B/param of HBM traffic for untouched sparse-segment params.                    */
int dvt_adam_step(const DvtAdamArgs* h_args, float* p, float* m, float* v, float* g,
                  uint32_t* touched, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused fit loop (the product fast path)
 * ---------------------------------------------------------------------------------- */

typedef struct DvtFitConfig {
  int32_t feat_dim;     /* C */
  int32_t hidden;       /* field MLP hidden = C/2 */
  int32_t res_hidden;   /* residual predictor hidden = C/4 */
  int32_t lattice;      /* H*W of the shared-artifact map G */
  int32_t n_rows;       /* rows in the feature store = (views+1)*lattice */
  int32_t batch;        /* pixel_bsz */
  int32_t num_iters;
  int32_t switch_step;  /* int(freeze_shared_artifacts_after*num_iters); phase 2 iff step > switch_step */
  int32_t enable_residual; /* enable_residual_predictor */
  int32_t mlp_bf16;        /* 1: the MLP GEMMs of the loop round their operands to bf16 (fp32 accumulate,
                            * fp32 outputs) = the reference's `--dtype bfloat16` autocast mode; 0: fp32 */
  double grad_scale;    /* 1024: GradScaler quirk, main_img_denoising.py:55,:88 */
  double beta1, beta2, eps, weight_decay;
  DvtGridTable grid;
  /* arena layout (float offsets, each a multiple of 256) */
  int64_t off_grid, off_w1, off_b1, off_w2, off_b2, off_G, off_wh1, off_bh1, off_wh2, off_bh2,
      off_wh3, off_bh3, arena_floats;
} DvtFitConfig;

typedef struct DvtFitBuffers {
  const float* feat;   /* [n_rows, C] feature store (NHWC patch tokens of all views) */
  const float* xy;     /* [n_rows, 2] global pixel coords in [0,1] */
  const int32_t* idx;  /* [num_iters, batch] row index stream (host-generated, np.random.randint) */
  float* params;       /* arena */
  float* adam_m;
  float* adam_v;
  float* grads;        /* arena-shaped, must be zero on entry of step 0 */
  uint32_t* touched;   /* bitmap, (off_w1/256) words, zero on entry */
  float* workspace;    /* dvt_fit_workspace_floats() floats */
  float* losses;       /* [num_iters, 8] written only for steps where (step % log_every == 0) or last */
  const double* h_lr;  /* HOST: lr per step (misc.adjust_learning_rate) */
  int32_t log_every;   /* 0: never */
  int32_t pad_;
} DvtFitBuffers;

/* HOST: fill the arena offsets of cfg from its dims (layout documented in DESIGN.md). */
int dvt_fit_layout(DvtFitConfig* h_cfg);
int64_t dvt_fit_workspace_floats(const DvtFitConfig* h_cfg);

/* Enqueue steps [step_begin, step_end) of the inner loop on `stream` (asynchronous). */
int dvt_fit_run(const DvtFitConfig* h_cfg, const DvtFitBuffers* h_bufs, int step_begin,
                int step_end, void* stream);

/* k independent fits (k images) of ONE configuration advanced in lock step by SHARED launches:
 * every kernel of a step covers all k fits (BASELINE.json configs[2]: many concurrent neural
 * fields per GPU).  A fit step is a chain of ~10 small dependent launches, each paying a fixed
 * launch latency that does not grow with the grid.  Every fit keeps its own arena, Adam state,
 * index stream and workspace (h_bufs[j]); results equal k separate dvt_fit_run calls up to the
 * ordering of fp32 atomics.  1 <= k <= DVT_FIT_BATCH_MAX; h_lr is read from h_bufs[0]. */
#define DVT_FIT_BATCH_MAX 4
int dvt_fit_run_batched(const DvtFitConfig* h_cfg, int k, const DvtFitBuffers* const* h_bufs,
                        int step_begin, int step_end, void* stream);

/* out[n, C] = field(xy[n,2]) using arena params; workspace >= n*(L*F + hidden) floats. */
int dvt_field_infer(const DvtFitConfig* h_cfg, const float* params, const float* xy, float* out,
                    float* workspace, int n, void* stream);

/* ------------------------------------------------------------------------------------
 * View synthesis (dvt/dataset/transform.py:48-52, :70; single_image_dataset.py:33-38)
 * ---------------------------------------------------------------------------------- */
/* out[v, 3, OH, OW] = hflip?(resize_bicubic_antialias(img[3, top:top+h, left:left+w] -> OH x OW)) for
 * boxes[v] = {top, left, h, w, flip} (int32, device).  The box must lie inside the H x W image and
 * h / OH, w / OW <= 3.5 (16 filter taps).  Semantics of torch/torchvision's anti-aliased bicubic
 * (a = -0.5, align_corners = False, windows truncated at the crop border). */
int dvt_render_views(const float* img, int H, int W, const int32_t* boxes, float* out, int V, int OH,
                     int OW, void* stream);

/* Schedule selection (developer use).  Every selectable value computes the SAME results (each one
 * is parity-tested); the knobs only trade speed -- with ONE documented exception: key 10 selects the
 * arithmetic of the lazy Adam replay, and its DEFAULT (0: v_rcp_f32 / v_sqrt_f32, 1 ulp each) is a
 * tolerance-tested approximation of the dense sweep, not bit-identical to it; 1 selects the IEEE replay
 * that is.  The state is process-global: set it before any
 * work is enqueued, never concurrently with launches (the compute entry points themselves are
 * re-entrant on distinct streams).
 * key 0 = fp32 GEMM tile configuration override
 * (-1 heuristic, 0: 64x64, 1: 32x64, 2: 32x32, 3: 64x32 per workgroup);
 * key 1 = bf16 ViT extractor.  The PRODUCT library accepts only values under which every entry point still computes its
 *         documented result, and returns DVT_E_BADARG for anything else:
 *           4 [default] / 3 / 1: GEMM schedule -- 256x256 8-phase ring (whole 256-tiles and an even number of 64-deep
 *             k-tiles, else 3) / 256x128 ping-pong (M a whole 256-tile, else 1) / 128x128 two-stage;
 *           values >= 16: KiB of W kept L2-resident per N-tile group of the tile rasterisation; -100 - b: b M panels per
 *             block of the tile order (0 = auto); -50 / -51: non-temporal bf16 output stores off / on;
 *           -60 / -61: LayerNorm as its own kernels / folded into the qkv and fc1 GEMMs [default];
 *           -700 - pct (pct 0..400, default 0 = off): de-synchronised start of the 256x256 kernel's first round of workgroups,
 *             spread over pct % of the modelled tile time (round 5: no gain; results do not depend on it);
 *           -520 / -521 and -522 / -523: inside dvt_vit_forward_f32x3, exact-fp32 attention on / off [off] and split kernels
 *             instead of split epilogues on / off [off];
 *           -502 and -525 (= -510 - 15): the one attention kernel / schedule mask the product contains (accepted, no effect);
 *           -570 [default] / -571 / -572 (round 6): the 256x256 kernel's workgroups touch the first 0 / 1 / 2 k-tiles' operand lines of
 *             the tile the next workgroup of their XCD walks (an L2 prefetch late in the epilogue; measured null, results unaffected);
 *           -531 [default] / -530 (round 6): inside dvt_vit_forward the qkv GEMM writes q * log2(e) / 8 and the log2-domain
 *             attention kernel runs (dvt_vit_attention_log2q, include/dvt_vit.h) / q as it is and dvt_vit_attention.  The two
 *             differ in WHICH bf16 value q rounds to (q against q * 0.18033688: one rounding each), i.e. by the bf16 rounding
 *             noise of the logits, not in error class (tests/test_gpu_vit.py holds both against the fp32 oracle).
 *         None of the others changes a result beyond summation order (fp32 row statistics of the folded LayerNorm, ~1e-7 relative).
 *         Developer builds (-DDVT_LAB, csrc/lab/, include/dvt_vit.h) add: schedules 0, 2 (superseded), 13 (round 5's walk of the
 *         8-phase ring) and 5, 10 (re-schedules of it), 11 (the product's kernel as a persistent workgroup with an overlapped
 *         tile boundary) -- all bit-identical to 4 --, 6..9 (4-wave persistent kernel; 8 / 9 with an approximate GELU),
 *         -200 - n / -600 - n (tiles per workgroup of 6..9 and 11 / grid of 6..9), -300 - n (ablation mask of the selected 4-wave schedule, or timing build of schedule 5:
 *         TIMING ONLY, results wrong by construction; reset by every change of schedule), -501 (round-2 attention loop),
 *         -510 - mask (attention schedule masks; any of these also selects q as it is), -540 - x (the log2-domain attention kernel with bits x of
 *         its schedule mask toggled: 2 / 512 / 514 = P.V fragment by fragment / K reads unplaced / both; 128 / 256 / 384 = ablation
 *         builds: idle waves not skipped / no half tail tile / neither; -540 = the product's);
 * key 18 = stage-2 trainer (include/dvt_stage2.h), mask 0..63, 63 = default: bit 0 / 1 / 2 = forward / data-gradient / weight-gradient
 *         GEMMs of the linear layers on the 128 x 128 x 32 tile, bit 3 = softmax fused into the attention products, bit 4 = softmax
 *         backward without a dP pass, bit 5 = a layer's weight gradient on a side stream beside its data gradient (0 = round 5's
 *         flow); results differ in summation order only (tests/test_gpu_stage2.py);
 * key 2 = grid backward: levels with more entries than `value` use global atomics (default 0 = all);
 * key 5 = fp32 GEMM k-depth of the register-staged kernel: 64 (default), 32, or 16 (10 KB LDS per
 *         workgroup, lets fit kernels co-reside with the ViT extractor's 136-144 KB workgroups);
 * key 4 = fp32 GEMM: 1 (default) use the 3-stage LDS-DMA kernel when eligible, 0 = register-staged only;
 *         2 / 3 = LDS stages of the stage-2 GEMMs (2, default: two workgroups per CU);
 *         11 (default) / 10 = the fp32 extractor's linear layers on the 128 x 128 x 32 tile with fused GELU / residual
 *         epilogues (round 5) / on the 64 x 64 x 64 LDS-DMA kernel + separate row-local passes (results differ in summation
 *         order only);
 * key 6 = fit step: 1 (default) the fused row kernel (dvt_fit_fused.hip), 0 = one launch per layer (both operand
 *         precisions); 3 (default) / 2 = the same switch for the fp32-operand mode only (round 5: fp32 row kernel on
 *         v_mfma_f32_16x16x4_f32 where its LDS images fit, feat_dim 384 / 768; results agree with the layer-by-layer launches to
 *         summation order -- tested bound: per-step losses within 5e-5 relative, mean parameter difference 2e-6 after 24 steps
 *         (tests/test_gpu_fit.py) --, the same 200 us per step alone; the choice is latched once per dvt_fit_run[_batched] call; dvt_amd.stage1 selects 2 beside the fp32 extractor, profiles/r05/r05f_*);
 * key 7 = fit step (both operand precisions since round 4): 1 (default) hash-grid gradient gathered from per-step sorted
 *         corner lists, 0 = scattered with atomics;
 * key 9 = fit step (both precisions): 1 (default) lazy Adam over the fine hash-grid levels -- with fp32 operands always the
 *         IEEE replay of key 10 = 1: the ADAM ARITHMETIC is bit-identical to the dense sweep (never-touched entries end
 *         equal bit for bit, tests/test_gpu_fit.py::test_fp32_lazy_adam_vs_dense_sweep); touched entries carry the summation
 *         order of the gathered grid gradient, run-to-run rounding noise in either mode --, 0 = dense Adam over the whole arena,
 *         n >= 2 = lazy with a full refresh every n steps (default 32);
 * key 16 = fit step with 2 or 4 fits per launch (dvt_fit_run_batched): 1 (default, round 6) the row kernel maps fit f to the XCDs
 *         [8 f / k, 8 (f + 1) / k) so that an XCD's L2 streams ONE fit's weights, 0 = the plain (row block, fit) grid; same results;
 * key 12 = the merged Adam launch also stores the bf16 weight shadow (1, default) or shadow_build_kernel runs (0);
 * key 11 = the lazy catch-up of the next step shares the Adam launch (1, default) or is its own launch (0);
 * key 10 = lazy Adam replay arithmetic: 0 (default) v_rcp_f32 / v_sqrt_f32 (1 ulp each), an APPROXIMATION of the
 *         reference's Adam on the fine grid levels: agrees with the dense sweep to rounding over ~150 steps,
 *         decorrelates in the unread weight-decay jitter over thousands, and matches the CPU oracle over the whole
 *         1000-step schedule exactly as well as the IEEE mode does (per-patch cosine 0.99994 / 0.9992, fixture test);
 *         1: IEEE division / square root, the dense kernel's own update function -- every entry ends bit-identical
 *         to the dense sweep; +18 us per step = -5.4 % bench `value` in a same-box A/B (profiles/r03);
 * key 13 = fused row kernel: rows per workgroup -- 1 (default) 32 when k >= 4 fits share the launch and the LDS images fit
 *         (C <= 768; C = 1024 in phase 1), else 16; 0 = always 16; 2 = 32 whenever the images fit.  Same arithmetic per row;
 * key 14 = fit step, bf16 operands: workgroup shape of fit_rows / fit_backward -- 0 (default) 8 / 16 waves, 1 = 4 / 8 waves (one /
 *          two waves per SIMD, the footprint one attention workgroup of the extractor leaves; same arithmetic, same results);
 * key 8 = Adam sweeps the arena in alternating directions on consecutive steps (1, default) or always forward (0);
 * key 3 = Adam zero-writes the whole sparse gradient region every step (1, default) or only touched entries (0);
 *         10 (default) / 11 = the dense sweep's p / m / v streams without / with the non-temporal hint (round 5: measured
 *         no effect on the pipelined rate; same results). */
int dvt_tune_set(int key, int value);

/* ------------------------------------------------------------------------------------
 * In-library profiling probes: hipEventRecord pairs on the LAUNCH stream around selected
 * kernel launches, so that bench.py can report per-kernel durations measured inside its
 * timed region (torch.cuda.Event only sees torch's current stream).
 * ---------------------------------------------------------------------------------- */
#define DVT_PROBE_ADAM 0      /* fused Adam launches; work = algorithmic bytes (24 B/param + grads) */
#define DVT_PROBE_VIT_GEMM 1  /* bf16 MFMA GEMM launches of the ViT; work = ALGORITHMIC flops: 2 * real tokens
                               * (1370/image, not the rows padded to 1408) * N * real K (588 for the patch embedding) */
#define DVT_PROBE_VIT_ATTN 2  /* attention launches; work = 4*S*S*64*heads*batch flops */
#define DVT_PROBE_FIT_GEMM 3  /* fp32 MFMA linear fwd/bwd launches; work = flops */
#define DVT_PROBE_GRID 4      /* hash-grid fwd+bwd launches; work = algorithmic bytes */
#define DVT_PROBE_FIT_ROWS 5  /* fused row kernel of the fit (forward + loss + dgrad); work = its MLP flops */
#define DVT_N_PROBES 8
/* HOST: enable probes whose bit is set in mask (0 disables all); resets accumulated samples. */
int dvt_prof_enable(unsigned mask);
/* HOST: synchronises the recorded events of `probe` and returns the summed duration (ms), the
 * number of launches sampled and the summed work units; clears the probe's samples.      */
int dvt_prof_collect(int probe, double* h_total_ms, int64_t* h_count, double* h_work);

#ifdef __cplusplus
}
#endif
#endif /* DVT_HIP_H */
