/*
 * dvt_vit.h -- C ABI of the frozen-ViT feature extractor in libdvt_hip.so (gfx950).
 *
 * Replaces, for the DINOv2 ViT-B/14 and ViT-L/14 backbones of BASELINE.json, the timm call
 *   vit.get_intermediate_layers(x, n=[layer_index], reshape=True)[-1].permute(0, 2, 3, 1)
 * of the reference (dvt/models/vit_wrapper.py:122-143 -> timm 1.0.7
 * VisionTransformer.forward_intermediates; call sites main_img_denoising.py:317-323, :332-336):
 * patch embedding (Conv2d k=patch, stride) -> + pos_embed, cls token -> n_blocks x
 * {LN, MHA, LayerScale, residual, LN, MLP(GELU), LayerScale, residual} -> final LN ->
 * drop the cls token -> patch-token map written NHWC fp32 straight into the feature store
 * (the NCHW round trip of the reference, vit_wrapper.py:142 / main_img_denoising.py:323, is
 * omitted).
 *
 * Arithmetic: bf16 operands on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; residual
 * stream, LayerNorm statistics, softmax and all epilogues in fp32 (== the reference's
 * `--dtype bfloat16` autocast mode).  Same conventions as dvt_hip.h (int return codes,
 * device pointers owned by the caller, hipStream_t as void*).
 */
#ifndef DVT_VIT_H
#define DVT_VIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVT_VIT_MAX_DEPTH 48

typedef struct DvtVitConfig {
  int32_t dim;       /* 768 (B) / 1024 (L); multiple of 128 */
  int32_t depth;     /* 12 / 24 */
  int32_t heads;     /* dim / 64: head_dim is fixed to 64 */
  int32_t mlp_dim;   /* 4 * dim */
  int32_t patch;     /* 14 */
  int32_t stride;    /* conv stride (14; the reference's --stride_size) */
  int32_t img_h, img_w; /* 518 x 518 */
  int32_t grid_h, grid_w; /* (img - patch) / stride + 1 = 37 */
  int32_t n_tokens;  /* n_prefix + grid_h * grid_w = 1370 */
  int32_t s_pad;     /* token rows per image: >= n_tokens, a multiple of 32 for dvt_vit_forward and dvt_vit_forward_f32 (dvt_amd
                      * uses 1376 for 1370 tokens since round 6), of 128 for the bf16x3 forward; dvt_vit_config writes the next
                      * multiple of 128 */
  int32_t k_patch;   /* 3 * patch * patch padded to a multiple of 64 (588 -> 640) */
  int32_t n_prefix;  /* prefix tokens: 1 (cls) + register tokens (4 for the *_reg4_* models) */
  float ln_eps;      /* 1e-6 */
  int32_t pos_has_cls; /* 1: pos_embed[0] belongs to cls, patches follow (DINOv2); 0: pos_embed covers
                        * the patch tokens only (timm no_embed_class=True, the reg4 models) */
} DvtVitConfig;

/* All matrices bf16 row-major [out, in] (nn.Linear layout), vectors fp32. */
typedef struct DvtVitBlockWeights {
  const float* norm1_w; const float* norm1_b;
  const void* qkv_w;  const float* qkv_b;   /* [3*dim, dim] */
  const void* proj_w; const float* proj_b;  /* [dim, dim] */
  const float* ls1;                         /* LayerScale gamma [dim] */
  const float* norm2_w; const float* norm2_b;
  const void* fc1_w;  const float* fc1_b;   /* [mlp_dim, dim] */
  const void* fc2_w;  const float* fc2_b;   /* [dim, mlp_dim] */
  const float* ls2;
  /* Optional (bf16 path; all six or none): LayerNorm folded into the GEMMs.  With W' = bf16(norm.weight (.) W)
   * [out, in], cs[n] = sum_k W'[n][k] (fp32) and b' = b + W norm.bias, the qkv / fc1 GEMMs read bf16(x) directly and
   * correct their accumulators with the row's (mean, rstd); the LayerNorm kernels of the block are not launched. */
  const void* qkv_wf; const float* qkv_cs; const float* qkv_bf;   /* folded with norm1 */
  const void* fc1_wf; const float* fc1_cs; const float* fc1_bf;   /* folded with norm2 */
} DvtVitBlockWeights;

typedef struct DvtVitWeights {
  const void* patch_w;    /* bf16 [dim, k_patch], k = c*patch*patch + ky*patch + kx, zero padded */
  const float* patch_b;   /* [dim] */
  const float* cls_token; /* [n_prefix, dim]: cls token, then the register tokens */
  const float* pos_embed; /* [pos_has_cls + grid_h * grid_w, dim] (already resampled to grid_h x grid_w) */
  const float* norm_w; const float* norm_b; /* final LayerNorm */
  DvtVitBlockWeights blocks[DVT_VIT_MAX_DEPTH];
} DvtVitWeights;

/* HOST: derive grid/n_tokens/s_pad/k_patch/heads/mlp_dim from (dim, depth, patch, stride, img). */
int dvt_vit_config(int dim, int depth, int patch, int stride, int img_h, int img_w,
                   DvtVitConfig* h_out);
/* Same with register tokens (vit_wrapper.py:27-30, the *_reg4_dinov2 models: n_reg_tokens = 4). */
int dvt_vit_config_reg(int dim, int depth, int patch, int stride, int img_h, int img_w,
                       int n_reg_tokens, DvtVitConfig* h_out);
/* HOST: bytes of scratch needed for a forward of `batch` images. */
int64_t dvt_vit_workspace_bytes(const DvtVitConfig* h_cfg, int batch);
int dvt_vit_struct_sizes(int64_t* h_out3); /* {DvtVitConfig, DvtVitBlockWeights, DvtVitWeights} */

/* img [batch, 3, img_h, img_w] fp32 (already normalised) -> feat [batch, grid_h, grid_w, dim] fp32
 * = final-LayerNorm'ed patch tokens after blocks[0 .. n_blocks-1] (n_blocks = layer_index + 1).
 * `workspace` must be zero-filled ONCE by the caller before its first use (padding rows).   */
int dvt_vit_forward(const DvtVitConfig* h_cfg, const DvtVitWeights* h_w, const float* img,
                    float* feat, int batch, int n_blocks, void* workspace, void* stream);

/* The same forward with fp32 operands everywhere = the reference's DEFAULT `--dtype float32`
 * (main_img_denoising.py:173, :299: autocast disabled).  `h_w` is a DvtVitWeights whose matrices are fp32
 * ([out, in] row-major, patch_w [dim, k_patch] zero padded) instead of bf16.  Exact-fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate): seconds per image; it exists so that the flag means what
 * the reference means.  Workspace: dvt_vit_workspace_bytes_f32 (no zero-fill needed).                  */
int64_t dvt_vit_workspace_bytes_f32(const DvtVitConfig* h_cfg, int batch);
int dvt_vit_forward_f32(const DvtVitConfig* h_cfg, const DvtVitWeights* h_w, const float* img, float* feat,
                        int batch, int n_blocks, void* workspace, void* stream);
/* Opt-in "bf16x3" arithmetic for the LINEAR layers of the fp32 forward -- what torch.set_float32_matmul_precision("high")
 * permits for fp32 matmuls ("bfloat16_3x"); the reference never sets it, so `--dtype float32` keeps meaning
 * dvt_vit_forward_f32 unless the caller asks for this.  Every fp32 operand is split into two bf16 (x = hi + lo, 2^-17
 * relative), and ONE bf16 GEMM over the K-concatenated operands [hi | hi | lo] . [hi | lo | hi]^T accumulates
 * a_hi w_hi + a_hi w_lo + a_lo w_hi in fp32: ~1e-5 relative per product instead of 6e-8, at 3 x the bf16 flops on the
 * 2.5 PF/s pipe instead of the 157 TF/s fp32 one.  The attention products go the same way (dvt_vit_attention_x3).
 * LayerNorm, softmax, GELU and the residual stream stay fp32 as in dvt_vit_forward_f32.
 *   dvt_vit_split3: x fp32 [rows, k] -> out3 bf16 [rows, 3k]; weights = 1: [hi | lo | hi] (nn.Linear weights, once at
 *     load time), else [hi | hi | lo] (activations; gelu = 1 applies nn.GELU() first).
 *   dvt_vit_linear_f32x3: y = x . W^T + b with x, y fp32; scratch3 = bf16 [m, 3k]; m % 128 == n % 128 == k % 64 == 0.
 *   dvt_vit_forward_f32x3: `h_w` matrices = bf16 [out, 3 * in] from dvt_vit_split3(weights = 1) (patch_w: [dim,
 *     3 * k_patch], zero padded BEFORE the split); workspace dvt_vit_workspace_bytes_f32x3, zero-filled once.        */
int dvt_vit_split3(const float* x, void* out3, long long rows, int k, int weights, int gelu, void* stream);
int dvt_vit_linear_f32x3(const float* x, const void* w3, const float* b, float* y, void* scratch3, int m, int n, int k,
                         void* stream);
int dvt_vit_gemm_f32out(const void* a_bf16, const void* w_bf16, const float* b, float* y, int m, int n, int k,
                        void* stream);
/* Attention of the same mode: qkv fp32 [batch*s_pad, 3*heads*64] -> out fp32 [batch*s_pad, heads*64].  q, k, v are split
 * into (hi, lo) bf16 pairs (v transposed per head) in `scratch` (dvt_vit_attention_x3_scratch_bytes), then
 * S = K_lo Q_hi + K_hi Q_lo + K_hi Q_hi and O += V_lo P_hi + V_hi P_lo + V_hi P_hi on the bf16 matrix pipe with the
 * probabilities split in registers; softmax statistics and accumulation fp32.  dvt_vit_forward_f32x3 uses it unless
 * dvt_tune_set(1, -520) selects the exact-fp32 attention kernel (dvt_vit_attention_f32).                                */
int64_t dvt_vit_attention_x3_scratch_bytes(int batch, int heads, int s_pad);
int dvt_vit_attention_x3(const float* qkv, float* out, void* scratch, int batch, int heads, int s_pad, int n_valid,
                         void* stream);
/* The split-output epilogues dvt_vit_forward_f32x3 uses by default (dvt_tune_set(1, -522): fp32 outputs + split kernels):
 *   dvt_vit_gemm_gelu_x3: out3 bf16 [m, 3n] = [hi | hi | lo] of GELU(a . w^T + b), the next linear layer's A operand;
 *   dvt_vit_gemm_qkv_x3:  q | k (hi, lo) and V^T (hi, lo) written straight into the attention scratch; m must be
 *     batch * s_pad rounded up to 256;  dvt_vit_attention_x3_presplit then runs the attention kernel on that scratch. */
int dvt_vit_gemm_gelu_x3(const void* a_bf16, const void* w_bf16, const float* b, void* out3, int m, int n, int k,
                         void* stream);
int dvt_vit_gemm_qkv_x3(const void* a_bf16, const void* w_bf16, const float* b, void* scratch, int m, int dim, int heads,
                        int s_pad, int batch, int k, void* stream);
int dvt_vit_attention_x3_presplit(const void* scratch, void* out, int batch, int heads, int s_pad, int n_valid,
                                  int split_out /* 0: out fp32 [T, dim]; 1: out bf16 [T, 3 dim] = [hi | hi | lo] */, void* stream);
int64_t dvt_vit_workspace_bytes_f32x3(const DvtVitConfig* h_cfg, int batch);
int dvt_vit_forward_f32x3(const DvtVitConfig* h_cfg, const DvtVitWeights* h_w, const float* img, float* feat,
                          int batch, int n_blocks, void* workspace, void* stream);
/* fp32 attention on qkv [batch*s_pad, 3*heads*64] (q | k | v, head-major inside): out [batch*s_pad, heads*64].  s_pad % 32 == 0
 * (round 6); blocks of 128 queries: where s_pad is not a multiple of 128 the kernel READS up to 96 rows behind an image's rows of
 * qkv as queries (the next image's, or whatever lies behind the last one: nothing of them is stored -- the caller keeps those
 * rows allocated, dvt_vit_workspace_bytes_f32 does); keys are never read behind n_valid's 32-key tile, i.e. inside the image. */
int dvt_vit_attention_f32(const float* qkv, float* out, int batch, int heads, int s_pad, int n_valid, void* stream);

/* ---- building blocks, exported for parity tests ---- */
/* y[m, n] (bf16) = x[m, k] (bf16) . w[n, k]^T (bf16) + b[n]; m % 128 == n % 128 == k % 64 == 0 */
int dvt_vit_gemm_bias(const void* x, const void* w, const float* b, void* y, int m, int n, int k,
                      void* stream);
/* the fc1 GEMM as the extractor launches it: y[m, n] (bf16) = GELU(rstd[m] * (x[m, :] . w'[n, :] - mean[m] * cs[n]) + b'[n]),
 * LayerNorm folded into the weights (ln_stats [m] float2 (mean, rstd), ln_cs [n]; both NULL: plain x . w^T + b);
 * gelu = 0 (and no fold): the bias epilogue.  m % 256 == n % 256 == k % 64 == 0 for the folded form. */
int dvt_vit_gemm_lnfold(const void* x, const void* w, const float* b, void* y, int m, int n, int k,
                        const void* ln_stats, const float* ln_cs, int gelu, void* stream);
/* x[m, n] (fp32, in place) += gamma[n] * (a[m, k] (bf16) . w[n, k]^T (bf16) + b[n]): the attention-proj /
 * fc2 GEMM with the LayerScale + residual epilogue (timm Block.forward: x = x + ls(f(norm(x)))) */
int dvt_vit_gemm_residual(const void* a, const void* w, const float* b, const float* gamma, float* x,
                          int m, int n, int k, void* stream);
/* y (bf16) [rows, dim] = LayerNorm(x fp32 [rows, dim]) * w + b */
int dvt_vit_layernorm(const float* x, const float* w, const float* b, void* y, int rows, int dim,
                      float eps, void* stream);
/* s_pad % 16 == 0.  out[b, s, h*64 + d] (bf16, [batch*s_pad, heads*64]) = softmax(q k^T / 8) v over the first
 * n_valid keys; qk bf16 [batch*s_pad, 2*heads*64] (q then k), vt bf16 [batch, heads, 64, s_pad].  The kernel works in blocks of
 * 128 queries and tiles of 64 keys: where s_pad is not a multiple of 128 it READS up to 127 rows behind image b's rows of qk
 * (the next image's, or whatever lies there: masked keys, unstored queries -- any bit pattern is harmless) -- the caller keeps
 * 128 rows of qk allocated behind the last image (dvt_vit_workspace_bytes does).  vt is never read behind a row's s_pad keys. */
int dvt_vit_attention(const void* qk, const void* vt, void* out, int batch, int heads, int s_pad,
                      int n_valid, void* stream);
/* The same attention on q rows PRE-SCALED by log2(e) / 8 -- qk's q half = bf16(q * 0.18033688...), what dvt_vit_forward's qkv
 * GEMM writes since round 6 (one rounding to bf16, as for the unscaled q; dvt_tune_set(1, -530) restores q as it is +
 * dvt_vit_attention inside dvt_vit_forward, -531 = default): the logits come out of the matrix pipe in units of log2 with the
 * running max already subtracted (the S accumulation starts from -max), so a probability is ONE v_exp_f32 -- the kernel is
 * VALU-issue bound at head_dim 64.  Waves behind an image's rows only stage, a last key tile with <= 32 valid keys is half a
 * tile.  Same layouts, same read-behind contract, same argument checks as dvt_vit_attention.                              */
int dvt_vit_attention_log2q(const void* qk, const void* vt, void* out, int batch, int heads, int s_pad,
                            int n_valid, void* stream);

/* Developer builds (no reference counterpart).  The product library libdvt_hip.so contains only the kernels the extractor
 * launches; superseded schedules, experiments and timing builds live under csrc/lab/ and are compiled only with -DDVT_LAB
 * (tools/build_lab.py -> csrc/libdvt_hip_lab.so, loaded by tools/lab_*.py and tests/test_gpu_lab.py, never by dvt_amd).
 * dvt_vit_is_lab_build: 1 in such a build, 0 in the product library.
 * dvt_vit_debug_buffer: lab builds only (product: DVT_E_BADARG) -- device buffer of (M/256)*(N/256) tiles x 2 wave groups x 24
 * uint32 that the timing builds of the 8p GEMM schedule (dvt_tune_set(1, 5) then dvt_tune_set(1, -300 - build), build 3 or 6..9;
 * EPI_BIAS entry point only; builds 6, 7, 9 are ablations whose results are wrong by construction) fill per workgroup: [0..13]
 * s_memtime stamps of build 3, [16] XCC_ID, [17] HW_ID, [18] entry tick, [19] ticks entry -> last store retired, [20] k-loop
 * ticks, [21] k-tiles.  tools/lab_gemm8p_ablate.py, tools/lab_gemm8p_stamps.py.  NULL switches it off. */
int dvt_vit_is_lab_build(void);
int dvt_vit_debug_buffer(void* dev_u32);

#ifdef __cplusplus
}
#endif
#endif /* DVT_VIT_H */
