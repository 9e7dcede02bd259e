/*
 * dvt_stage2.h -- C ABI of the stage-2 generalizable denoiser in libdvt_hip.so (gfx950).
 *
 * SURVEY.md section 8(f), row N3.  Replaces, for the reference's stage 2,
 *   - `Denoiser.forward` (dvt/models/online_denoiser.py:61-104: + pos_embed, `num_blocks` timm `Block`s --
 *     pre-norm, qkv_bias=True, no LayerScale, LayerNorm eps 1e-6, exact GELU, mlp_ratio 4, head_dim 64), and
 *   - one optimisation step of main_denoiser.py:204-221: forward, `F.mse_loss + (1 - cosine_similarity.mean())`
 *     (:213-217), `loss.backward()` (:220), and `torch.optim.AdamW(betas=(0.9, 0.999))` (:174-178, :221).
 * The torch/cuBLAS/timm ops of that path have no FFI of their own in the reference; these entry points stand
 * where they stand, under the Python mirror `dvt_amd.models.Denoiser` / `python -m dvt_amd.stage2`.
 *
 * Arithmetic: fp32 end to end, like the reference (stage 2 does not autocast): exact-fp32 MFMA GEMMs
 * (v_mfma_f32_32x32x2_f32), attention with materialised fp32 probabilities (kept for the backward pass),
 * two-pass LayerNorm statistics, erf GELU.
 *
 * Data layout: the caller's tensors are the reference's -- `x`, `target`, `pred` are [batch, tokens, dim] fp32,
 * contiguous.  Inside, an image's tokens are padded to `tokens_pad` rows (multiple of 64, rows >= tokens are
 * zero and never contribute).  Parameters, gradients and both AdamW moments are flat fp32 arenas with the
 * layout reported by dvt_s2_param_offsets; the gradient arena is what a data-parallel trainer all-reduces
 * (ONE flat RCCL all-reduce per step; main_denoiser.py:138-140 uses DistributedDataParallel for this).
 *
 * Kernels (round 6; every choice changes summation order only, results are held against autograd by tests/test_gpu_stage2.py):
 * the linear layers' forward, data-gradient (as a forward layer of a transposed weight copy) and weight-gradient GEMMs run the
 * 128 x 128 x 32 exact-fp32 tile; the softmax is fused into the two [tokens_pad][tokens_pad]-sized attention products around it
 * (tokens_pad a multiple of 128, else the GEMM + softmax passes).  Environment switches of the process, read once, for A/B runs
 * only: DVT_S2_BIG=0 / DVT_S2_BIG_BWD=0 / DVT_S2_BIG_WGRAD=0 (64 x 64 tile for forward / data gradient / weight gradient),
 * DVT_S2_ATTN_ROWS=0 (GEMM + softmax passes), DVT_S2_FUSE_SOFTMAX_BWD=0 (with it: round 5's flow), DVT_S2_FORK_WGRAD=0 (a layer's
 * weight gradient on the caller's stream instead of a side stream beside its data gradient; the side stream and its two events
 * are created once per process).
 *
 * Conventions as in dvt_hip.h: int return codes (0 = ok, DVT_E_* / hipError_t otherwise), device pointers
 * owned by the caller, `stream` is a hipStream_t, nothing synchronises.
 */
#ifndef DVT_STAGE2_H
#define DVT_STAGE2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVT_S2_MAX_BLOCKS 8
#define DVT_S2_TENSORS_PER_BLOCK 12

typedef struct DvtS2Config {
  int32_t dim;        /* feat_dim: 768 (ViT-B) / 1024 / 384; multiple of 128 */
  int32_t heads;      /* dim / 64: head_dim is fixed to 64 (online_denoiser.py:27) */
  int32_t mlp_dim;    /* 4 * dim */
  int32_t tokens;     /* noise_map_height * noise_map_width = 1369 */
  int32_t tokens_pad; /* tokens rounded up to a multiple of 64 = 1408 */
  int32_t n_blocks;   /* num_blocks, 1 by default (main_denoiser.py:35) */
  int32_t enable_pe;  /* learnable pos_embed [tokens, dim] (online_denoiser.py:54-57) */
  float ln_eps;       /* 1e-6 */
} DvtS2Config;

/* Arena layout (floats).  out[0] = pos_embed (only meaningful when enable_pe), then for every block b the
 * 12 tensors out[1 + 12 b + i], i = norm1.weight, norm1.bias, attn.qkv.weight [3 dim, dim], attn.qkv.bias,
 * attn.proj.weight [dim, dim], attn.proj.bias, norm2.weight, norm2.bias, mlp.fc1.weight [mlp, dim],
 * mlp.fc1.bias, mlp.fc2.weight [dim, mlp], mlp.fc2.bias (nn.Linear layout, row-major [out, in]);
 * out[1 + 12 n_blocks] = total number of floats.  `out` must hold 2 + 12 n_blocks entries. */
int dvt_s2_param_offsets(const DvtS2Config* cfg, int64_t* out);

/* Bytes of scratch for `batch` images; training != 0 keeps every block's activations for the backward pass. */
int64_t dvt_s2_workspace_bytes(const DvtS2Config* cfg, int batch, int training);

/* pred = Denoiser(x)  (online_denoiser.py:86-91 with vit = None). */
int dvt_s2_forward(const DvtS2Config* cfg, const float* params, const float* x, float* pred, int batch,
                   void* work, int64_t work_bytes, void* stream);

/* Forward + loss + backward of one step (main_denoiser.py:212-220).  Gradients are ACCUMULATED into `grads`
 * (same layout as `params`; must be zero on entry unless accumulation is wanted -- dvt_adamw_step re-zeroes
 * it).  loss_out: device float[4] = {loss, l2_loss, cosine_similarity_loss, 0} (overwritten).  pred may be
 * NULL. */
int dvt_s2_train_step(const DvtS2Config* cfg, const float* params, float* grads, const float* x,
                      const float* target, float* pred, int batch, void* work, int64_t work_bytes,
                      float* loss_out, void* stream);

/* torch.optim.AdamW, one step over a flat arena of n floats (n % 4 == 0):
 *   g *= grad_scale (1 / world_size after a SUM all-reduce);  p *= 1 - lr * weight_decay;
 *   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
 *   p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps);  g = 0 (fused zero_grad).
 * step counts from 1. */
int dvt_adamw_step(float* params, float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
