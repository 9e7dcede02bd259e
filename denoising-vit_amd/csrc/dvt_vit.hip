// Frozen DINOv2 ViT forward for gfx950: bf16 operands on v_mfma_f32_16x16x32_bf16, fp32
// accumulation / residual stream / LayerNorm / softmax.  C ABI in include/dvt_vit.h.
//
// Reference: dvt/models/vit_wrapper.py:122-143 (get_intermediate_layers ->
// timm.VisionTransformer.forward_intermediates [timm 1.0.7, third party, not in the tree]),
// called from main_img_denoising.py:317-323 and :332-336.  Block structure restated in
// x += ls1 * proj(MHA(LN1(x))); x += ls2 * fc2(GELU(fc1(LN2(x)))).
//
// Data layout in HBM (one forward of `batch` images):
//   tokens of an image are padded from 1370 to s_pad = 1408 (multiple of 128) rows so that
//   no 128-row GEMM tile and no 64-key attention tile straddles two images; pad keys are
//   masked in the softmax, pad rows never reach the output.
//   x      fp32 [batch*s_pad, dim]        residual stream
//   xn     bf16 [batch*s_pad, dim]        LayerNorm output / attention output (GEMM A operand)
//   qk     bf16 [batch*s_pad, 2*dim]      q | k, head-major inside (timm qkv layout)
//   vt     bf16 [batch, heads, 64, s_pad] V TRANSPOSED per head: written by the qkv GEMM
//                                         epilogue so that P.V can read key-contiguous operands
//   hid    bf16 [batch*s_pad, mlp_dim]    GELU(fc1)
//   col    bf16 [batch*s_pad, k_patch]    im2col of the input image
//
// Kernels and what bounds them:
//   gemm_bf16   128x128x64 tiles, 4 waves x (64x64) = 16 MFMA accumulators per wave, operands
//               staged with global_load_lds (16 B/lane, XOR-swizzled via the SOURCE address),
//               2 LDS stages; MFMA-bound (2.5 PF/s bf16 dense).  Epilogues fused: +bias (q/k),
//               V transposed store, +bias+GELU, LayerScale+residual (fp32 in place), patch
//               embedding + pos_embed + cls.
//   attention   flash-style, S^T = K.Q^T so that the softmax statistics of a query live in
//               one lane column and P is already in B-operand layout for O^T = V^T.P^T.
//   layernorm   one wave per row, HBM-bound.
#include <math.h>

#include <type_traits>

#include "../../include/dvt_vit.h"
#include "dvt_common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

extern int g_f32x3_exact_attention, g_f32x3_unfused;  // dvt_vit_f32.hip (dvt_tune_set(1, -520 ... -523))

namespace {

typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// fp32 -> bf16, round-to-nearest-even, through the gfx950 conversion instruction
// (v_cvt_pk_bf16_f32: one instruction per pair)
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const hw_bf16x2 v = __builtin_convertvector((f32x2){a, b}, hw_bf16x2);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2(f, 0.f) & 0xffffu); }

// ======================================================================================
// GEMM  C[M,N] = A[M,K] . W[N,K]^T   (both operands K-contiguous bf16)
// ======================================================================================
constexpr int GBM = 128, GBN = 128, GBK = 64;
// GEMM schedule (dvt_tune_set(1, v)).  4 (default): 256x256 8-phase half-tile ring where M and N are whole
// 256-tiles, else 3; 3: 256x128 ping-pong where M is a whole 256-tile, else 1; 1: 128x128 two-stage.
// Every selectable schedule computes the same result (tests/test_gpu_vit.py runs them all).
// Builds with -DDVT_LAB (tools/build_lab.py -> libdvt_hip_lab.so, never loaded by dvt_amd) add the superseded /
// experimental schedules of lab/: 0 256x256 two-stage, 2 256x128 lock-step, 5 "8m", 10 "8h", 6..9 the 4-wave
// persistent kernel, plus the timing builds; the product library rejects those values (DVT_E_BADARG).
int g_vit_gemm_variant = 4;
constexpr int STAGE_BYTES = (GBM + GBN) * GBK * 2;  // 32 KB

// EPI_F32: y = acc + bias written as fp32 into `x` (the bf16x3 GEMMs of the fp32 extractor, dvt_vit_f32.hip)
// EPI_GELU_X3 / EPI_QKV_X3 (same mode): the outputs leave as (hi, lo) bf16 splits of the fp32 values -- GELU(h) as the
// next GEMM's [hi | hi | lo] row (`out`, row stride 3 N), q | k as two [M, 2 dim] arrays (`out`, `out_lo`) and V^T as two
// [batch, heads, 64, s_pad] arrays (`vt`, `vt_lo`): what the split kernels + the qkv prep kernels would write.
enum { EPI_BIAS = 0, EPI_QKV = 1, EPI_GELU = 2, EPI_RESID = 3, EPI_EMBED = 4, EPI_F32 = 5, EPI_GELU_X3 = 6, EPI_QKV_X3 = 7 };
#define IS_QKV(E) ((E) == EPI_QKV || (E) == EPI_QKV_X3)
#define IS_GELU(E) ((E) == EPI_GELU || (E) == EPI_GELU_X3)
#define IS_X3(E) ((E) == EPI_GELU_X3 || (E) == EPI_QKV_X3)

struct GemmBArgs {
  const bf16_t* A;
  const bf16_t* W;
  int M, N, K;
  const float* bias;  // [N]
  bf16_t* out;        // EPI_BIAS / EPI_GELU: [M, N]; EPI_QKV: qk [M, 2*dim]
  bf16_t* vt;         // EPI_QKV: [batch, heads, 64, s_pad]
  bf16_t* out_lo;     // EPI_QKV_X3: the lo parts of q | k ...
  bf16_t* vt_lo;      // ... and of V^T
  float* x;           // EPI_RESID / EPI_EMBED: residual stream [M, N]; EPI_F32: the fp32 output [M, N]
  const float* gamma; // EPI_RESID: LayerScale [N]
  const float* pos;   // EPI_EMBED: pos_embed [n_tokens, N]
  const float* cls;   // EPI_EMBED: prefix tokens [n_prefix, N] (cls, then register tokens)
  int dim, heads, s_pad, n_tokens;
  int n_prefix, pos_has_cls;  // EPI_EMBED: prefix rows; pos_embed row 0 belongs to cls (else patches only)
  int group;          // N tiles per L2-resident group (set by launch_gemm)
  int mblock;         // M panels per block of the tile order (1: n fastest)
  int nt_store;       // bf16 outputs with the non-temporal hint
  // LayerNorm folded into the GEMMs (see ln_fold): consumer side (EPI_QKV / EPI_GELU) ...
  const float2* ln_stats;  // [M] (mean, rstd) of the fp32 residual rows; A is then bf16(x), W is bf16(gamma (.) W)
  const float* ln_cs;      // [N] column sums of the folded weights
  // ... producer side (EPI_RESID): besides x, emit bf16(x) and the per-row partial sums of this 64-column block
  bf16_t* xb;              // [M, N]
  float2* st_part;         // [N / 64][M] (sum, sum of squares)
  float q_scale;      // EPI_QKV: the q columns (n < dim) leave multiplied by this (log2(e) / 8 for the log2-domain attention
                      // kernel); 0 = as they are
  int dim_ok_sq;      // 256-wide tiles may be used (no q|k|v boundary inside a tile)
  int lda, ldw;       // leading dimensions (elements) of A and W; 0 = K
  double work;        // profiling probe: ALGORITHMIC flops of this launch (0: 2*M*N*K of the padded shape)
  unsigned* dbg;      // timing builds of the 8p kernel (dvt_vit_debug_buffer): 24 u32 per wave group and workgroup
  int stagger_ticks;  // 8p: the first round of workgroups (one per CU) starts spread over this many 100-MHz ticks (0: together)
  int tpw;            // lab 8t: tiles per workgroup (generations of 256 workgroups walk tpw x 256 consecutive tiles)
  // map_tile_fast: the tile order's divisors as multiply-shift pairs (fd_make), set by launch_gemm for the 256 x 256 kernels
  unsigned fd_pg[2], fd_pb[2], fd_w[2], fd_hb[2];  // per_group = group * mt; per_block = mblock * group; width = group; hb = mblock
  int tiles_full;     // N tiles in whole groups (nt / group * group): ids beyond group `tiles_full / group` take the slow path
  int pf_next;        // 8p: late in its epilogue a workgroup touches the first pf_next (0..2) k-tiles' operand lines of tile
                      // blockIdx + 256, the tile the next workgroup of its XCD walks (NextTilePf; dvt_tune_set(1, -570 - n))
  int abl;            // developer library, TIMING ONLY (dvt_tune_set(1, -560 - mask)): bit 0 = the epilogues do not park their
                      // accumulators in LDS (outputs are garbage): what the parking writes cost a tile
};

// async global -> LDS copy of 16 B per lane; the LDS address is wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Stage one 128x64 bf16 tile (rows r0.., k0..) into `lds` (16 KB, rows of 128 B).  The LDS
// image is lane-linear; the 16-B chunk index inside a row is XOR-ed with (row & 7) by
// permuting the SOURCE address, and the fragment reads apply the same XOR.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ X, int ld, int r0, int k0,
                                           char* lds, int wave, int lane) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = it * 256 + wave * 64 + lane;  // LDS slot (16 B units)
    const int row = s >> 3, cp = s & 7;
    const int c = cp ^ (row & 7);
    glds16(X + (size_t)(r0 + row) * ld + k0 + c * 8, lds + (it * 256 + wave * 64) * 16);
  }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// ---- L2-aware tile rasterisation ------------------------------------------------------------
// Workgroups are dealt round-robin to the 8 XCDs (private 4 MiB L2 each), so (1) every XCD gets a
// CONTIGUOUS run of the tile order, and (2) the order walks N in groups of `group` tiles whose W
// slices (group * 128 * K * 2 B <= ~2.4 MB) stay L2-resident while all M panels stream past:
//     order = (n_group, m_panel, n_in_group)
// Plain "N fastest" re-streamed the whole W (4.7 MB for fc1/fc2 > L2) for every M panel:
// ~3.6 GB of memory-side reads per fc1 launch instead of ~0.6 GB.
struct TileMap {
  int m, n;
};
__device__ __forceinline__ TileMap map_tile(int bid, int nwg, int mt, int nt, int group, int mblock = 1) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;  // bijective
  const int per_group = group * mt;
  int g = id / per_group;
  const int full = nt / group;
  int width = group;
  if (g >= full) {  // last, narrower group
    g = full;
    width = nt - full * group;
  }
  const int rem = id - g * per_group;
  TileMap t;
  if (mblock <= 1) {
    t.m = rem / width;
    t.n = g * group + rem % width;
  } else {
    // (m block, n, m in block): the `mblock` tiles that share a W slab are ADJACENT in the order (they start
    // together and walk k in lock-step), and an A panel is re-touched every `mblock` ids
    const int per_block = mblock * width;
    const int mb = rem / per_block, r2 = rem - mb * per_block;
    const int left = mt - mb * mblock, hb = left < mblock ? left : mblock;
    t.n = g * group + r2 / hb;
    t.m = mb * mblock + r2 % hb;
  }
  return t;
}

// Unsigned division by a launch-invariant divisor d (1 <= d < 2^31) as multiply-high + shifts (Granlund-Montgomery, the
// round-up form): l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1;  x / d = (t + ((x - t) >> 1)) >> (l - 1), t = hi32(m x),
// exact for every x < 2^32 (l = 0, i.e. d = 1: m = 0, the formula degenerates -- handled).  The tile map of the 256 x 256 GEMM
// divides wave-UNIFORM values by run-time divisors four to six times per workgroup; the scalar unit has no divide, so hipcc
// expands each into ~30 VALU instructions with dependent v_rcp chains: 0.6 us from kernel entry to the first LDS-DMA issue,
// 2.4 % of a K = 768 tile (profiles/r06/r06j_*).  With these it is a handful of s_mul_hi_u32 (tests/test_kernel_math_cpu.py
// states the arithmetic; every GEMM test exercises it).
inline void fd_make(unsigned d, unsigned (&fd)[2]) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  fd[0] = d <= 1 ? 0u : (unsigned)((((1ull << l) - d) << 32) / d + 1);
  fd[1] = l;
}
__device__ __forceinline__ unsigned fd_div(unsigned x, const unsigned (&fd)[2]) {
  if (fd[1] == 0) return x;  // d = 1
  const unsigned t = __umulhi(x, fd[0]);
  return (t + ((x - t) >> 1)) >> (fd[1] - 1);
}
// map_tile with the regular divisors taken from the launch arguments; ids in the last, narrower N group and tiles of a short
// last M block (rare) fall back to the generic divisions.  Same result as map_tile for every id.
__device__ __forceinline__ TileMap map_tile_fast(const GemmBArgs& p, int bid, int nwg, int mt, int nt) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;  // bijective
  const int group = p.group, mblock = p.mblock, per_group = group * mt;
  if (id >= p.tiles_full * mt) return map_tile(bid, nwg, mt, nt, group, mblock);  // the narrower last group
  const int g = (int)fd_div((unsigned)id, p.fd_pg), rem = id - g * per_group;
  TileMap t;
  if (mblock <= 1) {
    t.m = (int)fd_div((unsigned)rem, p.fd_w);
    t.n = g * group + rem - t.m * group;
    return t;
  }
  const int per_block = mblock * group;
  const int mb = (int)fd_div((unsigned)rem, p.fd_pb), r2 = rem - mb * per_block;
  if (mt - mb * mblock < mblock) return map_tile(bid, nwg, mt, nt, group, mblock);  // a short last M block
  const int nq = (int)fd_div((unsigned)r2, p.fd_hb);
  t.n = g * group + nq;
  t.m = mb * mblock + r2 - nq * mblock;
  return t;
}

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU(), timm Mlp) = max(x, 0) - 0.5 |x| E(|x|), E = erfc(|x| / sqrt 2).
// Round 5: E by Abramowitz-Stegun 7.1.28, erfc(z) = (1 + a1 z + ... + a6 z^6)^-16 (|abs err| <= 3e-7; measured in fp32
// against fp64: 7.1e-7 on gelu over [-12, 12], i.e. fp32 rounding at |x| ~ 4, far below the bf16 rounding of the output), with
// the powers of 1 / sqrt 2 folded into the coefficients: 6 FMAs + 4 squarings + ONE v_rcp_f32.  Rounds 1-4 used 7.1.26
// (1.5e-7): one v_rcp_f32 AND one v_exp_f32 + 9 FMAs; transcendentals issue at a quarter of the FMA rate and the fc1
// epilogue is VALU-bound (96 of its ~120 cycles per element-wave were this function).  Overflow: the 16th power reaches inf
// beyond |x| ~ 27, v_rcp_f32(inf) = 0, the result is max(x, 0) exactly.  (libm's erff costs ~45 instructions.)
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  float p = fmaf(ax, 5.3829749049e-06f, 4.8890637117e-05f);
  p = fmaf(p, ax, 3.8003574446e-05f);
  p = fmaf(p, ax, 3.2776263542e-03f);
  p = fmaf(p, ax, 2.1141005680e-02f);
  p = fmaf(p, ax, 4.9867346883e-02f);
  p = fmaf(p, ax, 1.0f);
  p *= p;
  p *= p;
  p *= p;
  p *= p;
  const float e = __builtin_amdgcn_rcpf(p);
  return fmaf(-0.5f * ax, e, fmaxf(x, 0.f));
}

// Two elements at a time on the packed-fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: both halves are IEEE fma / mul, so every
// result is bit-identical to gelu_erf for non-denormal x).  The compiler packs the LayerNorm affine in front of it by itself but leaves the
// polynomial scalar; written as 2-vectors it is 17 VALU issues per PAIR (2 v_and for |x| -- VOP3P has no abs modifier --, 13
// packed, 2 v_rcp_f32) instead of 30 (15 per element: fmaxf also canonicalises its operand).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_erf_pair(float& x0, float& x1) {
  const f32x2_t ax = {fabsf(x0), fabsf(x1)};
  const auto k = [](float c) { return (f32x2_t){c, c}; };
  f32x2_t p = __builtin_elementwise_fma(ax, k(5.3829749049e-06f), k(4.8890637117e-05f));
  p = __builtin_elementwise_fma(p, ax, k(3.8003574446e-05f));
  p = __builtin_elementwise_fma(p, ax, k(3.2776263542e-03f));
  p = __builtin_elementwise_fma(p, ax, k(2.1141005680e-02f));
  p = __builtin_elementwise_fma(p, ax, k(4.9867346883e-02f));
  p = __builtin_elementwise_fma(p, ax, k(1.0f));
  p *= p;
  p *= p;
  p *= p;
  p *= p;
  const f32x2_t e = {__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
  // max(x, 0) = 0.5 x + 0.5 |x| EXACTLY for every x whose half is representable (all normal numbers >= 2^-125, +-0, +inf;
  // tests/test_kernel_math_cpu.py): one packed FMA with a neg modifier instead of two v_max_f32 per element (fmaxf
  // canonicalises its operand first).  A denormal x may lose its last bit (<= 2^-149), far below the bf16 store.
  const f32x2_t xv = {x0, x1};
  const f32x2_t mh = ax * k(-0.5f);
  const f32x2_t pos = __builtin_elementwise_fma(xv, k(0.5f), -mh);
  const f32x2_t r = __builtin_elementwise_fma(mh, e, pos);
  x0 = r.x;
  x1 = r.y;
}

// Fused epilogues.  acc[i][j][r] = C[m0 + wm*64 + i*16 + 4*(lane>>4) + r][n0 + wn*64 + j*16 + (lane&15)]
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmBArgs& p, f32x4 (&acc)[4][4], int m0, int n0,
                                              int wm, int wn, int lane) {
  const int g = lane >> 4, lc = lane & 15;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + lc;
    const float bias = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mrow = m0 + wm * 64 + i * 16 + 4 * g;  // first of this lane's 4 rows
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bias;
      if (EPI == EPI_QKV && p.q_scale != 0.f && n < p.dim) {  // q columns for the log2-domain attention kernel (tile-uniform)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= p.q_scale;
      }
      if (EPI == EPI_RESID) {
        const float gm = p.gamma[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* px = p.x + (size_t)(mrow + r) * p.N + n;
          *px = *px + gm * v[r];
        }
      } else if (EPI == EPI_EMBED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = mrow + r, s = t % p.s_pad;
          float o;
          if (s < p.n_prefix)
            o = p.cls[(size_t)s * p.N + n] + ((p.pos_has_cls && s == 0) ? p.pos[n] : 0.f);
          else if (s < p.n_tokens)
            o = v[r] + p.pos[(size_t)(s - p.n_prefix + p.pos_has_cls) * p.N + n];
          else
            o = 0.f;
          p.x[(size_t)t * p.N + n] = o;
        }
      } else if (EPI == EPI_F32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p.x[(size_t)(mrow + r) * p.N + n] = v[r];
      } else if (IS_QKV(EPI) && n0 >= 2 * p.dim) {
        // V: transposed store vt[b][h][d][s], the lane's 4 rows are 4 consecutive tokens
        const int f = n - 2 * p.dim, h = f >> 6, d = f & 63;
        const int b = mrow / p.s_pad, s = mrow - b * p.s_pad;
        uint2 pk;
        pk.x = pack2(v[0], v[1]);
        pk.y = pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p.vt + ((size_t)(b * p.heads + h) * 64 + d) * p.s_pad + s) = pk;
        if (EPI == EPI_QKV_X3) {
          uint2 pl;
          pl.x = pack2(v[0] - __uint_as_float(pk.x << 16), v[1] - __uint_as_float(pk.x & 0xffff0000u));
          pl.y = pack2(v[2] - __uint_as_float(pk.y << 16), v[3] - __uint_as_float(pk.y & 0xffff0000u));
          *reinterpret_cast<uint2*>(p.vt_lo + ((size_t)(b * p.heads + h) * 64 + d) * p.s_pad + s) = pl;
        }
      } else {
        if (IS_GELU(EPI)) {
          gelu_erf_pair(v[0], v[1]);
          gelu_erf_pair(v[2], v[3]);
        }
        const int ldo = (IS_QKV(EPI)) ? 2 * p.dim : p.N;
        // pair adjacent columns across lanes (l, l^1): even lanes store rows r=0,1 of the
        // column pair, odd lanes rows r=2,3 -> 4-B stores instead of 2-B stores
        const bool odd = lane & 1;
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
          const float mine0 = v[rp], mine1 = v[rp + 2];
          const float send = odd ? mine0 : mine1;
          const float recv = __shfl_xor(send, 1, 64);
          const int r = odd ? rp + 2 : rp;
          const uint32_t w = odd ? pack2(recv, mine1) : pack2(mine0, recv);
          const int ncol = odd ? n - 1 : n;
          *reinterpret_cast<uint32_t*>(p.out + (size_t)(mrow + r) * ldo + ncol) = w;
        }
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmBArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const TileMap tm = map_tile(blockIdx.x, gridDim.x, p.M / GBM, p.N / GBN, p.group);
  const int m0 = tm.m * GBM, n0 = tm.n * GBN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / GBK;
  stage_tile(p.A, p.K, m0, 0, smem, wave, lane);
  stage_tile(p.W, p.K, n0, 0, smem + GBM * GBK * 2, wave, lane);
  __syncthreads();  // waits vmcnt(0) for the LDS-DMA before releasing the workgroup
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * STAGE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
    if (kt + 1 < nk) {
      stage_tile(p.A, p.K, m0, (kt + 1) * GBK, nxt, wave, lane);
      stage_tile(p.W, p.K, n0, (kt + 1) * GBK, nxt + GBM * GBK * 2, wave, lane);
    }
    const char* As = cur;
    const char* Bs = cur + GBM * GBK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[4];
      const int chunk = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = read_frag(As, wm * 64 + i * 16 + (lane & 15), chunk);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = read_frag(Bs, wn * 64 + j * 16 + (lane & 15), chunk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();  // next stage landed (vmcnt(0)) and everyone is done reading `cur`
  }

  gemm_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane);
}

// ---- LDS-staged epilogue of the 256x128 kernel ---------------------------------------------
// The accumulator layout gives every lane 4 rows x 1 column per 16x16 tile, i.e. 4-byte
// global accesses in 64-B segments (measured: the K=768 proj GEMM spent 630 us where its MFMA
// work is 85 us).  Each wave therefore parks its 64x64 fp32 block (+bias) in its own LDS
// region (stride 68 floats) and reads it back row-wise: fp32 outputs move as float4 (a
// 256-B row segment per 16 lanes), bf16 outputs as 8 columns = 16 B per lane.
constexpr int EP_LD = 68;                      // floats per row of a wave's LDS block
constexpr int EP_WAVE_BYTES = 64 * EP_LD * 4;  // 17408 B x 8 waves = 136 KB <= 144 KB

template <int NI>
__device__ __forceinline__ void ln_fold(const GemmBArgs& p, f32x4 (&acc)[NI][4], int mrow0, int ncol0, int lane);

// L2 prefetch for the NEXT workgroup of this XCD (round 6).  A 256 x 256 tile starts cold: its first LDS-DMAs miss the L2 and a
// K = 768 tile waits ~2.3 us for them while every CU loads at once (profiles/r06/r06j_*: prologue 3.0 us of 25.2).  Workgroups go
// round-robin to the XCDs, so tile blockIdx + 256 is walked by a workgroup of THIS XCD (this L2) about when this one exits: late in
// its epilogue a workgroup touches the first one or two k-tiles' operand lines of that tile -- one dword per 128-B line and thread,
// 256 A rows + 256 W rows.  Placement: program order behind the last load whose result the epilogue still waits for (gfx9 retires
// vector-memory operations of a wave in issue order, and hipcc's counted s_waitcnt does not know these loads: older ones would be
// waited for).  The destination registers stay reserved to the kernel's end (pf_retire): the data lands long after the asm
// statement, and a "dead" destination would be handed out again -- as an address register, the first cut of this faulted.
// MEASURED NULL (profiles/r06/r07a_*, r06z_*): the prologue shrinks as predicted (qkv 3.1 -> 1.1 us, proj 2.5 -> 1.4, fc2 2.4 -> 1.6) and
// the first k-tiles run faster, but the launch does not get shorter (qkv 1793 -> 1812 us, fc1 2786 -> 2769, proj 946 -> 981):
// s_endpgm waits for a wave's outstanding loads, so the fetch latency moves from the next workgroup's prologue into this one's
// exit, and the K = 768 launches are bound by the THROUGHPUT of the shared L2 -> LDS / memory-side path, which a prefetch does
// not add to (the same reason the persistent and staggered variants gave nothing).  Off by default; kept as a knob.
struct NextTilePf {
  const bf16_t* src;  // this thread's operand row of the next tile (nullptr: nothing to touch)
  int n;              // k-tiles to touch (1..2)
  unsigned d0, d1;
};
__device__ __forceinline__ void pf_issue(NextTilePf& f) {
  if (f.src == nullptr) return;
  asm volatile("global_load_dword %0, %1, off" : "=v"(f.d0) : "v"(f.src) : "memory");
  if (f.n > 1) asm volatile("global_load_dword %0, %1, off offset:128" : "=v"(f.d1) : "v"(f.src) : "memory");
}
__device__ __forceinline__ void pf_retire(NextTilePf& f) {
  if (f.src == nullptr) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::"v"(f.d0), "v"(f.d1));
}

// NI: 16-row blocks per call (4: a wave's 64 x 64 block, 17 KB of LDS; 2: 32 x 64, 8.5 KB -- the persistent kernel's passes,
// which leave half of the ring to the next tile's operands); `blk0` overrides the wave's LDS block (default: smem + wave *
// EP_WAVE_BYTES)
template <int EPI, int NI = 4>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmBArgs& p, f32x4 (&acc)[NI][4], int m0,
                                                  int n0, int wm, int wn, int wave, int lane,
                                                  char* smem, char* blk0 = nullptr, NextTilePf* pf = nullptr) {
  static_assert(NI == 4 || NI == 2, "64- or 32-row passes");
  constexpr int ROWS = NI * 16;
  const int g = lane >> 4, lc = lane & 15;
  // LayerNorm folded into this GEMM (see ln_fold): the accumulators are x . W'^T of the UN-normalised rows
  const bool ln = (IS_QKV(EPI) || IS_GELU(EPI)) && p.ln_stats != nullptr;
  float* blk = reinterpret_cast<float*>(blk0 != nullptr ? blk0 : smem + wave * EP_WAVE_BYTES);
  const int nb = n0 + wn * 64;
  if (IS_QKV(EPI) && n0 >= 2 * p.dim) {
    // V tiles: vt[b][h][d][s] wants the TOKENS contiguous.  Round 5: the block goes through LDS transposed -- image
    // [64 features d][64 tokens] (pitch EP_LD), written as ONE ds_write_b128 per accumulator tuple (a lane's 4 rows are 4
    // consecutive tokens of one feature), read back as 8 tokens per lane and stored as 16 B per lane, 8 lanes = one whole
    // 128-B line of a feature row.  Before, every lane stored its tuples straight from the accumulators as 8-byte pieces (64
    // store instructions per block on 32-B segments) behind 32 scattered (mean, rstd) loads: a V tile's epilogue took ~15 us
    // against 3.6 us for a q / k tile, i.e. the qkv GEMM paid +4 us per tile on average (profiles/r05/README.md).
    const int mbv = m0 + wm * 64;
    // NI = 4: 8 chunks of 8 tokens x 8 features per pass, 8 passes; NI = 2: 4 chunks x 16 features, 4 passes
    constexpr int TCH = ROWS / 8, FPP = 64 / TCH, NPASS = 64 / FPP;
    const int tc = lane & (TCH - 1), dl = lane / TCH;  // 8-token chunk of the block, feature within a pass
    // folded LayerNorm: (mean, rstd) of this lane's 8 tokens -- 64 B per lane, coalesced, requested before the LDS traffic
    float4 stq[4];
    float csd[NPASS], bfd[NPASS];  // ... and the column sums / biases of its features (one per pass)
    if (ln) {
      const float4* sp = reinterpret_cast<const float4*>(p.ln_stats + mbv + tc * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) stq[q] = sp[q];
#pragma unroll
      for (int it = 0; it < NPASS; ++it) {
        csd[it] = p.ln_cs[nb + it * FPP + dl];
        bfd[it] = p.bias[nb + it * FPP + dl];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float bias = (p.bias != nullptr && !ln) ? p.bias[nb + j * 16 + lc] : 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const float4 t4 = make_float4(acc[i][j][0] + bias, acc[i][j][1] + bias, acc[i][j][2] + bias, acc[i][j][3] + bias);
        *reinterpret_cast<float4*>(blk + (j * 16 + lc) * EP_LD + i * 16 + 4 * g) = t4;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // each wave re-reads its own block only
    const int f0 = nb - 2 * p.dim, hh = f0 >> 6;        // (the block is one head's 64 features: dim_ok_sq / 64-aligned)
    // the image of THIS lane's 8 tokens: a block may straddle two images (s_pad % 64 != 0), an 8-token chunk never (s_pad % 8 == 0)
    const int trow = mbv + tc * 8, bimg = trow / p.s_pad, s0 = trow - bimg * p.s_pad;
    bf16_t* const vrow = p.vt + ((size_t)(bimg * p.heads + hh) * 64) * p.s_pad + s0;
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      if (pf != nullptr && it == 1) pf_issue(*pf);  // (behind the pass that consumed the last loaded parameters)
      const int d = it * FPP + dl;
      const float4 a = *reinterpret_cast<const float4*>(blk + d * EP_LD + tc * 8);
      const float4 b = *reinterpret_cast<const float4*>(blk + d * EP_LD + tc * 8 + 4);
      float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      if (ln) {  // rstd * (acc - mean * cs) + b'
        const float cs = csd[it], bf = bfd[it];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[2 * q] = fmaf(stq[q].y, v[2 * q] - stq[q].x * cs, bf);
          v[2 * q + 1] = fmaf(stq[q].w, v[2 * q + 1] - stq[q].z * cs, bf);
        }
      }
      typedef unsigned u32x4v_t __attribute__((ext_vector_type(4)));
      const u32x4v_t hi = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
      *reinterpret_cast<u32x4v_t*>(vrow + (size_t)d * p.s_pad) = hi;
      if constexpr (EPI == EPI_QKV_X3) {
        u32x4v_t lo;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          lo[q] = pack2(v[2 * q] - __uint_as_float(hi[q] << 16), v[2 * q + 1] - __uint_as_float(hi[q] & 0xffff0000u));
        *reinterpret_cast<u32x4v_t*>(p.vt_lo + (vrow - p.vt) + (size_t)d * p.s_pad) = lo;
      }
    }
    return;
  }
  // (mean, rstd) of the block's 64 rows: ONE coalesced load per lane, requested now and parked in the padding
  // columns of the LDS block with the accumulators (as 32 loads per lane at the top of the epilogue they cost a full
  // memory latency per tile)
  float2 st_row = make_float2(0.f, 0.f);
  if (ln && lane < ROWS) st_row = p.ln_stats[m0 + wm * 64 + lane];
  // ... and so are the folded LayerNorm's column sums / biases of the lane's 8 output columns (round 5; before, they were
  // requested after the LDS round trip, right in front of the loop that needs them.  Same-box A/B: no difference -- the other
  // waves of the CU cover that latency -- kept here because it is the natural place)
  const int c8_ln = (lane & 7) * 8;
  float4 cs0 = make_float4(0.f, 0.f, 0.f, 0.f), cs1 = cs0, bf0 = cs0, bf1 = cs0;
  if (ln && EPI != EPI_RESID && EPI != EPI_EMBED && EPI != EPI_F32) {
    cs0 = *reinterpret_cast<const float4*>(p.ln_cs + nb + c8_ln);
    cs1 = *reinterpret_cast<const float4*>(p.ln_cs + nb + c8_ln + 4);
    bf0 = *reinterpret_cast<const float4*>(p.bias + nb + c8_ln);
    bf1 = *reinterpret_cast<const float4*>(p.bias + nb + c8_ln + 4);
  }
#ifdef DVT_LAB
  if (!(p.abl & 1))
#endif
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float bias = (p.bias != nullptr && !ln) ? p.bias[nb + j * 16 + lc] : 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[(i * 16 + 4 * g + r) * EP_LD + j * 16 + lc] = acc[i][j][r] + bias;
  }
  if (ln && lane < ROWS) *reinterpret_cast<float2*>(blk + lane * EP_LD + 64) = st_row;
  // each wave only re-reads its own block: no workgroup barrier needed, only LDS completion
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int mb = m0 + wm * 64;
  if (EPI == EPI_RESID || EPI == EPI_EMBED || EPI == EPI_F32) {
    const int c4 = lc * 4;  // 16 lanes x float4 = one 64-float row; 4 rows per pass
    float4 gm = make_float4(1.f, 1.f, 1.f, 1.f);
    if (EPI == EPI_RESID) gm = *reinterpret_cast<const float4*>(p.gamma + nb + c4);
#pragma unroll 4
    for (int it = 0; it < ROWS / 4; ++it) {
      const int row = it * 4 + g;
      if (pf != nullptr && it == ROWS / 8 && EPI == EPI_F32) pf_issue(*pf);  // (EPI_F32 loads nothing per row; the others do)
      const float4 v = *reinterpret_cast<const float4*>(blk + row * EP_LD + c4);
      const int t = mb + row;
      float4* px = reinterpret_cast<float4*>(p.x + (size_t)t * p.N + nb + c4);
      if (EPI == EPI_RESID) {
        float4 o = *px;
        o.x += gm.x * v.x;
        o.y += gm.y * v.y;
        o.z += gm.z * v.z;
        o.w += gm.w * v.w;
        *px = o;
      } else if (EPI == EPI_F32) {
        *px = v;
      } else {
        const int sidx = t % p.s_pad;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sidx < p.n_prefix) {  // cls / register tokens (timm: cls gets pos_embed[0] unless no_embed_class)
          o = *reinterpret_cast<const float4*>(p.cls + (size_t)sidx * p.N + nb + c4);
          if (p.pos_has_cls && sidx == 0) {
            const float4 q = *reinterpret_cast<const float4*>(p.pos + nb + c4);
            o = make_float4(o.x + q.x, o.y + q.y, o.z + q.z, o.w + q.w);
          }
        } else if (sidx < p.n_tokens) {
          const float4 q = *reinterpret_cast<const float4*>(
              p.pos + (size_t)(sidx - p.n_prefix + p.pos_has_cls) * p.N + nb + c4);
          o = make_float4(v.x + q.x, v.y + q.y, v.z + q.z, v.w + q.w);
        }
        *px = o;
      }
    }
  } else {
    const int ldo = (IS_QKV(EPI)) ? 2 * p.dim : p.N;
    const int c8 = (lane & 7) * 8;  // 8 lanes x 8 columns = one row; 8 rows per pass
    const float qsc = (EPI == EPI_QKV && nb < p.dim) ? p.q_scale : 0.f;  // (wave-uniform: a 64-column block is q, k or v)
    // (GELU: one row at a time -- four rows of erf polynomials in flight took the fc1 kernel to 254 VGPRs,
    // and at 2 x 256 registers per SIMD no wave of the fit's streaming kernels can share the CU)
#pragma unroll(IS_GELU(EPI) ? 1 : 4)
    for (int it = 0; it < ROWS / 8; ++it) {
      if (pf != nullptr && it == 1) pf_issue(*pf);  // (pass 0 consumed the last loaded parameters)
      const int row = it * 8 + (lane >> 3);
      float4 a = *reinterpret_cast<const float4*>(blk + row * EP_LD + c8);
      float4 b = *reinterpret_cast<const float4*>(blk + row * EP_LD + c8 + 4);
      if (ln) {  // rstd * (acc - mean * cs) + b'
        const float2 stv = *reinterpret_cast<const float2*>(blk + row * EP_LD + 64);
        const float mu = stv.x, rs = stv.y;
        a.x = fmaf(rs, a.x - mu * cs0.x, bf0.x);
        a.y = fmaf(rs, a.y - mu * cs0.y, bf0.y);
        a.z = fmaf(rs, a.z - mu * cs0.z, bf0.z);
        a.w = fmaf(rs, a.w - mu * cs0.w, bf0.w);
        b.x = fmaf(rs, b.x - mu * cs1.x, bf1.x);
        b.y = fmaf(rs, b.y - mu * cs1.y, bf1.y);
        b.z = fmaf(rs, b.z - mu * cs1.z, bf1.z);
        b.w = fmaf(rs, b.w - mu * cs1.w, bf1.w);
      }
      if (IS_GELU(EPI)) {
        gelu_erf_pair(a.x, a.y);
        gelu_erf_pair(a.z, a.w);
        gelu_erf_pair(b.x, b.y);
        gelu_erf_pair(b.z, b.w);
      }
      if (EPI == EPI_QKV && qsc != 0.f) {  // q columns (tile-uniform): * log2(e) / 8 on the fp32 value, ONE rounding to bf16 as before
        a.x *= qsc; a.y *= qsc; a.z *= qsc; a.w *= qsc;
        b.x *= qsc; b.y *= qsc; b.z *= qsc; b.w *= qsc;
      }
      uint4 pk;
      pk.x = pack2(a.x, a.y);
      pk.y = pack2(a.z, a.w);
      pk.z = pack2(b.x, b.y);
      pk.w = pack2(b.z, b.w);
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t val = {pk.x, pk.y, pk.z, pk.w};
      if constexpr (IS_X3(EPI)) {
        u32x4_t vlo;
        vlo.x = pack2(a.x - __uint_as_float(pk.x << 16), a.y - __uint_as_float(pk.x & 0xffff0000u));
        vlo.y = pack2(a.z - __uint_as_float(pk.y << 16), a.w - __uint_as_float(pk.y & 0xffff0000u));
        vlo.z = pack2(b.x - __uint_as_float(pk.z << 16), b.y - __uint_as_float(pk.z & 0xffff0000u));
        vlo.w = pack2(b.z - __uint_as_float(pk.w << 16), b.w - __uint_as_float(pk.w & 0xffff0000u));
        if constexpr (EPI == EPI_GELU_X3) {  // the next GEMM's A row: [hi | hi | lo], each N wide
          bf16_t* r3 = p.out + (size_t)(mb + row) * (3 * p.N) + nb + c8;
          *reinterpret_cast<u32x4_t*>(r3) = val;
          *reinterpret_cast<u32x4_t*>(r3 + p.N) = val;
          *reinterpret_cast<u32x4_t*>(r3 + 2 * p.N) = vlo;
        } else {
          *reinterpret_cast<u32x4_t*>(p.out + (size_t)(mb + row) * ldo + nb + c8) = val;
          *reinterpret_cast<u32x4_t*>(p.out_lo + (size_t)(mb + row) * ldo + nb + c8) = vlo;
        }
        continue;
      }
      u32x4_t* dst = reinterpret_cast<u32x4_t*>(p.out + (size_t)(mb + row) * ldo + nb + c8);
      if (p.nt_store)
        __builtin_nontemporal_store(val, dst);  // streamed output: do not displace the operand panels in L2
      else
        *dst = val;
    }
  }
}

// ---- LayerNorm folded into the neighbouring GEMMs ------------------------------------------------------------
// xn = (x - mu) * rstd * gamma + beta, then xn . W^T + b, is   rstd * (x . W'^T - mu * cs) + b'   with
// W' = gamma (.) W, cs[n] = sum_k W'[n][k], b' = b + W beta.  So the consumer GEMM (qkv, fc1) reads bf16(x)
// itself and fixes its accumulators up with the row's (mu, rstd); the producer (the residual epilogue of proj /
// fc2) writes bf16(x) and per-row partial (sum, sum of squares) next to the fp32 x it writes anyway.  The
// LayerNorm kernel between them -- 553 MB read + 277 MB written per call, 168 calls per image -- disappears.
template <int NI>
__device__ __forceinline__ void ln_fold(const GemmBArgs& p, f32x4 (&acc)[NI][4], int mrow0, int ncol0, int lane) {
  const int g = lane >> 4, lc = lane & 15;
  float cs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = p.ln_cs[ncol0 + j * 16 + lc];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 st = p.ln_stats[mrow0 + i * 16 + 4 * g + r];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j][r] = st.y * (acc[i][j][r] - st.x * cs[j]);
    }
}

// (mu, rstd) of every row from the P column-block partials written by the residual epilogue (fixed order: the
// result does not depend on scheduling)
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float2* __restrict__ part, int P, int M, float inv_n,
                                                                float eps, float2* __restrict__ stats) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float s1 = 0.f, s2 = 0.f;
  for (int q = 0; q < P; ++q) {
    const float2 v = part[(size_t)q * M + m];
    s1 += v.x;
    s2 += v.y;
  }
  const float mu = s1 * inv_n;
  const float var = fmaxf(s2 * inv_n - mu * mu, 0.f);
  stats[m] = make_float2(mu, rsqrtf(var + eps));
}

// After the patch embedding (whose epilogue is not a residual epilogue): bf16(x) and exact two-pass (mu, rstd)
__global__ __launch_bounds__(256) void ln_cast_stats_kernel(const float* __restrict__ x, bf16_t* __restrict__ xb,
                                                            float2* __restrict__ stats, int rows, int dim, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * dim);
  const int nq = dim >> 2;
  float4 v[4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      v[i] = xr[q];
      sum += v[i].x + v[i].y + v[i].z + v[i].w;
      uint2 pk;
      pk.x = pack2(v[i].x, v[i].y);
      pk.y = pack2(v[i].z, v[i].w);
      reinterpret_cast<uint2*>(xb + (size_t)row * dim)[q] = pk;
    }
  }
  const float mean = wave_sum(sum) / (float)dim;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      var += a * a + b * b + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(var) / (float)dim + eps);
  if (lane == 0) stats[row] = make_float2(mean, rstd);
}

// sum over the 16 lanes of a DPP row (rotations inside the row: v_add_f32 with a row_ror modifier, no LDS traffic)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// EPI_RESID epilogue of the 256x256 kernels: x[m, n] += gamma[n] * (acc + bias[n]) on a wave's
// 128 x 64 block.  The fp32 residual stream is a read-modify-write of 8 B per element (1.1 GB per
// launch); with the loads issued four at a time inside the row loop a tile's epilogue took 4 x 2
// rounds of memory latency -- longer than the whole K = 768 k-loop.  Here all 16 float4 rows of a
// 64-row half are requested up front (64 VGPRs; the k-loop's fragment registers are dead), and the
// second half's requests go out as soon as the first half's accumulators are parked in LDS.
__device__ __forceinline__ void gemm_epilogue_resid_sq(const GemmBArgs& p, f32x4 (&acc)[8][4], int mb,
                                                       int nb, int wave, int lane, char* smem, NextTilePf* pf = nullptr) {
  const int g = lane >> 4, lc = lane & 15, c4 = lc * 4;
  float* blk = reinterpret_cast<float*>(smem + wave * EP_WAVE_BYTES);
  // buffer addressing: one resource (SGPRs) based at the block, ONE 32-bit lane offset and a
  // scalar row offset per access -- 32 rows of 64-bit global addresses would cost 64 VGPRs
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      p.x + (size_t)mb * p.N + nb, 0, 0x7fffffff, 0x00020000);
  const int loff = (g * p.N + c4) * 4;
  const int rstep = p.N * 16;  // bytes per 4 rows
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
#define RS_LD(it) __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, loff, (it) * rstep, 0))
// Stores put the row offset into the VGPR offset (soffset = literal 0): with an SGPR soffset the
// compiler assumes there is no ">64-bit store data overwritten by the next VALU" hazard and emits
// no wait state, but gfx950 showed exactly that corruption (lanes 12-15 of each 16, one dword).
#define RS_ST(it, v) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), xr, loff + (it) * rstep, 0, 0)
// LayerNorm producer side: bf16 copy of the new row piece (16 lanes x 8 B = 128 B per row) and the row's partial
// (sum, sum of squares) over this wave's 64 columns: xor-reduce over the 16 lanes that hold the row
#define RS_LN(it, o)                                                                                    \
  do {                                                                                                  \
    const int row_ = mb + (it) * 4 + g;                                                                 \
    uint2 pk_;                                                                                          \
    pk_.x = pack2((o).x, (o).y);                                                                        \
    pk_.y = pack2((o).z, (o).w);                                                                        \
    *reinterpret_cast<uint2*>(p.xb + (size_t)row_ * p.N + nb + c4) = pk_;                               \
    float s1_ = ((o).x + (o).y) + ((o).z + (o).w);                                                      \
    float s2_ = ((o).x * (o).x + (o).y * (o).y) + ((o).z * (o).z + (o).w * (o).w);                      \
    s1_ = row16_sum(s1_);                                                                               \
    s2_ = row16_sum(s2_);                                                                               \
    if (lc == 0) p.st_part[(size_t)(nb >> 6) * p.M + row_] = make_float2(s1_, s2_);                     \
  } while (0)
  float4 xl[16], xh[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) xl[it] = RS_LD(it);
  float bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] = p.bias != nullptr ? p.bias[nb + j * 16 + lc] : 0.f;
  const float4 gm = *reinterpret_cast<const float4*>(p.gamma + nb + c4);
#ifdef DVT_LAB
  if (!(p.abl & 1))
#endif
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[(i * 16 + 4 * g + r) * EP_LD + j * 16 + lc] = acc[i][j][r] + bias[j];
#pragma unroll
  for (int it = 0; it < 16; ++it) xh[it] = RS_LD(16 + it);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const float4 v = *reinterpret_cast<const float4*>(blk + (it * 4 + g) * EP_LD + c4);
    float4 o = xl[it];
    o.x += gm.x * v.x;
    o.y += gm.y * v.y;
    o.z += gm.z * v.z;
    o.w += gm.w * v.w;
    RS_ST(it, o);
    if (p.xb != nullptr) RS_LN(it, o);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own reads of the block are complete
  if (pf != nullptr) pf_issue(*pf);  // younger than every row load, older than the second half's stores only
#ifdef DVT_LAB
  if (!(p.abl & 1))
#endif
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[(i * 16 + 4 * g + r) * EP_LD + j * 16 + lc] = acc[4 + i][j][r] + bias[j];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const float4 v = *reinterpret_cast<const float4*>(blk + (it * 4 + g) * EP_LD + c4);
    float4 o = xh[it];
    o.x += gm.x * v.x;
    o.y += gm.y * v.y;
    o.z += gm.z * v.z;
    o.w += gm.w * v.w;
    RS_ST(16 + it, o);
    if (p.xb != nullptr) RS_LN(16 + it, o);
  }
#undef RS_LD
#undef RS_ST
#undef RS_LN
}

// ---- 256x128x64 tile, 8 waves (4 x 2, 64x64 each), 3 LDS stages, counted vmcnt: the kernel of shapes
// with N % 256 != 0 (ViT-S: dim 384) -------------------------------------------------------------
// The loads of stage kt+2 are issued while stage kt is being multiplied: `s_waitcnt vmcnt(6)` retires
// exactly the oldest stage (6 LDS-DMA instructions per thread per stage: 4 for the 256-row A tile, 2
// for the 128-row W tile); never vmcnt(0) inside the loop.
constexpr int G2_BM = 256, G2_STAGE = (G2_BM + GBN) * GBK * 2;  // 48 KB

// Ping-pong schedule (the lock-step version of this tile -- one barrier per k-tile, lab/dvt_vit_lab.inc --
// ran the MFMA pipe 26-33 % busy: the two waves of a SIMD want the matrix pipe in the same window and
// both leave it idle while they stage / read the next tile).  Here every k-tile
// has a LOAD segment (issue the DMA of tile kt+2, read all fragments of tile kt into registers)
// and a COMPUTE segment (32 MFMAs), each closed by a barrier, and waves 4-7 ("group B") execute
// ONE extra barrier up front: for the whole loop group B is one segment behind group A, i.e. one
// group's MFMAs run beside the other group's DMA issue + ds_reads (the 8-phase idea of the CDNA
// guide, reduced to two phases).  Correctness of the hand-off: a wave waits for its own stage-
// (kt+1) DMA (`vmcnt(6)`) and its fragment reads (`lgkmcnt(0)`) at the END of LOAD(kt), so by the
// time any wave starts LOAD(kt+1) both groups have passed a barrier behind those waits; the DMA
// into buffer (kt+2) % 3 is issued one barrier after the last read of tile kt-1 by either group.
template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_kernel_pp(GemmBArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[3 * G2_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const bool group_b = wave >= 4;
  const TileMap tm = map_tile(blockIdx.x, gridDim.x, p.M / G2_BM, p.N / GBN, p.group);
  const int m0 = tm.m * G2_BM, n0 = tm.n * GBN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / GBK;
  const bf16_t* srcA[4];
  const bf16_t* srcW[2];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s_ = it * 512 + tid, row = s_ >> 3, c = (s_ & 7) ^ (row & 7);
    srcA[it] = p.A + (size_t)(m0 + row) * p.K + c * 8;
    if (it < 2) srcW[it] = p.W + (size_t)(n0 + row) * p.K + c * 8;
  }
  const int ldsw = wave * 1024;
#define PP_ISSUE(kt)                                                                  \
  do {                                                                                \
    char* st_ = smem + ((kt) % 3) * G2_STAGE + ldsw;                                  \
    const int ko_ = (kt) * GBK;                                                       \
    _Pragma("unroll") for (int it = 0; it < 4; ++it)                                  \
        glds16(srcA[it] + ko_, st_ + it * 8192);                                      \
    _Pragma("unroll") for (int it = 0; it < 2; ++it)                                  \
        glds16(srcW[it] + ko_, st_ + G2_BM * GBK * 2 + it * 8192);                    \
  } while (0)
  PP_ISSUE(0);
  if (nk > 1) {
    PP_ISSUE(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();           // tile 0 is in LDS for everybody
  if (group_b) __builtin_amdgcn_s_barrier();  // group B: one segment behind from here on
  const int rowa = wm * 64 + (lane & 15), rowb = wn * 64 + (lane & 15), cg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    // ---------------- LOAD segment ----------------
    if (kt + 2 < nk) PP_ISSUE(kt + 2);
    const char* As = smem + (kt % 3) * G2_STAGE;
    const char* Bs = As + G2_BM * GBK * 2;
    bf16x8 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = read_frag(As, rowa + i * 16, cg);
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = read_frag(Bs, rowb + j * 16, cg);
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = read_frag(As, rowa + i * 16, 4 + cg);
#pragma unroll
    for (int j = 0; j < 4; ++j) b1[j] = read_frag(Bs, rowb + j * 16, 4 + cg);
    if (kt + 2 < nk)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // own DMA of tile kt+1 has landed
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- COMPUTE segment ----------------
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
#undef PP_ISSUE
  if (!group_b) __builtin_amdgcn_s_barrier();  // group A: balance group B's extra barrier
  __syncthreads();  // every wave is done with the operand stages: the buffers become epilogue space
  gemm_epilogue_lds<EPI>(p, acc, m0, n0, wm, wn, wave, lane, smem);
}

// ---- 256x256x64 tile, 8-phase schedule ("8p") ------------------------------------------------
// A two-stage ring of whole 64-KB k-tiles (lab/dvt_vit_lab.inc, `sq`) drains its LDS-DMA queue at a
// __syncthreads per k-tile, so every k-tile costs one loaded L2 latency (~1.7 us against 0.85 us of
// MFMA work).  Here the 64-KB k-tile is
// split into four 16-KB HALF-TILES (A0, A1, B0, B1: 128 operand rows x 64 k each) that are
// consumed and re-staged one per phase, so four half-tiles (64 KB per CU) are in flight at ALL
// times and nothing ever waits for vmcnt(0).  (A ten-slot ring over the whole 160-KB LDS with six
// half-tiles in flight measured the same rate -- the k-loop is not bytes-in-flight bound -- and
// would evict the co-resident fit kernels, so the ring stays at eight slots.)
//   half h of A holds rows m0 + (r>>6)*128 + h*64 + (r&63), half h of W rows n0 + (r>>5)*64 +
//   h*32 + (r&31) (r = local row): wave (wm, wn) still owns the contiguous 128 x 64 output block.
//   phase   ds_read (-> regs)   MFMA quadrant     LDS-DMA issued        s_waitcnt (end of L)
//   P1(t)   B0(t), A0(t)        (A0, B0)          B1(t+1)               vmcnt(8)  [B1(t) landed]
//   P2(t)   B1(t)               (A0, B1)          A1(t+1)               vmcnt(8)  [A1(t)]
//   P3(t)   A1(t)               (A1, B1)          A0(t+2)               --
//   P4(t)   --                  (A1, B0)          B0(t+2)               vmcnt(8)  [A0, B0 (t+1)]
// Every phase is {L: ds_reads + DMA issue + counted wait} barrier {M: 16 MFMAs} barrier; the
// waves with wm == 1 run half a phase behind (one extra barrier up front), so on each SIMD one
// wave is in its MFMA segment while the other one issues loads.  Hazards:
//   WAR  a half-tile is re-staged two phases after the phase that read it: all reads of phase p
//        (both groups) have retired at the second barrier of phase p.
//   RAW  a half-tile is read one phase after the phase whose L segment waited for it: the waits
//        of both groups precede the second barrier of that phase.
template <int MODE>  // 0 steady state, 1 tile nk-2 (no issue in P3/P4), 2 tile nk-1 (no issue)
struct P8Wait;
template <> struct P8Wait<0> { static constexpr int w1 = 8, w2 = 8, w4 = 8; };
template <> struct P8Wait<1> { static constexpr int w1 = 8, w2 = 8, w4 = 4; };
template <> struct P8Wait<2> { static constexpr int w1 = 2, w2 = 0, w4 = -1; };

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#define P8_BAR()                         \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define P8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// 16 MFMAs: 4 m-frags x 2 n-frags x 2 k-steps; dependent pairs are 8 issues apart
#define P8_MFMA(IB, JB, AF, BF)                                                                   \
  do {                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
        acc[IB + i][JB + j] =                                                                     \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[i][ks], BF[j][ks], acc[IB + i][JB + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                \
  } while (0)

// ---- round 6: the walk of the ring ("8b").  The table above is round 5's walk (12 / 4 / 8 / 0 fragment reads per phase, ring
// parity a run-time value); it lives on in lab/dvt_vit_gemm8p_lab.inc as the developer library's schedule 13, bit-identical to
// this one (tests/test_gpu_lab.py).  Round 4's per-barrier stamps (profiles/r04/r04w_*) showed seven of a k-tile's eight
// half-phases at 325-345 cycles -- the partner group's 16 MFMAs + two barriers -- and the eighth, the load segment of P1, at
// 570-590: it carried 12 of the k-tile's 24 ds_read_b128 (B0 and A0) behind the k-tile boundary's address arithmetic (VALU:
// the parity offset of 12 read addresses and four 64-bit DMA source pointers), while P4's load segment read nothing.  Now:
//   * the k-loop is unrolled by two, so the parity of every slot is a literal: fragment-read addresses are loop-invariant
//     registers + instruction offsets, the DMA goes through two buffer descriptors (SGPRs) with one 32-bit lane offset per pass
//     and the half / k-tile in the scalar offset -- no VALU instruction in any load segment;
//   * B0 of k-tile t+1 is read in the load segment of P4(t), into the B registers that died with P3's MFMAs: the phases read
//     8 / 4 / 8 / 4 fragments (A0 | B1 | A1 | B0'), and the two B register sets swap roles every k-tile (hence the unroll);
//   * B0 is therefore needed one phase earlier: the stages are issued in the order B1(t+1) A1(t+1) B0(t+2) A0(t+2), one per
//     phase, and EVERY phase waits vmcnt(8) (four stages in flight, as before).
//   phase   ds_read (-> regs)    MFMA quadrant    LDS-DMA issued   s_waitcnt vmcnt(8) retires
//   P1(t)   A0(t)                (A0, B0)         B1(t+1)          B1(t)
//   P2(t)   B1(t)    -> bx       (A0, B1)         A1(t+1)          A1(t)
//   P3(t)   A1(t)                (A1, B1)         B0(t+2)          B0(t+1)
//   P4(t)   B0(t+1)  -> bx       (A1, B0)         A0(t+2)          A0(t+1)
// Hazards: a slot is re-staged three phases after the phase that read it (round 5: two); a half-tile is read one phase after
// the phase whose load segment waited for it; the waits of both groups precede the second barrier of that phase.
// Measured (profiles/r06/r06b_*, 398 views, random operands, interleaved with round 5's walk): qkv -0.6 %, proj -0.7 %,
// fc1 -1.7 %, fc2 (48 k-tiles: the k-loop itself) -3.4 %.  Issuing the stage inside the MFMA segment instead ("8m"
// placement on this walk) LOSES 1-7 %.  K / 64 must be even (launch_gemm).
// MODE of a k-tile: 0 steady state; 1 k-tile nk-2 (nothing to issue in P3 / P4); 2 k-tile nk-1 (nothing to issue).  The
// persistent kernel (gemm_bf16_kernel_8t) adds 3 = k-tile 0 of a tile whose k-tile 0 was staged under the previous tile
// (P1 / P2 need no wait) and 4 = k-tile nk-1 staging k-tile 0 of the NEXT tile, one half-tile per phase (rsAn / rsBn).
template <int MODE>
struct P8BWait;
template <> struct P8BWait<0> { static constexpr int w1 = 8, w2 = 8, w3 = 8, w4 = 8; };
template <> struct P8BWait<1> { static constexpr int w1 = 8, w2 = 8, w3 = 6, w4 = 4; };
template <> struct P8BWait<2> { static constexpr int w1 = 2, w2 = 0, w3 = -1, w4 = -1; };
template <> struct P8BWait<3> { static constexpr int w1 = -1, w2 = -1, w3 = 8, w4 = 8; };
template <> struct P8BWait<4> { static constexpr int w1 = 4, w2 = 4, w3 = -1, w4 = -1; };

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Per-lane constants of the 8p / 8t k-loop (declares locals in the caller's scope): the DMA goes through two buffer
// descriptors (SGPRs) based at the tile's A / W rows -- per lane ONE 32-bit byte offset per pass (512 threads x 16 B = 64
// rows of a half-tile), the half (h) and the k-tile in the scalar offset: no 64-bit VALU address arithmetic in the load
// segments, 4 address registers instead of 16.  Fragment read offsets inside a half-tile: row*128 + ((ks*4 + cg) ^ (row & 7))*16;
// the same four offsets into the ring's second parity are registers of their own (opaque to the compiler): ds_read's
// instruction offset is 16 bits, the ring 128 KB -- left alone hipcc forms one address register per READ of parity 1 (12).
#define P8_LANE_SETUP()                                                                                                  \
  int voA[2], voB[2];                                                                                                    \
  _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                                                     \
    const int s_ = it * 512 + tid, r = s_ >> 3, c = (s_ & 7) ^ (r & 7);                                                  \
    voA[it] = (((r >> 6) * 128 + (r & 63)) * p.lda + c * 8) * 2;                                                         \
    voB[it] = (((r >> 5) * 64 + (r & 31)) * p.ldw + c * 8) * 2;                                                          \
  }                                                                                                                      \
  const int hA = 64 * p.lda * 2, hB = 32 * p.ldw * 2; /* second half-tile: +64 A rows / +32 W rows */                    \
  constexpr int OFF_A0 = 0, OFF_A1 = 16384, OFF_B0 = 32768, OFF_B1 = 49152, BUF = 65536;                                 \
  char* const ldsw = smem + wave * 1024;                                                                                 \
  const int cg = lane >> 4;                                                                                              \
  const int ra = wm * 64 + (lane & 15), rb = wn * 32 + (lane & 15);                                                      \
  const int oa0 = ra * 128 + (((0 + cg) ^ (ra & 7)) << 4), oa1 = ra * 128 + (((4 + cg) ^ (ra & 7)) << 4);                \
  const int ob0 = rb * 128 + (((0 + cg) ^ (rb & 7)) << 4), ob1 = rb * 128 + (((4 + cg) ^ (rb & 7)) << 4);                \
  int oa0q = oa0 + 65536, oa1q = oa1 + 65536, ob0q = ob0 + 65536, ob1q = ob1 + 65536;                                    \
  asm volatile("" : "+v"(oa0q), "+v"(oa1q), "+v"(ob0q), "+v"(ob1q))
#define P8_RSRC(ptr, len) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (len), 0x00020000)
// stage half H (0 / 1) of operand X (A / B) of k-tile kt through descriptor RS into the slot at byte offset off
// (P8_AUX_A / P8_AUX_B: cache-policy bits of the two operand streams -- 0 in the product; the developer library's schedules
// 14 / 15 / 16 set nt (2) on A / W / both: profiles/r06/r06m_*)
#define P8_STAGE_RS(RS, X, H, kt, off)                                                                                   \
  do {                                                                                                                   \
    const int so_ = (kt) * (GBK * 2) + (H) * h##X;                                                                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lds_ptr_t)(ldsw + (off)), 16, vo##X[0], so_, 0, P8_AUX_##X);           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lds_ptr_t)(ldsw + (off) + 8192), 16, vo##X[1], so_, 0, P8_AUX_##X);    \
  } while (0)
#define P8_STAGE(X, H, kt, off) P8_STAGE_RS(rs##X, X, H, kt, off)
#define P8_RD(base, off) (*reinterpret_cast<const bf16x8*>((base) + (off)))
// k-tile t of parity PAR (a literal); BC: the B registers that hold B0(t), BX: the other set
#define P8B_TILE(MODE, PAR, t, BC, BX)                                                            \
  do {                                                                                            \
    constexpr int bo_ = (PAR) * BUF, bn_ = bo_ ^ BUF;                                             \
    constexpr bool st12_ = MODE == 0 || MODE == 1 || MODE == 3, st34_ = MODE == 0 || MODE == 3;   \
    const int ra0_ = (PAR) ? oa0q : oa0, ra1_ = (PAR) ? oa1q : oa1;  /* this parity's A reads */   \
    const int rb0_ = (PAR) ? ob0q : ob0, rb1_ = (PAR) ? ob1q : ob1;  /* ... B reads */             \
    const int rn0_ = (PAR) ? ob0 : ob0q, rn1_ = (PAR) ? ob1 : ob1q;  /* the other parity's B0 */   \
    /* P1 */                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      a[i][0] = P8_RD(smem + OFF_A0 + i * 2048, ra0_);                                            \
      a[i][1] = P8_RD(smem + OFF_A0 + i * 2048, ra1_);                                            \
    }                                                                                             \
    if (st12_) P8_STAGE(B, 1, (t) + 1, bn_ + OFF_B1);                                             \
    if (MODE == 4) P8_STAGE_RS(rsBn, B, 0, 0, OFF_B0);                                            \
    wait_vm<P8BWait<MODE>::w1>();                                                                 \
    P8_BAR();                                                                                     \
    P8_LGKM0();                                                                                   \
    P8_MFMA(0, 0, a, BC);                                                                         \
    P8_BAR();                                                                                     \
    /* P2 */                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
      BX[j][0] = P8_RD(smem + OFF_B1 + j * 2048, rb0_);                                           \
      BX[j][1] = P8_RD(smem + OFF_B1 + j * 2048, rb1_);                                           \
    }                                                                                             \
    if (st12_) P8_STAGE(A, 1, (t) + 1, bn_ + OFF_A1);                                             \
    if (MODE == 4) P8_STAGE_RS(rsAn, A, 0, 0, OFF_A0);                                            \
    wait_vm<P8BWait<MODE>::w2>();                                                                 \
    P8_BAR();                                                                                     \
    P8_LGKM0();                                                                                   \
    P8_MFMA(0, 2, a, BX);                                                                         \
    P8_BAR();                                                                                     \
    /* P3 */                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      a[i][0] = P8_RD(smem + OFF_A1 + i * 2048, ra0_);                                            \
      a[i][1] = P8_RD(smem + OFF_A1 + i * 2048, ra1_);                                            \
    }                                                                                             \
    if (st34_) P8_STAGE(B, 0, (t) + 2, bo_ + OFF_B0);                                             \
    if (MODE == 4) P8_STAGE_RS(rsBn, B, 1, 0, OFF_B1);                                            \
    wait_vm<P8BWait<MODE>::w3>();                                                                 \
    P8_BAR();                                                                                     \
    P8_LGKM0();                                                                                   \
    P8_MFMA(4, 2, a, BX);                                                                         \
    P8_BAR();                                                                                     \
    /* P4: B0 of the NEXT k-tile -> BX (dead: P3's MFMAs have been issued) */                      \
    if (MODE != 2 && MODE != 4) {                                                                 \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
        BX[j][0] = P8_RD(smem + OFF_B0 + j * 2048, rn0_);                                         \
        BX[j][1] = P8_RD(smem + OFF_B0 + j * 2048, rn1_);                                         \
      }                                                                                           \
    }                                                                                             \
    if (st34_) P8_STAGE(A, 0, (t) + 2, bo_ + OFF_A0);                                             \
    if (MODE == 4) P8_STAGE_RS(rsAn, A, 1, 0, OFF_A1);                                            \
    wait_vm<P8BWait<MODE>::w4>();                                                                 \
    P8_BAR();                                                                                     \
    P8_MFMA(4, 0, a, BC);                                                                         \
    P8_BAR();                                                                                     \
  } while (0)

// STAMP (developer library only, schedule 12): the tile's anatomy in WALL time -- s_memrealtime (100 MHz, not the shader clock:
// independent of DVFS) at kernel entry, at the first MFMA (prologue done), behind the k-loop, behind the last store's issue and
// behind its retirement, + HW_ID / XCC_ID (which CU): 8 u32 per workgroup at p.dbg (tools/lab_gemm8p_anatomy.py).
template <int EPI, bool STAMP = false, int AUX = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_kernel_8p(GemmBArgs p) {
  constexpr int P8_AUX_A = (AUX & 1) ? 2 : 0, P8_AUX_B = (AUX & 2) ? 2 : 0;  // nt on the A / W stream (lab schedules 14..16)
  __shared__ __attribute__((aligned(16))) char smem[8 * EP_WAVE_BYTES];  // 136 KB >= 8 half-tiles (128 KB)
  unsigned long long ts_[5] = {0, 0, 0, 0, 0};
  if constexpr (STAMP) ts_[0] = __builtin_amdgcn_s_memrealtime();
  // De-synchronised start (dvt_tune_set(1, -700 - pct); round 5: null except proj, default off): the workgroups of the first
  // round wait a hash-spread fraction of one tile time (s_memrealtime, 100 MHz).  Results do not depend on it.
  if (p.stagger_ticks > 0 && blockIdx.x < 256) {
    const unsigned h_ = ((unsigned)blockIdx.x * 2654435761u) >> 16;  // 16-bit hash of the block id
    const unsigned long long wait_ = ((unsigned long long)p.stagger_ticks * h_) >> 16;
    const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0_ < wait_) __builtin_amdgcn_s_sleep(16);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const TileMap tm = map_tile_fast(p, blockIdx.x, gridDim.x, p.M >> 8, p.N >> 8);
  const int m0 = tm.m * 256, n0 = tm.n * 256;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / GBK;  // even, >= 2 (launch_gemm checks)
  P8_LANE_SETUP();
  const __amdgpu_buffer_rsrc_t rsA = P8_RSRC(p.A + (size_t)m0 * p.lda, 0x7fffffff);
  const __amdgpu_buffer_rsrc_t rsB = P8_RSRC(p.W + (size_t)n0 * p.ldw, 0x7fffffff);
  const __amdgpu_buffer_rsrc_t rsAn = rsA, rsBn = rsB;  // (MODE 4 is the persistent kernel's: not instantiated here)

  // prologue: the whole k-tile 0 in the order it is needed, then what P3 / P4 of "k-tile -1" would have issued
  unsigned ts_setup_ = 0, ts_issued_ = 0;
  if constexpr (STAMP) {
    asm volatile("" : "+v"(voA[0]), "+v"(voB[0]), "+v"(oa0q));  // (the setup arithmetic is complete here, not sunk below)
    ts_setup_ = (unsigned)__builtin_amdgcn_s_memrealtime();
  }
  P8_STAGE(B, 0, 0, OFF_B0);
  P8_STAGE(A, 0, 0, OFF_A0);
  P8_STAGE(B, 1, 0, OFF_B1);
  P8_STAGE(A, 1, 0, OFF_A1);
  P8_STAGE(B, 0, 1, BUF + OFF_B0);
  P8_STAGE(A, 0, 1, BUF + OFF_A0);
  if constexpr (STAMP) ts_issued_ = (unsigned)__builtin_amdgcn_s_memrealtime();
  wait_vm<8>();  // B0(0), A0(0) have landed
  P8_BAR();
  if (wm == 1) P8_BAR();  // group 1: half a phase behind from here on

  bf16x8 a[4][2], be[2][2], bo[2][2];
  // "P4 of k-tile -1": B0(0) -> the even k-tiles' B0 registers
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    be[j][0] = P8_RD(smem + OFF_B0 + j * 2048, ob0);
    be[j][1] = P8_RD(smem + OFF_B0 + j * 2048, ob1);
  }
  if constexpr (STAMP) ts_[1] = __builtin_amdgcn_s_memrealtime();
  int t = 0;
  for (; t < nk - 2; t += 2) {
    P8B_TILE(0, 0, t, be, bo);
    P8B_TILE(0, 1, t + 1, bo, be);
  }
  P8B_TILE(1, 0, t, be, bo);
  P8B_TILE(2, 1, t + 1, bo, be);
  if (wm == 0) P8_BAR();  // balance group 1's extra barrier
  __syncthreads();        // operand buffers become epilogue space
  if constexpr (STAMP) ts_[2] = __builtin_amdgcn_s_memrealtime();
  NextTilePf pf_{nullptr, p.pf_next, 0u, 0u};
  if (p.pf_next > 0 && (int)blockIdx.x + 256 < (int)gridDim.x) {
    const TileMap tn = map_tile_fast(p, blockIdx.x + 256, gridDim.x, p.M >> 8, p.N >> 8);
    pf_.src = tid < 256 ? p.A + (size_t)(tn.m * 256 + tid) * p.lda : p.W + (size_t)(tn.n * 256 + tid - 256) * p.ldw;
  }
  if constexpr (EPI == EPI_RESID) {
    gemm_epilogue_resid_sq(p, acc, m0 + wm * 128, n0 + wn * 64, wave, lane, smem, &pf_);
  } else {
    f32x4(&lo)[4][4] = *reinterpret_cast<f32x4(*)[4][4]>(&acc[0]);
    f32x4(&hi)[4][4] = *reinterpret_cast<f32x4(*)[4][4]>(&acc[4]);
    gemm_epilogue_lds<EPI>(p, lo, m0 + wm * 128, n0 + wn * 64, 0, 0, wave, lane, smem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    gemm_epilogue_lds<EPI>(p, hi, m0 + wm * 128 + 64, n0 + wn * 64, 0, 0, wave, lane, smem, nullptr, &pf_);
  }
  pf_retire(pf_);
  if constexpr (STAMP) {
    ts_[3] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts_[4] = __builtin_amdgcn_s_memrealtime();
    if (p.dbg != nullptr && tid == 0) {  // wave 0 (group 0); one record per workgroup
      unsigned* d_ = p.dbg + (size_t)blockIdx.x * 8;
#pragma unroll
      for (int i = 0; i < 5; ++i) d_[i] = (unsigned)ts_[i];
      d_[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID: wave / simd / cu / sh / se
      d_[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
      d_[7] = ((ts_setup_ - (unsigned)ts_[0]) & 0xffffu) | ((ts_issued_ - (unsigned)ts_[0]) << 16);  // setup done / 12 DMAs issued, ticks after entry
    }
  }
}

#ifdef DVT_LAB
#include "lab/dvt_vit_gemm8t.inc"  // "8t": this kernel as a persistent workgroup with an overlapped tile boundary (round 6, negative)
#endif
#undef P8B_TILE
#undef P8_STAGE
#undef P8_STAGE_RS
#undef P8_RSRC
#undef P8_RD
#undef P8_LANE_SETUP

// (Rounds 3 / 4 built two persistent variants of this kernel -- "8q": register epilogue + tile loop, removed; "4w": four
// waves, deferred epilogue, lab/dvt_vit_gemm4w.inc -- neither beat it: profiles/LOG.md 5 finding 3, profiles/r04/README.md.)
#ifdef DVT_LAB
typedef const __attribute__((address_space(4))) GemmBArgs* kernarg_ptr_t;

// tile of position `id` in the L2-aware order of map_tile (see there)
__device__ __forceinline__ TileMap map_tile_id(int id, int mt, int nt, int group, int mblock) {
  const int per_group = group * mt;
  int g = id / per_group;
  const int full = nt / group;
  int width = group;
  if (g >= full) {
    g = full;
    width = nt - full * group;
  }
  const int rem = id - g * per_group;
  TileMap t;
  if (mblock <= 1) {
    t.m = rem / width;
    t.n = g * group + rem % width;
  } else {
    const int per_block = mblock * width;
    const int mbk = rem / per_block, r2 = rem - mbk * per_block;
    const int left = mt - mbk * mblock, hb = left < mblock ? left : mblock;
    t.n = g * group + r2 / hb;
    t.m = mbk * mblock + r2 % hb;
  }
  return t;
}


#include "lab/dvt_vit_gemm4w.inc"
#include "lab/dvt_vit_gemm8p_lab.inc"
#include "lab/dvt_vit_lab.inc"
#endif

// W bytes kept L2-resident per N-tile group.  4800 KiB = every N tile of a K = 768 GEMM in ONE group (qkv: 9
// tiles, fc1: 12): each 393-KB A panel is then fetched once instead of once per group (measured: GEMMs 907 ->
// 931 TF/s in the extractor; 9600 KiB the same, 1200 KiB 900).
int g_vit_group_bytes = 4800 * 1024;
// M panels per block of the tile order, 0 = auto: 4 when a group has >= 6 N tiles (qkv 9, fc1 12), else 1.
// PMC FETCH_SIZE per launch, 128 views (profiles/r02/pmc_vit/): qkv 1.57 -> 1.10 GB, fc1 1.65 -> 1.39 GB with
// blocks of 4; the N = 768 GEMMs (3 N tiles) get WORSE with blocks (1.53 -> 1.68 GB) and keep n-fastest.
// Time moves by +1 % only (919 -> 930 TF/s): the L2 misses are served by the memory-side cache.
int g_vit_mblock = 0;
// bf16 output stores with the non-temporal hint: measured no effect on time or FETCH_SIZE (kept as a knob)
int g_vit_nt_store = 0;
// LayerNorm folded into the qkv / fc1 GEMMs and the proj / fc2 residual epilogues (ln_fold)
int g_vit_fuse_ln = 1;
// schedule mask of attention_kernel_v2 (see its header).  15 = k-step-major S, accumulator-major P.V, no per-tile max tree,
// loop unrolled by two: 874-893 us against 908-928 us for mask 0 at 110 views (profiles/r03/r03i).  The product library
// instantiates mask 15 only; lab builds select others with dvt_tune_set(1, -510 - mask).
int g_vit_attn_mask = 15;
// dvt_tune_set(1, -700 - pct): the first round of 8p workgroups starts spread over pct % of the modelled tile time (k-loop
// 1.68 us per k-tile + epilogue); 0 = all together
int g_vit_stagger_pct = 0;
// dvt_tune_set(1, -531) (default) / (1, -530): dvt_vit_forward's qkv GEMM writes q * log2(e) / 8 and the attention kernel works
// in the log2 domain (attention_v2_body, VAR bit 32) / q as it is and the round-3..5 kernel
int g_vit_attn_log2q = 1;
constexpr float ATT_Q_PRESCALE = 0.125f * 1.4426950408889634f;
int g_vit_pf_next = 0;  // dvt_tune_set(1, -570 - n): GemmBArgs::pf_next (0 = off, the default: measured null, profiles/r06/r07a_*)
constexpr int ATT_L2_VAR = 559;  // schedule mask of the log2-domain attention kernel the product launches (attention_v2_body)
#ifdef DVT_LAB
int g_vit_tpw = 0;           // 4w kernel: target tiles per workgroup, 0 = auto (dvt_tune_set(1, -200 - n))
int g_vit_attn_l2_mask = ATT_L2_VAR;  // dvt_tune_set(1, -540 - x): the product's mask (559 = 1 + 2 + 4 + 8 + 32 + 512) with the bits x toggled: 2 = P.V
                                      // fragment by fragment, 512 = no group pattern for the K reads, 514 = both, 128 / 256 / 384 = ablations (idle
                                      // waves compute / whole tail tile / both)
int g_vit_epi_abl = 0;       // dvt_tune_set(1, -560 - mask): GemmBArgs::abl
int g_vit_attn_variant = 2;  // dvt_tune_set(1, -500 - v): 2 (default) = attention_kernel_v2, 1 = the round-2 kernel
int g_vit_abl4w = 0;         // dvt_tune_set(1, -300 - mask) while a 4w schedule (6..9) is selected: its ablation mask (EPI_BIAS, timing only)
int g_vit_8p_build = 0;      // ... while schedule 5 is selected: timing build of the 8p kernel (3 stamps, 6..9 ablations; EPI_BIAS only)
unsigned* g_vit_dbg = nullptr;  // dvt_vit_debug_buffer
int g_vit_w4_grid = 0;       // dvt_tune_set(1, -600 - n): workgroups of the 4w kernel (0 = auto: a whole number per CU)
#endif

template <int EPI>
int launch_gemm(const GemmBArgs& a0, hipStream_t s) {
  if (a0.M % GBM || a0.N % GBN || a0.K % GBK || a0.M <= 0) return DVT_E_BADARG;
  if (IS_X3(EPI) && a0.M % 256) return DVT_E_BADARG;  // written for the LDS-staged epilogues (256-row tiles) only
  GemmBArgs a = a0;
  if (!a.lda) a.lda = a.K;
  if (!a.ldw) a.ldw = a.K;
  a.dim_ok_sq = (!IS_QKV(EPI)) || (a.dim % 256 == 0);
  {
    const int nt = a.N / GBN;
    int g = g_vit_group_bytes / (GBN * a.K * 2);
    g = g < 1 ? 1 : (g > nt ? nt : g);
    while (g > 1 && nt % g) --g;  // prefer a divisor of the tile count (equal groups)
    a.group = g;
  }
  DvtProbeScope probe(DVT_PROBE_VIT_GEMM, s, a.work > 0.0 ? a.work : 2.0 * a.M * a.N * a.K);
  bool sq = g_vit_gemm_variant >= 4;  // (lab: 5..12 are variants of the same tile)
#ifdef DVT_LAB
  sq = sq || g_vit_gemm_variant == 0;
#endif
  if (a.M % 256 == 0 && a.N % 256 == 0 && a.dim_ok_sq && a.K >= 2 * GBK && (a.K / GBK) % 2 == 0 && sq) {  // (8p: k-tiles in pairs)
    const int nt = a.N / 256;
    // N tiles per group: W slices of a group stay L2-resident, but never fewer than 3 tiles share
    // an A panel (K = 3072: one tile per group re-read A three times from HBM, 1.03 -> 1.23 PF/s)
    int g = g_vit_group_bytes / (256 * a.K * 2);
    g = g < 3 ? 3 : g;
    g = g > nt ? nt : g;
    while (g > 1 && nt % g) --g;
    a.group = g;
    a.mblock = g_vit_mblock > 0 ? g_vit_mblock : (g >= 6 ? 4 : 1);
    a.nt_store = g_vit_nt_store;
    fd_make((unsigned)(a.group * (a.M / 256)), a.fd_pg);
    fd_make((unsigned)(a.mblock * a.group), a.fd_pb);
    fd_make((unsigned)a.group, a.fd_w);
    fd_make((unsigned)(a.mblock > 0 ? a.mblock : 1), a.fd_hb);
    a.tiles_full = nt / a.group * a.group;
    const dim3 grid8((a.M / 256) * nt);
    if (g_vit_stagger_pct > 0 && grid8.x > 512) {  // (a launch of fewer than two rounds has nothing to de-synchronise)
      const double epi_us = EPI == EPI_RESID ? 20.0 : (IS_GELU(EPI) ? 10.5 : 5.6);
      const double tile_us = 1.68 * (a.K / GBK) + epi_us;
      a.stagger_ticks = (int)(tile_us * 100.0 * g_vit_stagger_pct / 100.0);
    }
    a.pf_next = g_vit_pf_next;
#ifdef DVT_LAB
    a.abl = g_vit_epi_abl;
    const int nk = a.K / GBK;
    if constexpr (EPI == EPI_BIAS || EPI == EPI_GELU) {
      if (g_vit_gemm_variant >= 6 && g_vit_gemm_variant <= 9 && a.K >= W4_MIN_K) {
        // 4w: persistent runs of ~`tpw` tiles; the grid is a whole number of workgroups per CU
        const int tiles = (a.M / 256) * nt;
        const int tpw = g_vit_tpw > 0 ? g_vit_tpw : (nk <= 16 ? 3 : 1);
        int per_cu = (tiles + 256 * tpw - 1) / (256 * tpw);
        int nwg = 256 * (per_cu < 1 ? 1 : per_cu);
        if (g_vit_w4_grid > 0) nwg = g_vit_w4_grid;
        if (nwg > tiles) nwg = tiles;
        const dim3 grid(nwg);
        const int var = g_vit_gemm_variant - 6;  // 6: flush per tile, 7: deferred, 8: flush + fast GELU, 9: deferred + fast GELU
        bool abl_done = false;
        if constexpr (EPI == EPI_BIAS) {  // ablations (timing only, results are wrong): dvt_tune_set(1, -300 - mask)
#define W4_ABL(n) if (g_vit_abl4w == n) { hipLaunchKernelGGL((gemm_bf16_kernel_4w<EPI, 1, n>), grid, dim3(256), 0, s, a); abl_done = true; }
          W4_ABL(1) W4_ABL(2) W4_ABL(4) W4_ABL(8) W4_ABL(3) W4_ABL(5) W4_ABL(6) W4_ABL(7) W4_ABL(9) W4_ABL(14) W4_ABL(15)
#undef W4_ABL
        }
        if (abl_done) {
        } else if (var == 0) hipLaunchKernelGGL((gemm_bf16_kernel_4w<EPI, 0>), grid, dim3(256), 0, s, a);
        else if (var == 1) hipLaunchKernelGGL((gemm_bf16_kernel_4w<EPI, 1>), grid, dim3(256), 0, s, a);
        else if (var == 2) hipLaunchKernelGGL((gemm_bf16_kernel_4w<EPI, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_bf16_kernel_4w<EPI, 3>), grid, dim3(256), 0, s, a);
        DVT_CHECK_LAUNCH();
        return 0;
      }
    }
    bool lab_done = false;
    if constexpr (EPI == EPI_BIAS) {  // the timing builds exist for the bias epilogue only (tools/lab_gemm8p_stamps.py)
      if (g_vit_gemm_variant == 5 && g_vit_8p_build != 0) {
        a.dbg = g_vit_dbg;
        switch (g_vit_8p_build) {
          case 3: hipLaunchKernelGGL((gemm_bf16_kernel_8p_lab<EPI, 3>), grid8, dim3(512), 0, s, a); break;
          case 6: hipLaunchKernelGGL((gemm_bf16_kernel_8p_lab<EPI, 6>), grid8, dim3(512), 0, s, a); break;
          case 7: hipLaunchKernelGGL((gemm_bf16_kernel_8p_lab<EPI, 7>), grid8, dim3(512), 0, s, a); break;
          case 8: hipLaunchKernelGGL((gemm_bf16_kernel_8p_lab<EPI, 8>), grid8, dim3(512), 0, s, a); break;
          default: hipLaunchKernelGGL((gemm_bf16_kernel_8p_lab<EPI, 9>), grid8, dim3(512), 0, s, a); break;
        }
        lab_done = true;
      }
    }
    if constexpr (EPI == EPI_BIAS || EPI == EPI_QKV || EPI == EPI_GELU) {
      if (!lab_done && g_vit_gemm_variant == 11 && nk >= 4 && grid8.x > 256) {  // "8t": persistent, overlapped tile boundary
        // workgroups of g_vit_tpw tiles each (0: one workgroup per CU for the whole launch); a multiple of 8 (XCD affinity)
        // generations of 256 workgroups: generation g walks tiles [g, g + 1) x tpw x 256 of the order in tpw lock-step rounds
        const int tiles = (int)grid8.x;
        a.tpw = g_vit_tpw > 0 ? g_vit_tpw : (tiles + 255) / 256;
        const int nwg = (tiles + a.tpw * 256 - 1) / (a.tpw * 256) * 256;
        hipLaunchKernelGGL((gemm_bf16_kernel_8t<EPI>), dim3(nwg), dim3(512), 0, s, a);
        lab_done = true;
      }
    }
    if (!lab_done && g_vit_gemm_variant >= 14 && g_vit_gemm_variant <= 16) {  // nt on the A (14) / W (15) / both (16) operand streams
      if (g_vit_gemm_variant == 14) hipLaunchKernelGGL((gemm_bf16_kernel_8p<EPI, false, 1>), grid8, dim3(512), 0, s, a);
      else if (g_vit_gemm_variant == 15) hipLaunchKernelGGL((gemm_bf16_kernel_8p<EPI, false, 2>), grid8, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((gemm_bf16_kernel_8p<EPI, false, 3>), grid8, dim3(512), 0, s, a);
      lab_done = true;
    }
    if (!lab_done && g_vit_gemm_variant == 12) {  // the product kernel's tile anatomy (wall-clock stamps per workgroup)
      a.dbg = g_vit_dbg;
      hipLaunchKernelGGL((gemm_bf16_kernel_8p<EPI, true>), grid8, dim3(512), 0, s, a);
      lab_done = true;
    }
    if (lab_done) {
    } else if (g_vit_gemm_variant == 13) {  // round 5's walk of the ring (12 / 4 / 8 / 0 reads, run-time parity)
      hipLaunchKernelGGL((gemm_bf16_kernel_8p_lab<EPI, 0>), grid8, dim3(512), 0, s, a);
      lab_done = true;
    } else if (g_vit_gemm_variant == 10) {
      hipLaunchKernelGGL((gemm_bf16_kernel_8p_lab<EPI, 2>), grid8, dim3(512), 0, s, a);
      lab_done = true;
    } else if (g_vit_gemm_variant == 5) {
      hipLaunchKernelGGL((gemm_bf16_kernel_8p_lab<EPI, 1>), grid8, dim3(512), 0, s, a);
      lab_done = true;
    } else if (g_vit_gemm_variant == 0) {
      hipLaunchKernelGGL((gemm_bf16_kernel_sq<EPI>), grid8, dim3(512), 0, s, a);
      lab_done = true;
    }
    if (lab_done) {
      DVT_CHECK_LAUNCH();
      return 0;
    }
#endif
    hipLaunchKernelGGL((gemm_bf16_kernel_8p<EPI>), grid8, dim3(512), 0, s, a);
    DVT_CHECK_LAUNCH();
    return 0;
  }
  if (a.M % G2_BM == 0 && g_vit_gemm_variant != 1) {
    const int tiles = (a.M / G2_BM) * (a.N / GBN);
#ifdef DVT_LAB
    if (g_vit_gemm_variant == 2) {
      hipLaunchKernelGGL((gemm_bf16_kernel_256<EPI>), dim3(tiles), dim3(512), 0, s, a);
      DVT_CHECK_LAUNCH();
      return 0;
    }
#endif
    hipLaunchKernelGGL((gemm_bf16_kernel_pp<EPI>), dim3(tiles), dim3(512), 0, s, a);
    DVT_CHECK_LAUNCH();
    return 0;
  }
  if (IS_X3(EPI)) return DVT_E_BADARG;  // the 128 x 128 kernel's register epilogue does not write the split outputs
  const int tiles = (a.M / GBM) * (a.N / GBN);
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3(tiles), dim3(256), 0, s, a);
  DVT_CHECK_LAUNCH();
  return 0;
}

// ======================================================================================
// im2col for the patch embedding (Conv2d 3 -> dim, kernel = patch, stride)
// ======================================================================================
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img,
                                                     bf16_t* __restrict__ col, DvtVitConfig c, int real_rows) {
  const int t = blockIdx.x;  // token row in [0, batch*s_pad), then the phantom rows up to a whole 256-row tile
  const int b = t / c.s_pad, s = t - b * c.s_pad;
  bf16_t* dst = col + (size_t)t * c.k_patch;
  const int pp = c.patch * c.patch;
  if (t >= real_rows || s < c.n_prefix || s >= c.n_tokens) {
    for (int k = threadIdx.x; k < c.k_patch; k += 256) dst[k] = 0;
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
    dst[k] = f2bf(v);
  }
}

// The common geometry (round 6): patch size a compile-time EVEN constant, stride and image width even -- a thread moves PAIRS
// (k, k + 1) of one patch row (8-byte aligned loads, 4-byte stores) and the k -> (channel, ky, kx) split divides by literals
// (with run-time divisors hipcc spends ~25 VALU instructions per division: the generic kernel ran at 2.3 TB/s of traffic).
template <int PATCH>
__global__ __launch_bounds__(256) void im2col_pairs_kernel(const float* __restrict__ img, bf16_t* __restrict__ col, DvtVitConfig c,
                                                           int real_rows) {
  static_assert(PATCH % 2 == 0, "pairs never straddle a patch row");
  constexpr int PP = PATCH * PATCH;
  const int t = blockIdx.x;
  const int b = t / c.s_pad, s = t - b * c.s_pad;
  uint32_t* dst = reinterpret_cast<uint32_t*>(col + (size_t)t * c.k_patch);
  const int npair = c.k_patch >> 1;
  if (t >= real_rows || s < c.n_prefix || s >= c.n_tokens) {
    for (int q = threadIdx.x; q < npair; q += 256) dst[q] = 0u;
    return;
  }
  const int py = (s - c.n_prefix) / c.grid_w, px = (s - c.n_prefix) - py * c.grid_w;
  const float* src = img + (size_t)b * 3 * c.img_h * c.img_w + (size_t)(py * c.stride) * c.img_w + px * c.stride;
  for (int q = threadIdx.x; q < npair; q += 256) {
    const int k = 2 * q;
    uint32_t w = 0u;
    if (k < 3 * PP) {
      const int ch = k / PP, rem = k - ch * PP, ky = rem / PATCH, kx = rem - ky * PATCH;
      const float2 v = *reinterpret_cast<const float2*>(src + ((size_t)ch * c.img_h + ky) * c.img_w + kx);
      w = pack2(v.x, v.y);
    }
    dst[q] = w;
  }
}

// ======================================================================================
// LayerNorm: fp32 row -> bf16 row (or fp32 output for the final norm), one wave per row
// ======================================================================================
template <bool FINAL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        bf16_t* __restrict__ y_bf16,
                                                        float* __restrict__ y_f32, int rows,
                                                        int dim, float eps, int s_pad,
                                                        int n_tokens, int n_prefix) {
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
      const float4 ww = reinterpret_cast<const float4*>(w)[q];
      const float4 bb = reinterpret_cast<const float4*>(b)[q];
      const float o0 = (v[i].x - mean) * rstd * ww.x + bb.x, o1 = (v[i].y - mean) * rstd * ww.y + bb.y;
      const float o2 = (v[i].z - mean) * rstd * ww.z + bb.z, o3 = (v[i].w - mean) * rstd * ww.w + bb.w;
      if (FINAL) {
        reinterpret_cast<float4*>(y_f32 + (size_t)row * dim)[q] = make_float4(o0, o1, o2, o3);
      } else {
        uint2 pk;
        pk.x = pack2(o0, o1);
        pk.y = pack2(o2, o3);
        reinterpret_cast<uint2*>(y_bf16 + (size_t)row * dim)[q] = pk;
      }
    }
  }
}

__device__ __forceinline__ float max3f(float a, float b, float c) {
  return __builtin_fmaxf(__builtin_fmaxf(a, b), c);  // v_max3_f32
}

// ======================================================================================
// Attention (head_dim 64): one workgroup = 128 queries of one (image, head); 8 waves x 16 queries
// ======================================================================================
constexpr int KV_TILE = 64;
// V^T tile in LDS: 64 rows (d) x 128 B, keys PERMUTED inside a row so that the 8 keys a lane feeds into one
// MFMA k-step -- {4g .. 4g+3} and {16+4g .. 16+4g+3} of the step's 32, the order in which the S^T accumulators
// hold P -- are 16 contiguous bytes (logical chunk 4 ks + g), and the chunk index XOR-ed with (row >> 1) & 7:
// ONE conflict-free ds_read_b128 per fragment.  (Before: four 8-byte reads per fragment pair that the compiler
// merged into ds_read2_b64 -- half the LDS rate, 32-bank mode, 2-way conflicts at the 144-B pitch.  PMC:
// SQ_LDS_BANK_CONFLICT 1.9e8 cycles per launch, the LDS pipe busier than the matrix pipe.)
constexpr int VT_LD = 128;

constexpr int ATT_Q = 128;  // queries per workgroup

#ifdef DVT_LAB
#include "lab/dvt_vit_lab_attn.inc"
#endif


// ---- attention, round 3 ("v2"): the same tiling (128 queries per workgroup, 16 per wave, 64-key tiles, S^T = K.Q^T,
// O^T = V^T.P^T, register-staged K / V^T tiles) with the two serial chains of the v1 loop taken apart:
//   (1) S(t+1) is issued BEFORE the softmax of tile t.  In v1 a wave ran  S MFMAs -> softmax VALU -> PV MFMAs  strictly
//       in sequence, and the per-tile barrier keeps the 8 waves of a workgroup in lock step, so the matrix pipe idled
//       while everybody was in the softmax (SQ_VALU_MFMA_BUSY 22 %).  Now the 8 S MFMAs of the NEXT tile are
//       interleaved with the softmax of the CURRENT one (sched_group_barrier: one MFMA per ~7 VALU), their results are
//       needed an iteration later.  K therefore runs one tile further ahead: three K buffers, two V^T buffers (40 KB).
//   (2) deferred running max (the guide's T13).  A query's 64 keys of a tile live in 4 lanes; v1 reduced the tile max
//       across them with two dependent ds_bpermute round trips, recomputed alpha = exp(m_old - m_new) and rescaled l on
//       EVERY tile.  Now every lane only compares its own 16 logits with the running max: while no logit of the whole
//       wave exceeds it by more than THR = 8 (a wave vote, v_cmp + s_cbranch), nothing is reduced or rescaled and
//       P = exp(s - m_run) <= e^8 (bf16 keeps its relative precision there; accumulation is fp32).  Otherwise -- the
//       first tile, and rarely later -- the exact max is formed with v_permlane16/32_swap (no LDS) and o, l are rescaled.
//       tests/test_gpu_vit.py forces the late-rescale branch with a spiked key row (guide 5.4 rule 26).
constexpr int ATT2_KBUF = 3;

// VAR: experiment mask on top of the v2 structure (dvt_tune_set(1, -510 - mask); 0 = v2 as measured in r03):
//   1  S MFMAs issued k-step-major (the two MFMAs of one accumulator four MFMA slots apart instead of adjacent)
//   2  P.V: all eight V^T fragments read first, MFMAs accumulator-major per k-half (same-accumulator distance 4)
//   4  no per-tile max tree: P is formed against the running max and the tile is redone exactly only when a lane's
//      row sum leaves [0, e^8] (first tile: -1e30 running max -> inf -> exact path)
//   8  tile loop unrolled by two (S / S-next swap roles instead of being copied)
//   16 static priority for the second-dispatched half of the workgroup (waves 4-7)
//   64 two barriers per tile, waves 4-7 one phase behind waves 0-3 (softmax of one group over P.V of the other)
//   32 (round 6, needs 4) "log2 domain": q arrives PRE-SCALED by log2(e) / 8 (the qkv GEMM's epilogue does it on the fp32
//      accumulators, one rounding to bf16 as before), so S' = K.Q'^T is the logit in units of log2, and the S MFMA chain starts
//      from C = -m (the running max, one register quad) instead of 0: the accumulators hold t = s' - m and P = v_exp_f32(t)
//      with NO scale-and-subtract FMA (8 v_pk_fma_f32 of a tile's ~45 VALU issues; the kernel is VALU-issue bound: 16 quarter-rate
//      v_exp + ~30 others against 16 MFMAs per wave and tile).  Tile 0's exact max is formed in the prologue; when the running
//      max grows later (rare: the lane's 16-term row sum left [0, e^8]) this tile's t AND the already issued next tile's are
//      lowered by the growth.  Waves wholly behind the image's rows (s_pad < the block's 128 queries) stage and synchronise
//      but neither multiply nor exponentiate, and a last tile with <= 32 valid keys runs its softmax and P.V on half a tile.
//      Four K buffers and a tile loop unrolled by four (every buffer index a literal), staging unconditional with clamped tile
//      indices.  128 / 256: ablation builds (idle waves compute / whole tail tile).
//   512 (with 32) the scheduler's group pattern for the softmax block also places the next tile's K fragment READS: four up
//      front, then per MFMA slot 4 VALU, the MFMA, one read.  Left alone the reads sink behind the burst of 16 v_exp and every
//      MFMA waits out its own read (profiles/r06/r06s_*: 2856 -> 2799 us per 396-view launch).  The product launches 559 =
//      1 + 2 + 4 + 8 + 32 + 512.
// X3 (the fp32 extractor's opt-in "bf16x3" mode, include/dvt_vit.h): q, k, v arrive as (hi, lo) bf16 pairs of the fp32
// values (qk / vt = hi, qk_lo / vt_lo = lo), S = K_lo.Q_hi + K_hi.Q_lo + K_hi.Q_hi and O += V_lo.P_hi + V_hi.P_lo + V_hi.P_hi
// with P split in registers (3 x the MFMAs, fp32 accumulation, fp32 softmax as before) and `out` is fp32 [T, dim].
template <int VAR, bool X3>
__device__ __forceinline__ void attention_v2_body(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ vt,
                                                  bf16_t* __restrict__ out, int heads, int s_pad, int n_valid,
                                                  const bf16_t* __restrict__ qk_lo, const bf16_t* __restrict__ vt_lo,
                                                  int split_out = 0) {
  constexpr int NV = (VAR & 64) ? 3 : 2;  // V^T buffers: the half-tile offset of the two wave groups needs a third
  constexpr bool L2D = (VAR & 32) != 0;
  constexpr bool SKIPW = L2D && !(VAR & 128), HALFT = L2D && !(VAR & 256);  // (128 / 256: ablation builds without the idle waves / the half tail tile)
  static_assert(!L2D || ((VAR & 4) && !X3 && !(VAR & 64)), "log2-domain build: on the no-max-tree structure, bf16 only");
  // K buffers: three suffice (K runs two tiles ahead); the log2-domain build takes FOUR and walks the tiles in an unrolled
  // loop of four whose counter is a multiple of 4, so that every buffer index ((kt + c) & 3, (kt + c) & 1) is a literal and the
  // LDS addresses are lane constants + immediates (with three buffers each tile spent ~8 VALU + SALU issues on slot offsets)
  constexpr int KBUF = L2D ? 4 : ATT2_KBUF;
  constexpr int KSET = KBUF * KV_TILE * 128, VSET = NV * 64 * VT_LD;  // one precision part: 24 (32) KB + 16 KB
  __shared__ __attribute__((aligned(16))) char smem[(X3 ? 2 : 1) * (KSET + VSET)];
  char* const Kb = smem;
  char* const Vb = smem + KSET;
  char* const Kbl = smem + KSET + VSET;  // X3: the lo parts
  char* const Vbl = Kbl + KSET;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lc = lane & 15;
  const int nqb = (s_pad + ATT_Q - 1) / ATT_Q;  // the last block of an image may hang over its rows (s_pad % 16 == 0: whole waves)
  int id = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int qb = id % nqb, h = (id / nqb) % heads, b = id / (nqb * heads);
  const int dim = heads * 64, ldq = 2 * dim;
  const size_t row0 = (size_t)b * s_pad;

  bf16x8 qf[2], qfl[2];
  auto load_q = [&](const bf16_t* base, bf16x8 (&dst)[2]) {  // * head_dim^-0.5 = 2^-3: exact in bf16
    const bf16_t* qrow = base + (row0 + qb * ATT_Q + wave * 16 + lc) * ldq + h * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      union { bf16x8 v; uint32_t u[4]; } raw;
      raw.v = *reinterpret_cast<const bf16x8*>(qrow + ks * 32 + g * 8);
      if constexpr (!L2D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = __uint_as_float(raw.u[j] << 16) * 0.125f;
          const float hi = __uint_as_float(raw.u[j] & 0xffff0000u) * 0.125f;
          raw.u[j] = pack2(lo, hi);
        }
      }
      dst[ks] = raw.v;
    }
  };
  load_q(qk, qf);
  if constexpr (X3) load_q(qk_lo, qfl);
  const bf16_t* kbase = qk + row0 * ldq + dim + h * 64;
  const bf16_t* vbase = vt + ((size_t)(b * heads + h) * 64) * s_pad;

  f32x4 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = L2D ? 0.f : -1e30f, l_run = 0.f;  // (L2D: in units of log2; tile 0's exact max is formed in the prologue)
  f32x4 cinit = (f32x4){0.f, 0.f, 0.f, 0.f};      // L2D: -m_run in every element, the C operand an S chain starts from
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  // L2D: a wave whose 16 queries lie wholly behind the image's rows (the last block of an image where s_pad % 128 != 0)
  // only stages and synchronises (wave-uniform)
  const bool active = !SKIPW || qb * ATT_Q + __builtin_amdgcn_readfirstlane(wave) * 16 < s_pad;
  const float LOG2E = 1.4426950408889634f;
  const float THR = 8.0f;

  const int ntiles = (n_valid + KV_TILE - 1) / KV_TILE;
  const int sr0 = tid >> 3, sc = tid & 7;
  const bf16_t* kp0 = kbase + (size_t)sr0 * ldq + sc * 8;
  const bf16_t* vp0 = vbase + (size_t)sr0 * s_pad + sc * 8;
  const int kdo = sr0 * 128 + ((sc ^ (sr0 & 7)) << 4);
  const int vks = sc >> 2, vc = sc & 3, vsw = (sr0 >> 1) & 7;
  const int vdo0 = sr0 * VT_LD + (((vks * 4 + 2 * (vc & 1)) ^ vsw) << 4) + (vc >> 1) * 8;
  const int vdo1 = sr0 * VT_LD + (((vks * 4 + 2 * (vc & 1) + 1) ^ vsw) << 4) + (vc >> 1) * 8;
  uint4 kr0, vr0, kr0l, vr0l;
  const ptrdiff_t klo = X3 ? (qk_lo - qk) : 0, vlo = X3 ? (vt_lo - vt) : 0;  // element offsets hi -> lo arrays
  // L2D: staging loads as (wave-uniform base of the tile) + (32-bit lane offset): the base advances on the scalar unit, the
  // lane part is a loop constant (global_load saddr form; the 64-bit per-lane pointers above cost a v_lshl_add_u64 per load)
  const unsigned klane = (unsigned)(sr0 * ldq + sc * 8) * 2u;
  const unsigned vlane = (unsigned)(sr0 * s_pad + sc * 8) * 2u, vlane0 = (unsigned)(sr0 * s_pad) * 2u;
#define A2_LOADK(kt)                                                                                 \
  do {                                                                                               \
    if constexpr (L2D) {                                                                             \
      kr0 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(kbase + (size_t)(kt) * KV_TILE * ldq) + klane); \
    } else {                                                                                         \
      kr0 = *reinterpret_cast<const uint4*>(kp0 + (size_t)(kt) * KV_TILE * ldq);                     \
    }                                                                                                \
    if constexpr (X3) kr0l = *reinterpret_cast<const uint4*>(kp0 + klo + (size_t)(kt) * KV_TILE * ldq); \
  } while (0)
// (a lane's 8 keys lie wholly inside or wholly behind the image's s_pad keys (s_pad % 8 == 0).  Behind them -- the last tile
// where s_pad % 64 != 0 -- the lane re-reads a chunk of real keys instead: those keys' P is 0, but what lies behind a V^T row
// is the next row / head / image or unwritten workspace, and 0 x NaN would poison the row's output.  "Re-reads": the row's
// first chunk, which always exists)
// (L2D: the lane re-reads chunk 0 of the SAME tile -- kt * 64 < n_valid <= s_pad, both multiples of 8 -- so that the address
// stays tile base + a non-negative lane offset)
#define A2_LOADV(kt)                                                                   \
  do {                                                                                 \
    if constexpr (L2D) {                                                               \
      const unsigned vo_ = (kt) * KV_TILE + sc * 8 >= s_pad ? vlane0 : vlane;          \
      vr0 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(vbase + (kt) * KV_TILE) + vo_); \
    } else {                                                                           \
      const int vk_ = (kt) * KV_TILE + sc * 8 >= s_pad ? -sc * 8 : (kt) * KV_TILE;     \
      vr0 = *reinterpret_cast<const uint4*>(vp0 + vk_);                                \
      if constexpr (X3) vr0l = *reinterpret_cast<const uint4*>(vp0 + vlo + vk_);       \
    }                                                                                  \
  } while (0)
#define A2_STOREK(kt)                                                                                   \
  do {                                                                                                  \
    *reinterpret_cast<uint4*>(Kb + ((kt) % KBUF) * (KV_TILE * 128) + kdo) = kr0;                   \
    if constexpr (X3) *reinterpret_cast<uint4*>(Kbl + ((kt) % KBUF) * (KV_TILE * 128) + kdo) = kr0l; \
  } while (0)
#define A2_STOREV(kt)                                                                     \
  do {                                                                                    \
    char* vb_ = Vb + ((kt) % NV) * (64 * VT_LD);                                          \
    *reinterpret_cast<uint2*>(vb_ + vdo0) = make_uint2(vr0.x, vr0.y);                     \
    *reinterpret_cast<uint2*>(vb_ + vdo1) = make_uint2(vr0.z, vr0.w);                     \
    if constexpr (X3) {                                                                   \
      char* vl_ = Vbl + ((kt) % NV) * (64 * VT_LD);                                       \
      *reinterpret_cast<uint2*>(vl_ + vdo0) = make_uint2(vr0l.x, vr0l.y);                 \
      *reinterpret_cast<uint2*>(vl_ + vdo1) = make_uint2(vr0l.z, vr0l.w);                 \
    }                                                                                     \
  } while (0)
  // S^T of tile kt: acc s[mt][r] <-> key = 16*mt + 4*g + r, q = lc
#define A2_S(dst, kt)                                                                                              \
  do {                                                                                                             \
    const char* Ks_ = Kb + ((kt) % KBUF) * (KV_TILE * 128);                                                   \
    if constexpr (X3) {                                                                                            \
      const char* Kl_ = Kbl + ((kt) % KBUF) * (KV_TILE * 128);                                                \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                           \
        bf16x8 kh_[4], kl_[4];                                                                                     \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                         \
          const int krow = mt * 16 + lc, ko_ = krow * 128 + (((ks * 4 + g) ^ (krow & 7)) << 4);                     \
          kh_[mt] = *reinterpret_cast<const bf16x8*>(Ks_ + ko_);                                                   \
          kl_[mt] = *reinterpret_cast<const bf16x8*>(Kl_ + ko_);                                                   \
          if (ks == 0) dst[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                           \
          dst[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl_[mt], qf[ks], dst[mt], 0, 0, 0);                    \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                           \
          dst[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh_[mt], qfl[ks], dst[mt], 0, 0, 0);                   \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                           \
          dst[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh_[mt], qf[ks], dst[mt], 0, 0, 0);                    \
      }                                                                                                            \
    } else if constexpr (VAR & 1) {                                                                                       \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                           \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                         \
          const int krow = mt * 16 + lc;                                                                           \
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks_ + krow * 128 + (((ks * 4 + g) ^ (krow & 7)) << 4)); \
          if (ks == 0) dst[mt] = L2D ? cinit : zero4;                                                              \
          dst[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], dst[mt], 0, 0, 0);                         \
        }                                                                                                          \
      }                                                                                                            \
    } else {                                                                                                       \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                             \
      dst[mt] = L2D ? cinit : zero4;                                                                               \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                           \
        const int krow = mt * 16 + lc;                                                                             \
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks_ + krow * 128 + (((ks * 4 + g) ^ (krow & 7)) << 4)); \
        dst[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], dst[mt], 0, 0, 0);                           \
      }                                                                                                            \
    }                                                                                                              \
    }                                                                                                              \
  } while (0)

  // prologue: K0, V0, K1 staged; S(0); K2 / V1 in flight in registers
  A2_LOADK(0);
  A2_LOADV(0);
  A2_STOREK(0);
  A2_STOREV(0);
  if constexpr (L2D) {  // (tile indices clamped to the last tile, see the tile body)
    const int last = ntiles - 1;
    A2_LOADK(1 < last ? 1 : last);
    A2_STOREK(1);
    A2_LOADK(2 < last ? 2 : last);
    A2_LOADV(1 < last ? 1 : last);
  } else {
    if (ntiles > 1) {
      A2_LOADK(1);
      A2_STOREK(1);
    }
    if (ntiles > 2) A2_LOADK(2);
    if (ntiles > 1) A2_LOADV(1);
  }
  __syncthreads();
  f32x4 sA[4], sB[4];
  // the exact maximum of a query's logits over its 4 lanes (lc + 16 g): swap rows 0<->1 / 2<->3, then the wave halves (no LDS)
  auto lane_quad_max = [&](float v) {
    const unsigned u = __float_as_uint(v);
    const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
    const unsigned u2 = __float_as_uint(v);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    return fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
  };
  auto max16 = [&](const f32x4 (&s)[4]) {
    float tmax = __builtin_fmaxf(__builtin_fmaxf(s[0][0], s[0][1]), s[0][2]);
    tmax = max3f(tmax, s[0][3], s[1][0]);
    tmax = max3f(tmax, s[1][1], s[1][2]);
    tmax = max3f(tmax, s[1][3], s[2][0]);
    tmax = max3f(tmax, s[2][1], s[2][2]);
    tmax = max3f(tmax, s[2][3], s[3][0]);
    tmax = max3f(tmax, s[3][1], s[3][2]);
    return fmaxf(tmax, s[3][3]);
  };
  if (active) A2_S(sA, 0);
  if constexpr (L2D) {
    if (active) {
      // tile 0: the running max starts at the tile's exact max (every later tile only checks that nothing grew past e^8)
      if (n_valid < KV_TILE) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (mt * 16 + 4 * g + r >= n_valid) sA[mt][r] = -1e30f;
      }
      m_run = lane_quad_max(max16(sA));
      cinit = (f32x4){-m_run, -m_run, -m_run, -m_run};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) sA[mt] += cinit;
    }
  }
  if constexpr (VAR & 16) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  }
  if constexpr (VAR & 64) {
    // ping-pong: waves 4-7 (the second wave of every SIMD) run one phase behind waves 0-3, so that one group's
    // softmax (VALU) phase meets the other group's P.V (MFMA + LDS) phase on every SIMD
    if (wave >= 4) __syncthreads();
  }
  union PF { bf16x8 v; uint32_t u[4]; };
  // one tile.  LAST: the final tile (padding keys masked, no next tile to start).  From the vote onwards the body of
  // the common case is ONE basic block, so that the scheduler hints can interleave the next tile's S MFMAs with it.
  // (Tried and dropped, profiles/r03: also deferring P.V by one tile so that the block carries 16 independent MFMAs --
  // +1 % with hints, -7 % with a hand-placed MFMA / exp interleave whose LDS reads ran only two slots ahead.)
  auto tile = [&](int kt, auto last_tag, f32x4 (&s)[4], f32x4 (&sn)[4]) {
    constexpr bool LAST = decltype(last_tag)::value;
    if constexpr (L2D) {
      // unconditional staging (no branches, no phis around the in-flight registers): behind the last tile the LAST tile is
      // loaded again and lands in buffers nobody reads any more (K: four buffers, the readers are at (kt + 1) & 3 at most; V^T:
      // the buffer this tile does not read)
      const int last = ntiles - 1;
      A2_STOREK(kt + 2);
      A2_STOREV(kt + 1);
      if constexpr (!LAST) {
        const int k3 = kt + 3 < last ? kt + 3 : last, v2 = kt + 2 < last ? kt + 2 : last;
        A2_LOADK(k3);
        A2_LOADV(v2);
      }
    } else {
    if (kt + 2 < ntiles) A2_STOREK(kt + 2);  // loaded an iteration ago; that buffer was last read two barriers back
    if (kt + 1 < ntiles) A2_STOREV(kt + 1);
    if (kt + 3 < ntiles) A2_LOADK(kt + 3);
    if (kt + 2 < ntiles) A2_LOADV(kt + 2);
    }
    if constexpr (LAST) {
      const int kbase_idx = kt * KV_TILE;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kbase_idx + mt * 16 + 4 * g + r >= n_valid) s[mt][r] = -1e30f;
    }
    // exact running max over the query's 4 lanes (lc + 16 g) + rescale of o, l (everything still at the old max is scaled
    // exactly once: P.V of the previous tile is complete)
    auto exact_max = [&]() {
      if constexpr (L2D) {
        // s holds t = s' - m_run: the max grows by d = max(t) where that is positive; everything that was formed against
        // the old max is lowered by d -- o and l (scaled by 2^-d; P.V of the previous tile is complete), this tile's t, and
        // the NEXT tile's, whose chain already started from the old -m_run
        const float d = fmaxf(lane_quad_max(max16(s)), 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] *= alpha;
        m_run += d;
        cinit = (f32x4){-m_run, -m_run, -m_run, -m_run};
        const f32x4 d4 = {d, d, d, d};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          s[mt] -= d4;
          if constexpr (!LAST) sn[mt] -= d4;
        }
        return;
      }
      float tmax = __builtin_fmaxf(__builtin_fmaxf(s[0][0], s[0][1]), s[0][2]);
      tmax = max3f(tmax, s[0][3], s[1][0]);
      tmax = max3f(tmax, s[1][1], s[1][2]);
      tmax = max3f(tmax, s[1][3], s[2][0]);
      tmax = max3f(tmax, s[2][1], s[2][2]);
      tmax = max3f(tmax, s[2][3], s[3][0]);
      tmax = max3f(tmax, s[3][1], s[3][2]);
      tmax = fmaxf(tmax, s[3][3]);
      const unsigned u = __float_as_uint(tmax);
      const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
      tmax = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
      const unsigned u2 = __float_as_uint(tmax);
      const auto r32 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
      tmax = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i][0] *= alpha;
        o[i][1] *= alpha;
        o[i][2] *= alpha;
        o[i][3] *= alpha;
      }
      m_run = m_new;
    };
    if constexpr (!(VAR & 4)) {
    float tmax = __builtin_fmaxf(__builtin_fmaxf(s[0][0], s[0][1]), s[0][2]);
      tmax = max3f(tmax, s[0][3], s[1][0]);
      tmax = max3f(tmax, s[1][1], s[1][2]);
      tmax = max3f(tmax, s[1][3], s[2][0]);
      tmax = max3f(tmax, s[2][1], s[2][2]);
      tmax = max3f(tmax, s[2][3], s[3][0]);
      tmax = max3f(tmax, s[3][1], s[3][2]);
      tmax = fmaxf(tmax, s[3][3]);
      if (!__all(tmax <= m_run + THR)) {  // wave-uniform, rare after the first tile
        // exact max over the query's 4 lanes (lc + 16 g): swap rows 0<->1 / 2<->3, then the wave halves
        const unsigned u = __float_as_uint(tmax);
        const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        tmax = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
        const unsigned u2 = __float_as_uint(tmax);
        const auto r32 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
        tmax = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        l_run *= alpha;  // everything still at the old max is scaled exactly once: o and l (P.V of the previous tile is complete)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[i][0] *= alpha;
          o[i][1] *= alpha;
          o[i][2] *= alpha;
          o[i][3] *= alpha;
        }
        m_run = m_new;
      }
    }
    if constexpr (HALFT && LAST) {
      if (n_valid - kt * KV_TILE <= 32) {
        // (wave-uniform) keys 32..63 of the last tile are all padding (ViT-B/14 at 518 px: 1370 = 21 x 64 + 26): softmax, pack
        // and P.V on the tile's first k-half only -- 8 v_exp, one P fragment, 4 V^T fragments, 4 MFMAs
        float hp[2][4];
        f32x2 hs2;
        auto half_softmax = [&]() {
          hs2 = (f32x2){0.f, 0.f};
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              f32x2 e;
              e.x = __builtin_amdgcn_exp2f(s[mt][2 * h2]);
              e.y = __builtin_amdgcn_exp2f(s[mt][2 * h2 + 1]);
              hs2 += e;
              hp[mt][2 * h2] = e.x;
              hp[mt][2 * h2 + 1] = e.y;
            }
        };
        half_softmax();
        float hsum = hs2.x + hs2.y;
        if (!__all(hsum <= 2980.0f)) {
          exact_max();
          half_softmax();
          hsum = hs2.x + hs2.y;
        }
        l_run += hsum;
        union { bf16x8 v; uint32_t u[4]; } hf;
        hf.u[0] = pack2(hp[0][0], hp[0][1]); hf.u[1] = pack2(hp[0][2], hp[0][3]);
        hf.u[2] = pack2(hp[1][0], hp[1][1]); hf.u[3] = pack2(hp[1][2], hp[1][3]);
        const char* Vh = Vb + (kt % NV) * (64 * VT_LD);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int vr = mt * 16 + lc, vs_ = (vr >> 1) & 7;
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vh + vr * VT_LD + ((g ^ vs_) << 4));
          o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, hf.v, o[mt], 0, 0, 0);
        }
        __syncthreads();
        return;
      }
    }
    if constexpr (L2D && !LAST && (VAR & 16384)) {
      // (experiment, 16384) the softmax SPLIT over the tile's two blocks, so that both carry matrix work AND exps: block 1 = the
      // next tile's 8 S MFMAs beside the exps of keys 0..31 (P fragment 0); block 2 = the first four P.V MFMAs (fragment 0)
      // beside the exps of keys 32..63 (fragment 1), then the other four.  Each half has its own lane-sum vote.  A growth of
      // the max found by the SECOND vote rescales o and l -- which then already hold the first half's contribution, formed
      // against the same old max and finite (its vote passed): scaled by the same 2^-d it is what the new max asks for.
      union PH { bf16x8 v; uint32_t u[4]; };
      float hp[4][4];
      auto soft_half = [&](int h) {
        f32x2 a2, b2;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int mt = 2 * h + q;
          f32x2 e0, e1;
          e0.x = __builtin_amdgcn_exp2f(s[mt][0]);
          e0.y = __builtin_amdgcn_exp2f(s[mt][1]);
          e1.x = __builtin_amdgcn_exp2f(s[mt][2]);
          e1.y = __builtin_amdgcn_exp2f(s[mt][3]);
          if (q == 0) {
            a2 = e0;
            b2 = e1;
          } else {
            a2 += e0;
            b2 += e1;
          }
          hp[mt][0] = e0.x; hp[mt][1] = e0.y; hp[mt][2] = e1.x; hp[mt][3] = e1.y;
        }
        a2 += b2;
        return a2.x + a2.y;
      };
      auto pack_half = [&](int h, PH& f) {
        f.u[0] = pack2(hp[2 * h][0], hp[2 * h][1]); f.u[1] = pack2(hp[2 * h][2], hp[2 * h][3]);
        f.u[2] = pack2(hp[2 * h + 1][0], hp[2 * h + 1][1]); f.u[3] = pack2(hp[2 * h + 1][2], hp[2 * h + 1][3]);
      };
      PH f0, f1;
      __builtin_amdgcn_sched_barrier(0);
      A2_S(sn, kt + 1);
      float ps = soft_half(0);
      pack_half(0, f0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!__all(ps <= 2980.0f)) {
        exact_max();
        ps = soft_half(0);
        pack_half(0, f0);
      }
      l_run += ps;
      const char* Vs2 = Vb + (kt % NV) * (64 * VT_LD);
      bf16x8 vf[8];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int vr = mt * 16 + lc, vs_ = (vr >> 1) & 7;
        const char* vrow = Vs2 + vr * VT_LD;
        vf[mt] = *reinterpret_cast<const bf16x8*>(vrow + ((g ^ vs_) << 4));
        vf[4 + mt] = *reinterpret_cast<const bf16x8*>(vrow + (((4 + g) ^ vs_) << 4));
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[mt], f0.v, o[mt], 0, 0, 0);
      float ps1 = soft_half(1);
      pack_half(1, f1);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!__all(ps1 <= 2980.0f)) {
        exact_max();  // (o and l: the first half's share included, see above)
        ps1 = soft_half(1);
        pack_half(1, f1);
      }
      l_run += ps1;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[4 + mt], f1.v, o[mt], 0, 0, 0);
      __syncthreads();
      return;
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (VAR & 4096) __builtin_amdgcn_s_setprio(1);  // (experiment) ... or the S / softmax block does
    if constexpr (!LAST) A2_S(sn, kt + 1);  // K(kt+1) became visible at the previous barrier (or in the prologue)
    const f32x2 l2e2 = {LOG2E, LOG2E};
    float pv[4][4];
    f32x2 ps2;
    auto softmax = [&]() {
      const float mb = m_run * LOG2E;
      const f32x2 nmb2 = {-mb, -mb};
      f32x2 psb;  // L2D: two chains of four packed adds (a single chain of eight needs a wait state between any two), started
                  // from the first pairs instead of 0 + e
      if constexpr (!L2D) ps2 = (f32x2){0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x2 sv = {s[mt][2 * h2], s[mt][2 * h2 + 1]};
          const f32x2 t = L2D ? sv : __builtin_elementwise_fma(sv, l2e2, nmb2);
          f32x2 e;
          e.x = __builtin_amdgcn_exp2f(t.x);
          e.y = __builtin_amdgcn_exp2f(t.y);
          if constexpr (L2D) {
            if (mt == 0) (h2 == 0 ? ps2 : psb) = e;
            else (h2 == 0 ? ps2 : psb) += e;
          } else {
            ps2 += e;
          }
          pv[mt][2 * h2] = e.x;
          pv[mt][2 * h2 + 1] = e.y;
        }
      if constexpr (L2D) ps2 += psb;
    };
    softmax();
    float psum = ps2.x + ps2.y;
    PF pf0, pf1, pl0, pl1;
    auto pack = [&]() {
      pf0.u[0] = pack2(pv[0][0], pv[0][1]); pf0.u[1] = pack2(pv[0][2], pv[0][3]);
      pf0.u[2] = pack2(pv[1][0], pv[1][1]); pf0.u[3] = pack2(pv[1][2], pv[1][3]);
      pf1.u[0] = pack2(pv[2][0], pv[2][1]); pf1.u[1] = pack2(pv[2][2], pv[2][3]);
      pf1.u[2] = pack2(pv[3][0], pv[3][1]); pf1.u[3] = pack2(pv[3][2], pv[3][3]);
      if constexpr (X3) {  // P = hi + lo: lo = bf16(p - float(hi)), the subtraction is exact
        auto lo2 = [](uint32_t hb, float a, float b) {
          return pack2(a - __uint_as_float(hb << 16), b - __uint_as_float(hb & 0xffff0000u));
        };
        pl0.u[0] = lo2(pf0.u[0], pv[0][0], pv[0][1]); pl0.u[1] = lo2(pf0.u[1], pv[0][2], pv[0][3]);
        pl0.u[2] = lo2(pf0.u[2], pv[1][0], pv[1][1]); pl0.u[3] = lo2(pf0.u[3], pv[1][2], pv[1][3]);
        pl1.u[0] = lo2(pf1.u[0], pv[2][0], pv[2][1]); pl1.u[1] = lo2(pf1.u[1], pv[2][2], pv[2][3]);
        pl1.u[2] = lo2(pf1.u[2], pv[3][0], pv[3][1]); pl1.u[3] = lo2(pf1.u[3], pv[3][2], pv[3][3]);
      }
    };
    if constexpr (!(VAR & 1024)) pack();
    if constexpr (!LAST && (VAR & 32768)) {
      // (experiment) LLVM's own interleaving strategies for this block instead of the group pattern: 2 = MFMAExpInterleave,
      // written for exp-heavy attention loops (VAR & 65536: 3 = its simple form)
      __builtin_amdgcn_iglp_opt((VAR & 65536) ? 3 : 2);
    } else if constexpr (!LAST && (VAR & 512)) {
      // (experiment) the K fragment reads placed too: four up front, then per MFMA slot 4 VALU, the MFMA, the read that
      // re-fills its fragment registers -- left alone the reads sink behind the exps and every MFMA waits for its own read
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    } else if constexpr (!LAST) {
      // scheduler shape for the block above: one S MFMA of the next tile per ~6 VALU of this tile's softmax
#pragma unroll
      for (int i = 0; i < (X3 ? 24 : 8); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, (X3 || L2D) ? 4 : (VAR & 4) ? 5 : 6, 0);  // VALU
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (VAR & 4096) __builtin_amdgcn_s_setprio(0);
    if constexpr (VAR & 4) {
      // every P of this lane is within [0, e^8] iff its 16-term sum is (terms are >= 0; inf / NaN fail the compare)
      if (!__all(psum <= 2980.0f)) {
        exact_max();
        softmax();
        psum = ps2.x + ps2.y;
        if constexpr (!(VAR & 1024)) pack();
      }
    }
    l_run += psum;
    // ---- O^T[d][q] += V^T . P^T
    if constexpr (VAR & 64) __syncthreads();  // phase boundary: the other wave group starts its softmax phase here
    const char* Vs = Vb + (kt % NV) * (64 * VT_LD);
    if constexpr (X3) {
      const char* Vl = Vbl + (kt % NV) * (64 * VT_LD);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int vr = mt * 16 + lc, vs_ = (vr >> 1) & 7;
        const int o0_ = vr * VT_LD + ((g ^ vs_) << 4), o1_ = vr * VT_LD + (((4 + g) ^ vs_) << 4);
        const bf16x8 vh0 = *reinterpret_cast<const bf16x8*>(Vs + o0_), vh1 = *reinterpret_cast<const bf16x8*>(Vs + o1_);
        const bf16x8 vl0 = *reinterpret_cast<const bf16x8*>(Vl + o0_), vl1 = *reinterpret_cast<const bf16x8*>(Vl + o1_);
        o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl0, pf0.v, o[mt], 0, 0, 0);
        o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl1, pf1.v, o[mt], 0, 0, 0);
        o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh0, pl0.v, o[mt], 0, 0, 0);
        o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh1, pl1.v, o[mt], 0, 0, 0);
        o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh0, pf0.v, o[mt], 0, 0, 0);
        o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh1, pf1.v, o[mt], 0, 0, 0);
      }
    } else if constexpr (VAR & 2) {
      if constexpr (VAR & 2048) __builtin_amdgcn_s_setprio(1);  // (experiment) the P.V block wins the issue arbitration
      bf16x8 vf[8];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int vr = mt * 16 + lc, vs_ = (vr >> 1) & 7;
        const char* vrow = Vs + vr * VT_LD;
        vf[mt] = *reinterpret_cast<const bf16x8*>(vrow + ((g ^ vs_) << 4));
        vf[4 + mt] = *reinterpret_cast<const bf16x8*>(vrow + (((4 + g) ^ vs_) << 4));
      }
      if constexpr (VAR & 1024) pack();  // (experiment) P is packed BEHIND the V^T fragment reads: the eight v_cvt_pk cover their latency
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[mt], pf0.v, o[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[4 + mt], pf1.v, o[mt], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // the eight LDS reads
      if constexpr (VAR & 1024) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // the eight v_cvt_pk
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);  // then the MFMAs in source order
      if constexpr (VAR & 2048) __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int vr = mt * 16 + lc, vs_ = (vr >> 1) & 7;
        const char* vrow = Vs + vr * VT_LD;
        const bf16x8 vf0 = *reinterpret_cast<const bf16x8*>(vrow + ((g ^ vs_) << 4));
        const bf16x8 vf1 = *reinterpret_cast<const bf16x8*>(vrow + (((4 + g) ^ vs_) << 4));
        o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, pf0.v, o[mt], 0, 0, 0);
        o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, pf1.v, o[mt], 0, 0, 0);
      }
    }
    if constexpr (VAR & 8192) {  // TIMING ONLY (developer library): no tile barrier -- what the lock step of a workgroup's 8 waves costs
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      __syncthreads();
    }
  };
  if constexpr (SKIPW) {
    if (!active) {
      // an idle wave's loop: its share of the staging and the tile barriers (s_barrier counts arrivals, whatever the PC)
      const int last = ntiles - 1;
      for (int kt = 0; kt < ntiles; ++kt) {
        A2_STOREK(kt + 2);
        A2_STOREV(kt + 1);
        A2_LOADK(kt + 3 < last ? kt + 3 : last);
        A2_LOADV(kt + 2 < last ? kt + 2 : last);
        __syncthreads();
      }
      return;
    }
  }
  if constexpr (L2D) {
    int kt = 0;
    for (; kt + 4 <= ntiles - 1; kt += 4) {
      tile(kt, std::false_type{}, sA, sB);
      tile(kt + 1, std::false_type{}, sB, sA);
      tile(kt + 2, std::false_type{}, sA, sB);
      tile(kt + 3, std::false_type{}, sB, sA);
    }
    for (; kt < ntiles - 1; ++kt) {  // 0..3 whole tiles before the last one (buffer indices at run time)
      tile(kt, std::false_type{}, sA, sB);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) sA[mt] = sB[mt];
    }
    tile(ntiles - 1, std::true_type{}, sA, sB);
  } else if constexpr (VAR & 8) {
    int kt = 0;
    for (; kt + 2 <= ntiles - 1; kt += 2) {
      tile(kt, std::false_type{}, sA, sB);
      tile(kt + 1, std::false_type{}, sB, sA);
    }
    if (kt < ntiles - 1) {
      tile(kt, std::false_type{}, sA, sB);
      tile(ntiles - 1, std::true_type{}, sB, sA);
    } else {
      tile(ntiles - 1, std::true_type{}, sA, sB);
    }
  } else {
    for (int kt = 0; kt < ntiles - 1; ++kt) {
      tile(kt, std::false_type{}, sA, sB);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) sA[mt] = sB[mt];
    }
    tile(ntiles - 1, std::true_type{}, sA, sB);
  }
  if constexpr (VAR & 16) __builtin_amdgcn_s_setprio(0);
  if constexpr (VAR & 64) {
    if (wave < 4) __syncthreads();
  }
#undef A2_LOADK
#undef A2_LOADV
#undef A2_STOREK
#undef A2_STOREV
#undef A2_S
  if (qb * ATT_Q + wave * 16 >= s_pad) return;  // a wave past the image's rows (its 16 "queries" were the next image's): nothing to store
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
  if constexpr (X3) {
    if (split_out) {  // the proj GEMM's A row [hi | hi | lo] (3 dim wide) instead of fp32: no split pass in between
      bf16_t* r3 = out + (row0 + qb * ATT_Q + wave * 16 + lc) * (3 * dim) + h * 64;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float a0 = o[mt][0] * inv, a1 = o[mt][1] * inv, a2 = o[mt][2] * inv, a3 = o[mt][3] * inv;
        uint2 hh, ll;
        hh.x = pack2(a0, a1);
        hh.y = pack2(a2, a3);
        ll.x = pack2(a0 - __uint_as_float(hh.x << 16), a1 - __uint_as_float(hh.x & 0xffff0000u));
        ll.y = pack2(a2 - __uint_as_float(hh.y << 16), a3 - __uint_as_float(hh.y & 0xffff0000u));
        bf16_t* c = r3 + mt * 16 + 4 * g;
        *reinterpret_cast<uint2*>(c) = hh;
        *reinterpret_cast<uint2*>(c + dim) = hh;
        *reinterpret_cast<uint2*>(c + 2 * dim) = ll;
      }
      return;
    }
    float* orow = reinterpret_cast<float*>(out) + (row0 + qb * ATT_Q + wave * 16 + lc) * dim + h * 64;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      *reinterpret_cast<float4*>(orow + mt * 16 + 4 * g) =
          make_float4(o[mt][0] * inv, o[mt][1] * inv, o[mt][2] * inv, o[mt][3] * inv);
    return;
  }
  bf16_t* orow = out + (row0 + qb * ATT_Q + wave * 16 + lc) * dim + h * 64;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    uint2 pk;
    pk.x = pack2(o[mt][0] * inv, o[mt][1] * inv);
    pk.y = pack2(o[mt][2] * inv, o[mt][3] * inv);
    *reinterpret_cast<uint2*>(orow + mt * 16 + 4 * g) = pk;
  }
}

template <int VAR>
__global__ __launch_bounds__(512) void attention_kernel_v2(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ vt,
                                                           bf16_t* __restrict__ out, int heads, int s_pad, int n_valid) {
  attention_v2_body<VAR, false>(qk, vt, out, heads, s_pad, n_valid, nullptr, nullptr);
}
// the log2-domain builds (VAR bit 32).  Left alone hipcc takes 134-136 registers for them (the -m quad, twice in the loop
// unrolled by two) = ONE workgroup per CU; told to fit four waves per SIMD it parks three dwords (the output row pointer, across
// the loop) and a few more in the last-tile code in scratch and keeps the tile loop free of scratch traffic (ISA checked).
template <int VAR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attention_kernel_l2(
    const bf16_t* __restrict__ qk, const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, int heads, int s_pad, int n_valid) {
  static_assert(VAR & 32, "log2-domain builds only");
  attention_v2_body<VAR, false>(qk, vt, out, heads, s_pad, n_valid, nullptr, nullptr);
}
// the split-operand build: 156 registers, one 80-KB workgroup per CU (capped at 128 registers -- 24 dwords of scratch,
// two workgroups per CU -- it measured 1.71-1.76 ms per 64 views against 1.30-1.53: dropped)
__global__ __launch_bounds__(512) void attention_kernel_v2_x3(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ vt,
                                                              bf16_t* __restrict__ out, int heads, int s_pad, int n_valid,
                                                              const bf16_t* __restrict__ qk_lo,
                                                              const bf16_t* __restrict__ vt_lo, int split_out) {
  attention_v2_body<12, true>(qk, vt, out, heads, s_pad, n_valid, qk_lo, vt_lo, split_out);
}

inline int64_t up256b(int64_t x) { return (x + 255) / 256 * 256; }

struct VitWork {
  float* x;
  bf16_t *xn, *qk, *vt, *hid, *col;
  bf16_t* xb;       // bf16(x) for the LayerNorm-folded GEMMs
  float2* st_part;  // [dim / 64][T] partial row sums from the residual epilogues
  float2* stats;    // [T] (mean, rstd)
};

int64_t vit_carve(const DvtVitConfig* c, int batch, char* base, VitWork* w) {
  // rows: whole 256-row GEMM tiles -- an odd batch (s_pad = 1408 = 5.5 tiles) gets 128 phantom rows, computed and
  // never read, so that every batch takes the same kernels (results must not depend on how views are batched)
  const int64_t T = ((int64_t)batch * c->s_pad + 255) / 256 * 256;
  int64_t o = 0;
  auto take = [&](int64_t bytes) {
    char* p = base ? base + o : nullptr;
    o += up256b(bytes);
    return p;
  };
  VitWork t;
  t.x = (float*)take(T * c->dim * 4);
  t.xn = (bf16_t*)take(T * c->dim * 2);
  // (+ 128 rows: where s_pad is not a multiple of 128 the attention kernel's last query block / key tile of the LAST image read
  // up to 127 rows past batch * s_pad -- never used: masked keys, unstored queries -- which T does not always cover)
  t.qk = (bf16_t*)take((T + 128) * 2 * c->dim * 2);
  t.vt = (bf16_t*)take(((int64_t)batch + 1) * c->s_pad * c->dim * 2);  // the phantom rows' V^T lands in image `batch`
  t.hid = (bf16_t*)take(T * c->mlp_dim * 2);
  t.col = (bf16_t*)take(T * c->k_patch * 2);
  t.xb = (bf16_t*)take(T * c->dim * 2);
  t.st_part = (float2*)take((int64_t)(c->dim / 64) * T * 8);
  t.stats = (float2*)take(T * 8);
  if (w) *w = t;
  return o;
}

int check_vit_cfg(const DvtVitConfig* c) {
  if (!c || c->dim <= 0 || c->dim % 128 || c->dim > 1024 || c->heads * 64 != c->dim) return DVT_E_BADARG;
  if (c->depth < 1 || c->depth > DVT_VIT_MAX_DEPTH || c->mlp_dim % 128) return DVT_E_BADARG;
  // token rows per image: a multiple of 32 (bf16 path; dvt_vit_config writes the next multiple of 128, the fp32 paths' need)
  if (c->s_pad % 32 || c->s_pad < c->n_tokens || c->k_patch % 64) return DVT_E_BADARG;
  if (c->n_prefix < 1 || c->n_prefix > 9 || (c->pos_has_cls != 0 && c->pos_has_cls != 1)) return DVT_E_BADARG;
  if (c->n_tokens != c->n_prefix + c->grid_h * c->grid_w) return DVT_E_BADARG;
  return 0;
}

}  // namespace

// dvt_tune_set(1, v).  The product library accepts only values under which every entry point still computes its documented
// result (alternative schedules / orders that the GPU tests hold against the oracle); everything that selects a superseded
// kernel, an experiment or a timing build exists in lab builds (-DDVT_LAB) only and is DVT_E_BADARG here.
int dvt_vit_tune(int v) {
  if (v == -60 || v == -61) {  // LayerNorm kernels (-60) / folded into the GEMMs (-61, default)
    g_vit_fuse_ln = v == -61;
    return 0;
  }
  if (v == -50 || v == -51) {  // non-temporal bf16 output stores off / on
    g_vit_nt_store = v == -51;
    return 0;
  }
  if (v <= -570 && v >= -572) {  // 8p GEMM: L2 prefetch of the next workgroup's first 0 / 1 / 2 k-tiles (NextTilePf; results do not depend on it)
    g_vit_pf_next = -570 - v;
    return 0;
  }
  if (v == -530 || v == -531) {  // bf16 extractor: attention on q as it is (-530) / on q pre-scaled by log2(e) / 8 (-531, default)
    g_vit_attn_log2q = v == -531;
    return 0;
  }
  if (v == -520 || v == -521) {  // fp32 extractor, bf16x3 mode: exact-fp32 attention (-520) / bf16x3 attention (-521, default)
    g_f32x3_exact_attention = v == -520;
    return 0;
  }
  if (v == -522 || v == -523) {  // ... split kernels (-522) / split epilogues of the qkv and fc1 GEMMs (-523, default)
    g_f32x3_unfused = v == -522;
    return 0;
  }
#ifdef DVT_LAB
  if ((v <= -600 && v > -700) || (v < -1100 && v >= -1100 - 65536)) {  // -600 - n (n < 100) / -1100 - n: the 4w GEMM's grid forced to n workgroups (0 = auto)
    g_vit_w4_grid = v > -700 ? -600 - v : -1100 - v;
    return 0;
  }
  if (v <= -560 && v >= -563) {  // timing-only ablation of the GEMM epilogues (GemmBArgs::abl)
    g_vit_epi_abl = -560 - v;
    return 0;
  }
  if (v == -540 || v == -540 - 2 || v == -540 - 128 || v == -540 - 256 || v == -540 - 384 || v == -540 - 512 || v == -540 - 514 || v == -540 - 1024 || v == -540 - 2048 || v == -540 - 4096 || v == -540 - 8192 || v == -540 - 16384 || v == -540 - 32768 || v == -540 - 32768 - 65536) {
    g_vit_attn_l2_mask = ATT_L2_VAR ^ (-540 - v);
    return 0;
  }
  if (v <= -510 && v > -530) {
    g_vit_attn_mask = -510 - v;
    return 0;
  }
  if (v == -510 - 31 || v == -510 - 64 || v == -510 - 79) {
    g_vit_attn_mask = -510 - v;
    return 0;
  }
  if (v == -501 || v == -502) {
    g_vit_attn_variant = -500 - v;
    return 0;
  }
  if (v <= -300 && v > -399) {  // ablation mask of the selected 4w schedule / timing build of schedule 5; anything else: neither
    const int n = -300 - v;
    g_vit_abl4w = g_vit_8p_build = 0;
    if (n == 0) return 0;
    if (g_vit_gemm_variant >= 6 && g_vit_gemm_variant <= 9) g_vit_abl4w = n;
    else if (g_vit_gemm_variant == 5 && (n == 3 || (n >= 6 && n <= 9))) g_vit_8p_build = n;
    else return DVT_E_BADARG;
    return 0;
  }
  if (v <= -200 && v > -300) {  // -200 - n: target tiles per workgroup of the 4w kernel, 0 = auto
    g_vit_tpw = -200 - v > 64 ? 64 : -200 - v;
    return 0;
  }
#else
  if (v == -502 || v == -510 - 15) return 0;  // the one attention kernel / schedule mask the product library contains
#endif
  if (v <= -700 && v >= -1100) {  // -700 - pct: de-synchronised start of the 8p workgroups (see the kernel), 0 = off
    g_vit_stagger_pct = -700 - v;
    return 0;
  }
  if (v <= -100 && v > -200) {  // -100 - b: M panels per block of the tile order
    g_vit_mblock = -100 - v;  // 0 = auto
    return 0;
  }
  if (v >= 16) {  // values >= 16: L2 group budget in KiB
    g_vit_group_bytes = v * 1024;
    return 0;
  }
#ifdef DVT_LAB
  if (v < 0 || v > 16) return DVT_E_BADARG;
  g_vit_abl4w = g_vit_8p_build = 0;  // an ablation / timing build never survives a change of schedule
#else
  if (v != 1 && v != 3 && v != 4) return DVT_E_BADARG;
#endif
  g_vit_gemm_variant = v;
  return 0;
}

// device buffer of the 8p timing builds (lab builds; the product library has none: DVT_E_BADARG)
extern "C" int dvt_vit_debug_buffer(void* dev_u32) {
#ifdef DVT_LAB
  g_vit_dbg = static_cast<unsigned*>(dev_u32);
  return 0;
#else
  (void)dev_u32;
  return DVT_E_BADARG;
#endif
}

// 1 when this library was built with -DDVT_LAB (the developer build of tools/build_lab.py), else 0
extern "C" int dvt_vit_is_lab_build(void) {
#ifdef DVT_LAB
  return 1;
#else
  return 0;
#endif
}

extern "C" int dvt_vit_struct_sizes(int64_t* out) {
  if (!out) return DVT_E_BADARG;
  out[0] = sizeof(DvtVitConfig);
  out[1] = sizeof(DvtVitBlockWeights);
  out[2] = sizeof(DvtVitWeights);
  return 0;
}

extern "C" int dvt_vit_config(int dim, int depth, int patch, int stride, int img_h, int img_w,
                              DvtVitConfig* c) {
  return dvt_vit_config_reg(dim, depth, patch, stride, img_h, img_w, 0, c);
}

extern "C" int dvt_vit_config_reg(int dim, int depth, int patch, int stride, int img_h, int img_w,
                                  int n_reg_tokens, DvtVitConfig* c) {
  if (!c || n_reg_tokens < 0 || n_reg_tokens > 8 || dim <= 0 || dim % 64 || patch <= 0 || stride <= 0 || img_h < patch || img_w < patch)
    return DVT_E_BADARG;
  c->dim = dim;
  c->depth = depth;
  c->heads = dim / 64;
  c->mlp_dim = 4 * dim;
  c->patch = patch;
  c->stride = stride;
  c->img_h = img_h;
  c->img_w = img_w;
  c->grid_h = (img_h - patch) / stride + 1;  // vit_wrapper.py:84-88 dynamic_feat_size
  c->grid_w = (img_w - patch) / stride + 1;
  c->n_prefix = 1 + n_reg_tokens;
  c->pos_has_cls = n_reg_tokens == 0;  // timm: the reg4 DINOv2 models use no_embed_class=True
  c->n_tokens = c->n_prefix + c->grid_h * c->grid_w;
  c->s_pad = (c->n_tokens + 127) / 128 * 128;
  c->k_patch = (3 * patch * patch + 63) / 64 * 64;
  c->ln_eps = 1e-6f;
  return check_vit_cfg(c);
}

extern "C" int64_t dvt_vit_workspace_bytes(const DvtVitConfig* c, int batch) {
  if (check_vit_cfg(c) || batch <= 0) return -1;
  return vit_carve(c, batch, nullptr, nullptr);
}

extern "C" int dvt_vit_gemm_bias(const void* x, const void* w, const float* b, void* y, int m,
                                 int n, int k, void* stream) {
  if (!x || !w || !y) return DVT_E_BADARG;
  GemmBArgs a{};
  a.A = (const bf16_t*)x; a.W = (const bf16_t*)w; a.M = m; a.N = n; a.K = k;
  a.bias = b; a.out = (bf16_t*)y;
  a.lda = a.ldw = k;
  return launch_gemm<EPI_BIAS>(a, (hipStream_t)stream);
}

// fc1-type GEMM as the extractor launches it: y (bf16) = [GELU]( rstd * (x . w'^T - mean * cs) + b' ) with the LayerNorm folded
// (ln_stats [m] (mean, rstd), ln_cs [n]; both null: y = [GELU](x . w^T + b))
extern "C" int dvt_vit_gemm_lnfold(const void* x, const void* w, const float* b, void* y, int m, int n, int k,
                                   const void* ln_stats, const float* ln_cs, int gelu, void* stream) {
  if (!x || !w || !y || (ln_stats == nullptr) != (ln_cs == nullptr)) return DVT_E_BADARG;
  GemmBArgs a{};
  a.A = (const bf16_t*)x; a.W = (const bf16_t*)w; a.M = m; a.N = n; a.K = k;
  a.bias = b; a.out = (bf16_t*)y;
  a.lda = a.ldw = k;
  a.ln_stats = (const float2*)ln_stats;
  a.ln_cs = ln_cs;
  if (gelu) return launch_gemm<EPI_GELU>(a, (hipStream_t)stream);
  if (ln_stats != nullptr) return DVT_E_BADARG;  // the bias epilogue has no folded form
  return launch_gemm<EPI_BIAS>(a, (hipStream_t)stream);
}

extern "C" int dvt_vit_gemm_residual(const void* a_in, const void* w, const float* b, const float* gamma,
                                     float* x, int m, int n, int k, void* stream) {
  if (!a_in || !w || !gamma || !x) return DVT_E_BADARG;
  GemmBArgs a{};
  a.A = (const bf16_t*)a_in; a.W = (const bf16_t*)w; a.M = m; a.N = n; a.K = k;
  a.bias = b; a.x = x; a.gamma = gamma;
  return launch_gemm<EPI_RESID>(a, (hipStream_t)stream);
}

extern "C" int dvt_vit_gemm_f32out(const void* a_in, const void* w, const float* b, float* y, int m, int n, int k,
                                   void* stream) {
  if (!a_in || !w || !y) return DVT_E_BADARG;
  GemmBArgs a{};
  a.A = (const bf16_t*)a_in; a.W = (const bf16_t*)w; a.M = m; a.N = n; a.K = k;
  a.bias = b; a.x = y;
  return launch_gemm<EPI_F32>(a, (hipStream_t)stream);
}

// scratch of the bf16x3 attention (elements of bf16): q|k hi [m, 2 dim], q|k lo [m, 2 dim], V^T hi and lo
// [batch + 1, heads, 64, s_pad] each; m = batch * s_pad rounded up to whole 256-row GEMM tiles (the fused qkv epilogue
// writes those phantom rows, their V^T lands in image `batch`)
struct X3Scratch {
  long long m, ql, vh, vl, total;
};
static X3Scratch x3_scratch(int batch, int heads, int s_pad) {
  X3Scratch r;
  const long long dim = (long long)heads * 64;
  r.m = ((long long)batch * s_pad + 255) / 256 * 256;
  r.ql = r.m * 2 * dim;
  r.vh = 2 * r.ql;
  r.vl = r.vh + (long long)(batch + 1) * s_pad * dim;
  r.total = r.vl + (long long)(batch + 1) * s_pad * dim;
  return r;
}
// fc1 of the bf16x3 mode: out3 [m, 3 n] = split(GELU(a . w^T + b)) -- the next GEMM's A operand, no fp32 round trip
extern "C" int dvt_vit_gemm_gelu_x3(const void* a_in, const void* w, const float* b, void* out3, int m, int n, int k,
                                    void* stream) {
  if (!a_in || !w || !out3) return DVT_E_BADARG;
  GemmBArgs a{};
  a.A = (const bf16_t*)a_in; a.W = (const bf16_t*)w; a.M = m; a.N = n; a.K = k;
  a.bias = b; a.out = (bf16_t*)out3;
  return launch_gemm<EPI_GELU_X3>(a, (hipStream_t)stream);
}

// qkv of the bf16x3 mode: q | k as (hi, lo) [m, 2 dim] and V^T as (hi, lo) [batch, heads, 64, s_pad] in `scratch`
// (dvt_vit_attention_x3_scratch_bytes: the layout dvt_vit_attention_x3_presplit reads)
extern "C" int dvt_vit_gemm_qkv_x3(const void* a_in, const void* w, const float* b, void* scratch, int m, int dim, int heads,
                                   int s_pad, int batch, int k, void* stream) {
  if (!a_in || !w || !scratch || batch <= 0 || dim != heads * 64) return DVT_E_BADARG;
  const X3Scratch L = x3_scratch(batch, heads, s_pad);
  if (m != L.m) return DVT_E_BADARG;  // whole 256-row tiles over batch * s_pad rows, exactly
  GemmBArgs a{};
  a.A = (const bf16_t*)a_in; a.W = (const bf16_t*)w; a.M = m; a.N = 3 * dim; a.K = k;
  a.bias = b;
  a.out = (bf16_t*)scratch;
  a.out_lo = a.out + L.ql;
  a.vt = a.out + L.vh;
  a.vt_lo = a.out + L.vl;
  a.dim = dim; a.heads = heads; a.s_pad = s_pad;
  return launch_gemm<EPI_QKV_X3>(a, (hipStream_t)stream);
}

extern "C" int dvt_vit_layernorm(const float* x, const float* w, const float* b, void* y, int rows,
                                 int dim, float eps, void* stream) {
  if (!x || !w || !b || !y || rows < 0 || dim <= 0 || dim % 4 || dim > 1024) return DVT_E_BADARG;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(layernorm_kernel<false>, dim3(dvt_cdiv(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, x, w, b, (bf16_t*)y, (float*)nullptr, rows, dim, eps, 0, 0, 0);
  DVT_CHECK_LAUNCH();
  return 0;
}

// q pre-scaled by log2(e) / 8 (see attention_v2_body, VAR bit 32): the kernel dvt_vit_forward launches since round 6
extern "C" int dvt_vit_attention_log2q(const void* qk, const void* vt, void* out, int batch, int heads,
                                       int s_pad, int n_valid, void* stream) {
  if (!qk || !vt || !out || batch <= 0 || heads <= 0 || s_pad % 16 || n_valid <= 0 || n_valid > s_pad)
    return DVT_E_BADARG;
  DvtProbeScope probe(DVT_PROBE_VIT_ATTN, (hipStream_t)stream,
                      4.0 * (double)n_valid * n_valid * 64.0 * heads * batch);
  const dim3 grid(((s_pad + ATT_Q - 1) / ATT_Q) * heads * batch);
#ifdef DVT_LAB
  if (g_vit_attn_l2_mask != ATT_L2_VAR) {  // ablation builds / the other P.V order
#define A2L_VAR(n)                                                                                                    \
  if (g_vit_attn_l2_mask == (n)) {                                                                                    \
    hipLaunchKernelGGL(attention_kernel_l2<(n)>, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)qk,          \
                       (const bf16_t*)vt, (bf16_t*)out, heads, s_pad, n_valid);                                       \
    DVT_CHECK_LAUNCH();                                                                                               \
    return 0;                                                                                                         \
  }
    A2L_VAR(ATT_L2_VAR ^ 2) A2L_VAR(ATT_L2_VAR ^ 128) A2L_VAR(ATT_L2_VAR ^ 256) A2L_VAR(ATT_L2_VAR ^ 384) A2L_VAR(ATT_L2_VAR ^ 512)
    A2L_VAR(ATT_L2_VAR ^ 514) A2L_VAR(ATT_L2_VAR ^ 1024) A2L_VAR(ATT_L2_VAR ^ 2048) A2L_VAR(ATT_L2_VAR ^ 4096) A2L_VAR(ATT_L2_VAR ^ 8192) A2L_VAR(ATT_L2_VAR ^ 16384) A2L_VAR(ATT_L2_VAR ^ 32768) A2L_VAR(ATT_L2_VAR ^ (32768 + 65536))
#undef A2L_VAR
    return DVT_E_BADARG;
  }
#endif
  hipLaunchKernelGGL(attention_kernel_l2<ATT_L2_VAR>, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)qk,
                     (const bf16_t*)vt, (bf16_t*)out, heads, s_pad, n_valid);
  DVT_CHECK_LAUNCH();
  return 0;
}

extern "C" int dvt_vit_attention(const void* qk, const void* vt, void* out, int batch, int heads,
                                 int s_pad, int n_valid, void* stream) {
  if (!qk || !vt || !out || batch <= 0 || heads <= 0 || s_pad % 16 || n_valid <= 0 || n_valid > s_pad)
    return DVT_E_BADARG;
  DvtProbeScope probe(DVT_PROBE_VIT_ATTN, (hipStream_t)stream,
                      4.0 * (double)n_valid * n_valid * 64.0 * heads * batch);
  const dim3 grid(((s_pad + ATT_Q - 1) / ATT_Q) * heads * batch);
#ifdef DVT_LAB
  if (g_vit_attn_variant != 2) {
    if (s_pad % ATT_Q) return DVT_E_BADARG;  // (the round-2 loop has no query-block tail)
    hipLaunchKernelGGL(attention_kernel, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)qk, (const bf16_t*)vt,
                       (bf16_t*)out, heads, s_pad, n_valid);
    DVT_CHECK_LAUNCH();
    return 0;
  }
  bool done = false;
#define A2_VAR(n)                                                                                                     \
  if (g_vit_attn_mask == n) {                                                                                         \
    hipLaunchKernelGGL(attention_kernel_v2<n>, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)qk,            \
                       (const bf16_t*)vt, (bf16_t*)out, heads, s_pad, n_valid);                                       \
    done = true;                                                                                                      \
  }
  A2_VAR(0) A2_VAR(1) A2_VAR(2) A2_VAR(4) A2_VAR(8) A2_VAR(16) A2_VAR(64) A2_VAR(3) A2_VAR(15) A2_VAR(31) A2_VAR(79)
#undef A2_VAR
  if (!done) return DVT_E_BADARG;
#else
  hipLaunchKernelGGL(attention_kernel_v2<15>, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)qk,
                     (const bf16_t*)vt, (bf16_t*)out, heads, s_pad, n_valid);
#endif
  DVT_CHECK_LAUNCH();
  return 0;
}

// ---- bf16x3 attention for the fp32 extractor's `--fp32_matmul high` mode ---------------------------------------
namespace {
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack2(a, b);
  lo = pack2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
// q | k columns of the fp32 qkv rows -> (hi, lo) bf16 arrays [T, 2 dim]
__global__ __launch_bounds__(256) void qk_split_kernel(const float* __restrict__ qkv, bf16_t* __restrict__ hi,
                                                       bf16_t* __restrict__ lo, long long nq, int dq2, int ld3) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
    const long long t = q / dq2;
    const int c = (int)(q - t * dq2) * 4;
    const float4 v = *reinterpret_cast<const float4*>(qkv + t * ld3 + c);
    uint2 h, l;
    split_pair(v.x, v.y, h.x, l.x);
    split_pair(v.z, v.w, h.y, l.y);
    *reinterpret_cast<uint2*>(hi + t * (dq2 * 4) + c) = h;
    *reinterpret_cast<uint2*>(lo + t * (dq2 * 4) + c) = l;
  }
}
// v columns -> transposed (hi, lo) arrays vt[b][h][d][s]; one workgroup = 64 tokens of one (image, head)
__global__ __launch_bounds__(256) void v_split_transpose_kernel(const float* __restrict__ qkv, bf16_t* __restrict__ vth,
                                                                bf16_t* __restrict__ vtl, int heads, int s_pad) {
  __shared__ float tile[64 * 65];
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int dim = heads * 64, ld3 = 3 * dim;
  const float* src = qkv + ((size_t)b * s_pad + t0) * ld3 + 2 * dim + h * 64;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int f = tid + 256 * it, i = f >> 4, dq = f & 15;
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)i * ld3 + dq * 4);
    float* d = tile + i * 65 + dq * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const int d = tid >> 2, part = tid & 3;
  uint32_t hw[8], lw[8];
#pragma unroll
  for (int n = 0; n < 8; ++n)
    split_pair(tile[(part * 16 + 2 * n) * 65 + d], tile[(part * 16 + 2 * n + 1) * 65 + d], hw[n], lw[n]);
  const size_t o = ((size_t)(b * heads + h) * 64 + d) * s_pad + t0 + part * 16;
  *reinterpret_cast<uint4*>(vth + o) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  *reinterpret_cast<uint4*>(vth + o + 8) = make_uint4(hw[4], hw[5], hw[6], hw[7]);
  *reinterpret_cast<uint4*>(vtl + o) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  *reinterpret_cast<uint4*>(vtl + o + 8) = make_uint4(lw[4], lw[5], lw[6], lw[7]);
}
}  // namespace

extern "C" int dvt_vit_attention_x3_presplit(const void* scratch, void* out, int batch, int heads, int s_pad, int n_valid,
                                             int split_out, void* stream);
extern "C" int64_t dvt_vit_attention_x3_scratch_bytes(int batch, int heads, int s_pad) {
  if (batch <= 0 || heads <= 0 || s_pad <= 0) return -1;
  return x3_scratch(batch, heads, s_pad).total * 2;
}

extern "C" int dvt_vit_attention_x3(const float* qkv, float* out, void* scratch, int batch, int heads, int s_pad,
                                    int n_valid, void* stream) {
  if (!qkv || !out || !scratch || batch <= 0 || heads <= 0 || s_pad % ATT_Q || n_valid <= 0 || n_valid > s_pad)
    return DVT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int dim = heads * 64;
  const long long T = (long long)batch * s_pad;
  const X3Scratch L = x3_scratch(batch, heads, s_pad);
  bf16_t* qh = (bf16_t*)scratch;
  bf16_t* ql = qh + L.ql;
  bf16_t* vh = qh + L.vh;
  bf16_t* vl = qh + L.vl;
  const long long nq = T * (2 * dim / 4);
  hipLaunchKernelGGL(qk_split_kernel, dim3((unsigned)(nq / 256 + 1 < 4096 ? nq / 256 + 1 : 4096)), dim3(256), 0, s, qkv, qh,
                     ql, nq, 2 * dim / 4, 3 * dim);
  DVT_CHECK_LAUNCH();
  hipLaunchKernelGGL(v_split_transpose_kernel, dim3(s_pad / 64, heads, batch), dim3(256), 0, s, qkv, vh, vl, heads, s_pad);
  DVT_CHECK_LAUNCH();
  return dvt_vit_attention_x3_presplit(scratch, out, batch, heads, s_pad, n_valid, 0, stream);
}

extern "C" int dvt_vit_attention_x3_presplit(const void* scratch, void* out, int batch, int heads, int s_pad, int n_valid,
                                             int split_out, void* stream) {
  if (!out || !scratch || batch <= 0 || heads <= 0 || s_pad % ATT_Q || n_valid <= 0 || n_valid > s_pad)
    return DVT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const X3Scratch L = x3_scratch(batch, heads, s_pad);
  const bf16_t* qh = (const bf16_t*)scratch;
  const bf16_t* ql = qh + L.ql;
  const bf16_t* vh = qh + L.vh;
  const bf16_t* vl = qh + L.vl;
  DvtProbeScope probe(DVT_PROBE_VIT_ATTN, s, 3.0 * 4.0 * (double)n_valid * n_valid * 64.0 * heads * batch);
  hipLaunchKernelGGL(attention_kernel_v2_x3, dim3((s_pad / ATT_Q) * heads * batch), dim3(512), 0, s,
                     (const bf16_t*)qh, (const bf16_t*)vh, (bf16_t*)out, heads, s_pad, n_valid, (const bf16_t*)ql,
                     (const bf16_t*)vl, split_out);
  DVT_CHECK_LAUNCH();
  return 0;
}

extern "C" int dvt_vit_forward(const DvtVitConfig* c, const DvtVitWeights* w, const float* img,
                               float* feat, int batch, int n_blocks, void* workspace,
                               void* stream) {
  int rc = check_vit_cfg(c);
  if (rc) return rc;
  if (!w || !img || !feat || !workspace || batch <= 0 || n_blocks < 0 || n_blocks > c->depth)
    return DVT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  VitWork k;
  vit_carve(c, batch, (char*)workspace, &k);
  const int T = (batch * c->s_pad + 255) / 256 * 256, D = c->dim;  // incl. phantom rows (vit_carve)
  // algorithmic GEMM rows: the real tokens, not the rows padded to a multiple of 128
  const double rows = (double)batch * c->n_tokens;

#define DVT_TRY(x)         \
  do {                     \
    int rc__ = (x);        \
    if (rc__) return rc__; \
  } while (0)

  // patch embedding: im2col -> GEMM with the (+bias, +pos_embed, cls) epilogue
  if (c->patch == 14 && c->stride % 2 == 0 && c->img_w % 2 == 0 && c->k_patch % 2 == 0)
    hipLaunchKernelGGL(im2col_pairs_kernel<14>, dim3(T), dim3(256), 0, s, img, k.col, *c, batch * c->s_pad);
  else if (c->patch == 16 && c->stride % 2 == 0 && c->img_w % 2 == 0 && c->k_patch % 2 == 0)
    hipLaunchKernelGGL(im2col_pairs_kernel<16>, dim3(T), dim3(256), 0, s, img, k.col, *c, batch * c->s_pad);
  else
    hipLaunchKernelGGL(im2col_kernel, dim3(T), dim3(256), 0, s, img, k.col, *c, batch * c->s_pad);
  DVT_CHECK_LAUNCH();
  {
    GemmBArgs a{};
    a.A = k.col; a.W = (const bf16_t*)w->patch_w; a.M = T; a.N = D; a.K = c->k_patch;
    a.bias = w->patch_b; a.x = k.x; a.pos = w->pos_embed; a.cls = w->cls_token;
    a.s_pad = c->s_pad; a.n_tokens = c->n_tokens; a.dim = D; a.heads = c->heads;
    a.n_prefix = c->n_prefix; a.pos_has_cls = c->pos_has_cls;
    a.work = 2.0 * (double)batch * c->grid_h * c->grid_w * D * (3.0 * c->patch * c->patch);
    DVT_TRY(launch_gemm<EPI_EMBED>(a, s));
  }
  // LayerNorm folded into the GEMMs (ln_fold): needs the folded weights, the 8-phase kernel on every GEMM of the
  // block (whole 256-row / 256-column tiles) and is switched by dvt_vit_tune; otherwise the LayerNorm kernels run.
  bool fuse_ln = g_vit_fuse_ln && g_vit_gemm_variant >= 4 && T % 256 == 0 && D % 256 == 0 && c->mlp_dim % 256 == 0 &&
                 n_blocks > 0;
  for (int l = 0; l < n_blocks; ++l) {
    const DvtVitBlockWeights& bw = w->blocks[l];
    fuse_ln = fuse_ln && bw.qkv_wf && bw.qkv_cs && bw.qkv_bf && bw.fc1_wf && bw.fc1_cs && bw.fc1_bf;
  }
  const int n_part = D / 64;
  auto finalize_stats = [&]() {
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3(dvt_cdiv(T, 256)), dim3(256), 0, s, (const float2*)k.st_part,
                       n_part, T, 1.0f / (float)D, c->ln_eps, k.stats);
  };
  if (fuse_ln) {
    hipLaunchKernelGGL(ln_cast_stats_kernel, dim3(dvt_cdiv(T, 4)), dim3(256), 0, s, (const float*)k.x, k.xb, k.stats, T, D,
                       c->ln_eps);
    DVT_CHECK_LAUNCH();
  }
  bool log2q = g_vit_attn_log2q != 0;
#ifdef DVT_LAB
  log2q = log2q && g_vit_attn_variant == 2 && g_vit_attn_mask == 15;  // (the superseded kernels and schedule masks take q as it is)
#endif
  for (int l = 0; l < n_blocks; ++l) {
    const DvtVitBlockWeights& bw = w->blocks[l];
    if (!fuse_ln) DVT_TRY(dvt_vit_layernorm(k.x, bw.norm1_w, bw.norm1_b, k.xn, T, D, c->ln_eps, s));
    {
      GemmBArgs a{};
      a.A = fuse_ln ? k.xb : k.xn; a.W = (const bf16_t*)(fuse_ln ? bw.qkv_wf : bw.qkv_w); a.M = T; a.N = 3 * D; a.K = D;
      a.bias = fuse_ln ? bw.qkv_bf : bw.qkv_b; a.out = k.qk; a.vt = k.vt;
      if (fuse_ln) { a.ln_stats = k.stats; a.ln_cs = bw.qkv_cs; }
      a.dim = D; a.heads = c->heads; a.s_pad = c->s_pad; a.n_tokens = c->n_tokens;
      a.work = 2.0 * rows * 3.0 * D * D;
      a.q_scale = log2q ? ATT_Q_PRESCALE : 0.f;
      DVT_TRY(launch_gemm<EPI_QKV>(a, s));
    }
    if (log2q) DVT_TRY(dvt_vit_attention_log2q(k.qk, k.vt, k.xn, batch, c->heads, c->s_pad, c->n_tokens, s));
    else DVT_TRY(dvt_vit_attention(k.qk, k.vt, k.xn, batch, c->heads, c->s_pad, c->n_tokens, s));
    {
      GemmBArgs a{};
      a.A = k.xn; a.W = (const bf16_t*)bw.proj_w; a.M = T; a.N = D; a.K = D;
      a.bias = bw.proj_b; a.x = k.x; a.gamma = bw.ls1;
      if (fuse_ln) { a.xb = k.xb; a.st_part = k.st_part; }
      a.work = 2.0 * rows * D * D;
      DVT_TRY(launch_gemm<EPI_RESID>(a, s));
    }
    if (fuse_ln) {
      finalize_stats();
      DVT_CHECK_LAUNCH();
    } else {
      DVT_TRY(dvt_vit_layernorm(k.x, bw.norm2_w, bw.norm2_b, k.xn, T, D, c->ln_eps, s));
    }
    {
      GemmBArgs a{};
      a.A = fuse_ln ? k.xb : k.xn; a.W = (const bf16_t*)(fuse_ln ? bw.fc1_wf : bw.fc1_w); a.M = T; a.N = c->mlp_dim; a.K = D;
      a.bias = fuse_ln ? bw.fc1_bf : bw.fc1_b; a.out = k.hid;
      if (fuse_ln) { a.ln_stats = k.stats; a.ln_cs = bw.fc1_cs; }
      a.work = 2.0 * rows * (double)c->mlp_dim * D;
      DVT_TRY(launch_gemm<EPI_GELU>(a, s));
    }
    {
      GemmBArgs a{};
      a.A = k.hid; a.W = (const bf16_t*)bw.fc2_w; a.M = T; a.N = D; a.K = c->mlp_dim;
      a.bias = bw.fc2_b; a.x = k.x; a.gamma = bw.ls2;
      if (fuse_ln && l + 1 < n_blocks) { a.xb = k.xb; a.st_part = k.st_part; }
      a.work = 2.0 * rows * (double)c->mlp_dim * D;
      DVT_TRY(launch_gemm<EPI_RESID>(a, s));
    }
    if (fuse_ln && l + 1 < n_blocks) {
      finalize_stats();
      DVT_CHECK_LAUNCH();
    }
  }
#undef DVT_TRY
  // final LayerNorm, drop cls/pad rows, NHWC fp32 straight into the feature store
  const int out_rows = batch * (c->n_tokens - c->n_prefix);
  hipLaunchKernelGGL(layernorm_kernel<true>, dim3(dvt_cdiv(out_rows, 4)), dim3(256), 0, s, k.x,
                     w->norm_w, w->norm_b, (bf16_t*)nullptr, feat, out_rows, D, c->ln_eps, c->s_pad,
                     c->n_tokens, c->n_prefix);
  DVT_CHECK_LAUNCH();
  return 0;
}
