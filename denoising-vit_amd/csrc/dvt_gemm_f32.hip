// Exact-fp32 linear layers on the f32-input matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD, bitwise an fmaf chain).
//
// Replaces the cuBLAS SGEMMs behind `nn.Sequential(Linear, ReLU, Linear)` of the field MLP
// (dvt/models/neural_feature_field.py:40-44, :49) and of the residual predictor
// (dvt/models/offline_denoiser.py:40-46, :107), forward and backward.
//
// One kernel template covers the three contractions of a linear layer; the operands differ
// only in which index is contiguous in memory:
//   forward  y = x . w^T      A = x  [m][k] k-contiguous   B = w  [n][k] k-contiguous
//   dgrad    dx = dy . w      A = dy [m][n] k-contiguous   B = w  [n][k] row-contiguous
//   wgrad    dw = dy^T . x    A = dy [b][n] row-contiguous B = x  [b][k] row-contiguous
// Tile 64x64x64 per 256-thread workgroup (4 waves, each one 32x32 accumulator = 16 VGPRs);
// tiles are staged through LDS so that global reads are 16-B coalesced and the MFMA
// fragment reads (lane l: A[l&31][l>>5], B[l>>5][l&31]) are conflict-free:
//   k-contiguous operand  -> LDS [row][k] with leading dimension BK+1
//   row-contiguous operand-> LDS [k][row] with leading dimension 64
// The small batch (2048 rows) yields few tiles, so wgrad splits the batch reduction over
// blockIdx.z and accumulates with fp32 atomics into a gradient buffer that the fused Adam
// kernel clears (dvt_adam.hip).
#include "dvt_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

// BK = 64: 32 MFMAs (2048 cycles) per staged tile -- with BK = 32 every k-tile cost ~1 us, twice
// its MFMA time, because the single prefetched tile's L2 latency was only half hidden.
constexpr int BK = 64;
constexpr int LDK = BK + 4;  // [row][k] layout: rows stay 16-B aligned for b128 accesses; a 16-lane
                            // group reading 16 B at a 272-B row stride covers all 16 bank slots

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  const float* bias;  // [N] added per output column
  const float* mask;  // [M, ldmask]: output multiplied by (mask > 0)
  int ldmask;
  float* colsum;  // [M]: += sum_k A(m,k), written by the n-tile-0 blocks (A row-contiguous only)
  int relu;
  int kchunk;  // K range per blockIdx.z (multiple of BK)
  int atomic;  // accumulate into C with atomics
  // gemm_f32_big_kernel only (the fp32 extractor's fused epilogues): 1 = exact-erf GELU of (acc + bias); 2 = residual,
  // C[m][n] += gamma[n] * (acc + bias[n]) (timm Block: x = x + ls(f(norm(x))))
  int epi;
  const float* gamma;
  // gemm_f32_body, set by the batched small-k kernel only: C = oscale * smul[m][n] * (acc - rowsub[m]) -- the softmax backward
  // dS = scale P (.) (dP - rowsum(dP (.) P)) as the epilogue of dP = dO V^T, with rowsum(dP (.) P) = dO . O handed in
  const float* smul;
  const float* rowsub;
  float oscale;
};

// Operand tile of R rows x BK: R*BK/4 float4, spread over NT threads.  BK is a template
// parameter of the register-staged kernel: 64 by default, 16 (10 KB of LDS per workgroup) when
// the fit shares the GPU with the ViT extractor, whose 136-144 KB workgroups leave only 16-24 KB
// of a CU's LDS free (co-residency instead of waiting for CUs to drain).
template <bool KCONTIG, int R, int NT, int BK>
struct Tile {
  static constexpr int LDK = BK + 4;
  static constexpr int NF4 = R * BK / 4;
  static constexpr int ITERS = (NF4 + NT - 1) / NT;
  static constexpr int LDS_FLOATS = KCONTIG ? R * LDK : BK * R;

  __device__ static __forceinline__ void load(const float* __restrict__ X, int ld, int r0, int Rmax,
                                              int k0, int kend, int tid, float4 regs[ITERS]) {
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int f = tid + NT * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (NF4 % NT == 0 || f < NF4) {
        if (KCONTIG) {
          const int r = f / (BK / 4), kq = f % (BK / 4);
          const int gr = r0 + r, gk = k0 + 4 * kq;
          if (gr < Rmax && gk < kend) v = *reinterpret_cast<const float4*>(X + (size_t)gr * ld + gk);
        } else {
          const int k = f / (R / 4), rq = f % (R / 4);
          const int gk = k0 + k, gr = r0 + 4 * rq;
          if (gk < kend && gr < Rmax) v = *reinterpret_cast<const float4*>(X + (size_t)gk * ld + gr);
        }
      }
      regs[i] = v;
    }
  }

  __device__ static __forceinline__ void store(float* __restrict__ S, int tid,
                                               const float4 regs[ITERS]) {
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int f = tid + NT * i;
      if (NF4 % NT == 0 || f < NF4) {
        if (KCONTIG) {
          const int r = f / (BK / 4), kq = f % (BK / 4);
          *reinterpret_cast<float4*>(S + r * LDK + 4 * kq) = regs[i];
        } else {
          const int k = f / (R / 4), rq = f % (R / 4);
          *reinterpret_cast<float4*>(S + k * R + 4 * rq) = regs[i];
        }
      }
    }
  }

  // All BK/2 = 32 fragment values of one lane for this tile.  MFMA step kk uses k index
  // kh*32 + kk (kh = lane >> 5): any bijection of k is valid as long as A and B agree, and this
  // one makes a lane's values CONTIGUOUS in the k-contiguous layout (8 x ds_read_b128).
  __device__ static __forceinline__ void frags(const float* __restrict__ S, int row, int kh,
                                               float (&f)[BK / 2]) {
    if (KCONTIG) {
      const float4* p = reinterpret_cast<const float4*>(S + row * LDK + kh * (BK / 2));
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        const float4 v = p[q];
        f[4 * q + 0] = v.x;
        f[4 * q + 1] = v.y;
        f[4 * q + 2] = v.z;
        f[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) f[kk] = S[(kh * (BK / 2) + kk) * R + row];
    }
  }
};

// WM x WN waves, each owning one 32x32 accumulator: tile (32*WM) x (32*WN).
// Body shared by the single-problem kernel and the grouped kernel (explicit block coordinates).
template <bool A_KC, bool B_KC, int WM, int WN, int BK>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs& p, int bx, int by, int bz,
                                              float* __restrict__ As, float* __restrict__ Bs) {
  constexpr int NT = 64 * WM * WN, BM = 32 * WM, BN = 32 * WN;
  using TA = Tile<A_KC, BM, NT, BK>;
  using TB = Tile<B_KC, BN, NT, BK>;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = bx * BN, m0 = by * BM;
  const int kbeg = bz * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  if (kbeg >= kend) return;

  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float csum = 0.f;
  const bool do_colsum = (!A_KC) && p.colsum != nullptr && bx == 0;

  float4 ra[TA::ITERS], rb[TB::ITERS];
  TA::load(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra);
  TB::load(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    TA::store(As, tid, ra);
    TB::store(Bs, tid, rb);
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch the next tile while the MFMAs run
      TA::load(p.A, p.lda, m0, p.M, k0 + BK, kend, tid, ra);
      TB::load(p.B, p.ldb, n0, p.N, k0 + BK, kend, tid, rb);
    }
    // Fragments of the whole tile first, then 32 MFMAs back to back: the compiler otherwise
    // re-used one register pair and waited lgkmcnt(0) on a fresh ds_read every 2 dependent
    // MFMAs (each k-tile cost ~2x its MFMA time).
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    float fa[BK / 2], fb[BK / 2];
    TA::frags(As, ar, kh, fa);
    TB::frags(Bs, bc, kh, fb);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc, 0, 0, 0);
    if (do_colsum && tid < BM) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) csum += As[k * BM + tid];
    }
    __syncthreads();
  }

  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int gn = n0 + wn * 32 + (lane & 31);
  const float bias = (p.bias != nullptr && gn < p.N) ? p.bias[gn] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int gm = m0 + wm * 32 + row;
    if (gm < p.M && gn < p.N) {
      float v = acc[r] + bias;
      if (p.relu) v = fmaxf(v, 0.f);
      if (p.mask != nullptr) v = p.mask[(size_t)gm * p.ldmask + gn] > 0.f ? v : 0.f;
      if (p.smul != nullptr) v = p.oscale * p.smul[(size_t)gm * p.ldc + gn] * (v - p.rowsub[gm]);
      float* c = p.C + (size_t)gm * p.ldc + gn;
      if (p.atomic)
        atomic_add_f32(c, v);
      else
        *c = v;
    }
  }
  if (do_colsum && tid < BM && m0 + tid < p.M) atomic_add_f32(p.colsum + m0 + tid, csum);
}

template <bool A_KC, bool B_KC, int WM, int WN, int BK>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_kernel(GemmArgs p) {
  constexpr int NT = 64 * WM * WN;
  __shared__ __attribute__((aligned(16))) float As[Tile<A_KC, 32 * WM, NT, BK>::LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[Tile<B_KC, 32 * WN, NT, BK>::LDS_FLOATS];
  gemm_f32_body<A_KC, B_KC, WM, WN, BK>(p, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// ---- grouped launch: up to 4 independent GEMMs in ONE grid -------------------------------------
// At M = 2048 every GEMM of the fit fills only 96-430 workgroups and carries ~4.5 us of fixed
// dependent-latency cost; kernels on different HIP streams did not overlap (measured: wgrad2 ||
// dgrad2 on two streams = 59.8 us, the same as back to back).  Independent GEMMs of one step
// (wgrad + dgrad of a layer, field branch + residual branch) are therefore packed into one
// launch: a workgroup finds its problem from blockIdx.x, the operand layouts become a run-time
// switch over the three instantiations of the same body.
constexpr int MULTI_MAX = 16;  // 4 problems of one fit x DVT_FIT_BATCH_MAX fits
struct MultiArgs {
  int n;
  int blk0[MULTI_MAX + 1];  // first block of each problem
  int gx[MULTI_MAX], gy[MULTI_MAX];
  int layout[MULTI_MAX];  // 0: A k-contig, B k-contig; 1: A k-contig, B row-contig; 2: both row-contig
  GemmArgs g[MULTI_MAX];
};

template <int BK>
__global__ __launch_bounds__(256) void gemm_f32_multi_kernel(MultiArgs m) {
  __shared__ __attribute__((aligned(16))) float As[64 * (BK + 4)];  // >= BK * 64
  __shared__ __attribute__((aligned(16))) float Bs[64 * (BK + 4)];
  int i = 0;
#pragma unroll
  for (int j = 1; j < MULTI_MAX; ++j)
    if (j < m.n && (int)blockIdx.x >= m.blk0[j]) i = j;
  const int local = (int)blockIdx.x - m.blk0[i];
  const int bx = local % m.gx[i], by = (local / m.gx[i]) % m.gy[i], bz = local / (m.gx[i] * m.gy[i]);
  switch (m.layout[i]) {
    case 0: gemm_f32_body<true, true, 2, 2, BK>(m.g[i], bx, by, bz, As, Bs); break;
    case 1: gemm_f32_body<true, false, 2, 2, BK>(m.g[i], bx, by, bz, As, Bs); break;
    default: gemm_f32_body<false, false, 2, 2, BK>(m.g[i], bx, by, bz, As, Bs); break;
  }
}

// ==========================================================================================
// bf16-operand variant of the grouped body (DvtFitConfig.mlp_bf16, the reference's
// `--dtype bfloat16` autocast mode: nn.Linear runs on bf16 casts of fp32 master weights and
// activations; main_img_denoising.py:78).  Operands stay fp32 in HBM and are rounded to bf16
// (RNE, v_cvt_pk_bf16_f32) while they are staged into LDS; accumulation, bias, ReLU, masks and
// every output stay fp32 -- i.e. at least the reference's precision (autocast additionally rounds
// each layer output to bf16).  The fp32-operand body above is MFMA-rate bound: one
// v_mfma_f32_32x32x2_f32 (64 cycles) per 2 k per wave, a K = 768 contraction is a dependent chain
// of 24.6 k cycles, ~10 us, whatever the tile/BK/staging (measured).  v_mfma_f32_32x32x16_bf16
// covers 16 k in 32 cycles: the same chain is 1.5 k cycles.
// 64 x 64 tile, 2 x 2 waves, BK = 32 (64 selectable).  LDS rows are k-contiguous bf16, pitch BK + 8
// (conflict-free ds_read_b128 of a lane's 8 k); row-contiguous sources (k slow) are transposed on
// the way in: a thread loads (k, k+1) for 4 rows and writes four packed bf16x2 words.
// ==========================================================================================
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 hwbf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  const hwbf16x2_t v = __builtin_convertvector((f32x2_t){a, b}, hwbf16x2_t);
  return __builtin_bit_cast(uint32_t, v);
}


template <bool KCONTIG, int HBK>
struct TileH {
  static constexpr int HP = HBK + 8;        // bf16 elements per LDS row
  static constexpr int ITERS = HBK / 16;    // float4 per thread per operand tile (256 threads)
  __device__ static __forceinline__ void load(const float* __restrict__ X, int ld, int r0, int Rmax,
                                              int k0, int kend, int tid, float4 regs[ITERS]) {
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KCONTIG) {
        const int f = tid + 256 * i, r = f / (HBK / 4), kq = f % (HBK / 4);
        const int gr = r0 + r, gk = k0 + 4 * kq;
        if (gr < Rmax && gk < kend) v = *reinterpret_cast<const float4*>(X + (size_t)gr * ld + gk);
      } else {  // regs[2 * it + par] = rows 4*rq .. +3 at k = 2 * (kp + 16 * it) + par
        const int kp = (tid >> 4) + 16 * (i >> 1), rq = tid & 15;
        const int gk = k0 + 2 * kp + (i & 1), gr = r0 + 4 * rq;
        if (gk < kend && gr < Rmax) v = *reinterpret_cast<const float4*>(X + (size_t)gk * ld + gr);
      }
      regs[i] = v;
    }
  }
  __device__ static __forceinline__ void store(uint16_t* __restrict__ S, int tid, const float4 regs[ITERS]) {
    if (KCONTIG) {
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        const int f = tid + 256 * i, r = f / (HBK / 4), kq = f % (HBK / 4);
        uint2 w;
        w.x = pack_bf16x2(regs[i].x, regs[i].y);
        w.y = pack_bf16x2(regs[i].z, regs[i].w);
        *reinterpret_cast<uint2*>(S + r * HP + 4 * kq) = w;
      }
    } else {
      const int rq = tid & 15;
      uint32_t* S32 = reinterpret_cast<uint32_t*>(S);
#pragma unroll
      for (int it = 0; it < ITERS / 2; ++it) {
        const int kp = (tid >> 4) + 16 * it;
        const float4 lo = regs[2 * it], hi = regs[2 * it + 1];
        S32[((4 * rq + 0) * HP >> 1) + kp] = pack_bf16x2(lo.x, hi.x);
        S32[((4 * rq + 1) * HP >> 1) + kp] = pack_bf16x2(lo.y, hi.y);
        S32[((4 * rq + 2) * HP >> 1) + kp] = pack_bf16x2(lo.z, hi.z);
        S32[((4 * rq + 3) * HP >> 1) + kp] = pack_bf16x2(lo.w, hi.w);
      }
    }
  }
};

template <bool A_KC, bool B_KC, int HBK>
__device__ __forceinline__ void gemm_bf16op_body(const GemmArgs& p, int bx, int by, int bz,
                                                 uint16_t* __restrict__ As, uint16_t* __restrict__ Bs) {
  using TA = TileH<A_KC, HBK>;
  using TB = TileH<B_KC, HBK>;
  constexpr int HP = TA::HP, NR = TA::ITERS;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = bx * 64, m0 = by * 64;
  const int kbeg = bz * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  if (kbeg >= kend) return;

  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float csum = 0.f;
  const bool do_colsum = (!A_KC) && p.colsum != nullptr && bx == 0;

  // Cycle stamps of one workgroup (K = 384 .. 768): ~2000 cycles from a tile's global loads to its
  // MFMAs, i.e. with one tile in flight every k-step costs a full memory latency (0.85 us x 6-12
  // steps), and another ~2500 cycles of the epilogue went to the bias / mask loads.  Hence TWO
  // tiles in flight (register sets r0 / r1) and the epilogue operands requested up front.
  float4 ra0[NR], rb0[NR], ra1[NR], rb1[NR];
  TA::load(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra0);
  TB::load(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb0);
  TA::load(p.A, p.lda, m0, p.M, kbeg + HBK, kend, tid, ra1);  // past kend: zeros, no memory access
  TB::load(p.B, p.ldb, n0, p.N, kbeg + HBK, kend, tid, rb1);
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int gn = n0 + wn * 32 + (lane & 31);
  const float bias = (p.bias != nullptr && gn < p.N) ? p.bias[gn] : 0.f;
  float mk[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    mk[r] = (p.mask != nullptr && gm < p.M && gn < p.N) ? p.mask[(size_t)gm * p.ldmask + gn] : 1.f;
  }
  // fragment addresses: MFMA step s of a tile uses k = kh * (HBK / 2) + s * 8 + j (j = 0..7, kh = lane >> 5)
  const int kh = lane >> 5;
  const uint16_t* pa = As + (wm * 32 + (lane & 31)) * HP + kh * (HBK / 2);
  const uint16_t* pb = Bs + (wn * 32 + (lane & 31)) * HP + kh * (HBK / 2);
#define H_STEP(RA, RB, KNEXT)                                                                  \
  do {                                                                                         \
    TA::store(As, tid, RA);                                                                    \
    TB::store(Bs, tid, RB);                                                                    \
    __syncthreads();                                                                           \
    TA::load(p.A, p.lda, m0, p.M, (KNEXT), kend, tid, RA); /* two tiles ahead */               \
    TB::load(p.B, p.ldb, n0, p.N, (KNEXT), kend, tid, RB);                                     \
    bf16x8_t fa[HBK / 16], fb[HBK / 16];                                                       \
    _Pragma("unroll") for (int q = 0; q < HBK / 16; ++q) {                                     \
      fa[q] = *reinterpret_cast<const bf16x8_t*>(pa + 8 * q);                                  \
      fb[q] = *reinterpret_cast<const bf16x8_t*>(pb + 8 * q);                                  \
    }                                                                                          \
    _Pragma("unroll") for (int q = 0; q < HBK / 16; ++q)                                       \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q], fb[q], acc, 0, 0, 0);             \
    if (do_colsum && tid < 64) { /* bias gradient: column sums of the staged (bf16) dy tile */ \
      const uint32_t* row = reinterpret_cast<const uint32_t*>(As + tid * HP);                  \
      _Pragma("unroll") for (int k = 0; k < HBK / 2; ++k) {                                    \
        const uint32_t w = row[k];                                                             \
        csum += __uint_as_float(w << 16) + __uint_as_float(w & 0xffff0000u);                   \
      }                                                                                        \
    }                                                                                          \
    __syncthreads();                                                                           \
  } while (0)
  for (int k0 = kbeg; k0 < kend; k0 += 2 * HBK) {
    H_STEP(ra0, rb0, k0 + 2 * HBK);
    if (k0 + HBK < kend) H_STEP(ra1, rb1, k0 + 3 * HBK);
  }
#undef H_STEP

#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int gm = m0 + wm * 32 + row;
    if (gm < p.M && gn < p.N) {
      float v = acc[r] + bias;
      if (p.relu) v = fmaxf(v, 0.f);
      v = mk[r] > 0.f ? v : 0.f;
      float* c = p.C + (size_t)gm * p.ldc + gn;
      if (p.atomic)
        atomic_add_f32(c, v);
      else
        *c = v;
    }
  }
  if (do_colsum && tid < 64 && m0 + tid < p.M) atomic_add_f32(p.colsum + m0 + tid, csum);
}

template <int HBK>
__global__ __launch_bounds__(256) void gemm_bf16op_multi_kernel(MultiArgs m) {
  __shared__ __attribute__((aligned(16))) uint16_t As[64 * (HBK + 8)];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[64 * (HBK + 8)];
  int i = 0;
#pragma unroll
  for (int j = 1; j < MULTI_MAX; ++j)
    if (j < m.n && (int)blockIdx.x >= m.blk0[j]) i = j;
  const int local = (int)blockIdx.x - m.blk0[i];
  const int bx = local % m.gx[i], by = (local / m.gx[i]) % m.gy[i], bz = local / (m.gx[i] * m.gy[i]);
  switch (m.layout[i]) {
    case 0: gemm_bf16op_body<true, true, HBK>(m.g[i], bx, by, bz, As, Bs); break;
    case 1: gemm_bf16op_body<true, false, HBK>(m.g[i], bx, by, bz, As, Bs); break;
    default: gemm_bf16op_body<false, false, HBK>(m.g[i], bx, by, bz, As, Bs); break;
  }
}

// ==========================================================================================
// 3-stage LDS-DMA variant (used when K % 64 == 0 and row-contiguous operands are 64-aligned).
// PMC evidence for the register-staged kernel above at the fit's shapes: L2 hit rate 43 % (the
// operands were just written by other XCDs' kernels), waves 45-54 % in s_waitcnt, MFMA pipe
// ~13 % busy -> latency-bound with one tile in flight.  Here global_load_lds keeps TWO 32-KB
// stages in flight per workgroup behind the one being multiplied: one raw s_barrier per
// k-tile and a counted `s_waitcnt vmcnt(8)` (8 DMA instructions per thread per stage).
// LDS image is lane-linear; for k-contiguous operands the 16-B chunk index of a 256-B row is
// XOR-ed with (row & 15) through the SOURCE address, and the b128 fragment reads apply the
// same XOR (conflict-free: the 16 rows of a lane group hit 16 distinct slots).
// ==========================================================================================
constexpr int G3_STAGE_FLOATS = 2 * 64 * BK;  // A tile + B tile, 64 rows/cols x 64 k = 32 KB

__device__ __forceinline__ void glds16f(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <bool KCONTIG>
__device__ __forceinline__ void stage_f32(const float* __restrict__ X, int ld, int r0, int Rmax,
                                          int k0, float* lds, int wave, int lane) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = it * 256 + wave * 64 + lane;  // 16-B slot in the 16-KB tile
    const float* src;
    if (KCONTIG) {
      const int row = s >> 4, cp = s & 15, c = cp ^ (row & 15);
      src = X + (size_t)min(r0 + row, Rmax - 1) * ld + k0 + c * 4;  // rows >= Rmax: clamped, masked at the store
    } else {
      const int k = s >> 4, cq = s & 15;
      src = X + (size_t)(k0 + k) * ld + r0 + cq * 4;
    }
    glds16f(src, lds + (it * 256 + wave * 64) * 4);
  }
}

template <bool KCONTIG>
__device__ __forceinline__ void frags_f32(const float* __restrict__ S, int row, int kh,
                                          float (&f)[BK / 2]) {
  if (KCONTIG) {
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      const int chunk = (kh * (BK / 8) + q) ^ (row & 15);
      const float4 v = *reinterpret_cast<const float4*>(S + row * BK + chunk * 4);
      f[4 * q + 0] = v.x;
      f[4 * q + 1] = v.y;
      f[4 * q + 2] = v.z;
      f[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) f[kk] = S[(kh * (BK / 2) + kk) * 64 + row];
  }
}

// NS = 3 stages (96 KB, one workgroup per CU, loads two k-tiles ahead) or 2 stages (64 KB, TWO workgroups per CU,
// loads one k-tile ahead): with one workgroup per CU every k-tile exposes its barrier + fragment-read latency
// (the MFMA pipe idles ~45 % of the time); a second resident workgroup fills those gaps.
template <bool A_KC, bool B_KC, int NS = 3>
__device__ __forceinline__ void gemm_f32_glds_body(const GemmArgs& p, int bz, float* __restrict__ smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  const int kbeg = bz * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  if (kbeg >= kend) return;
  const int nk = (kend - kbeg) / BK;

  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float csum = 0.f;
  const bool do_colsum = (!A_KC) && p.colsum != nullptr && blockIdx.x == 0;

#define G3_ISSUE(kt)                                                                          \
  do {                                                                                        \
    float* st_ = smem + ((kt) % NS) * G3_STAGE_FLOATS;                                        \
    stage_f32<A_KC>(p.A, p.lda, m0, p.M, kbeg + (kt) * BK, st_, wave, lane);                  \
    stage_f32<B_KC>(p.B, p.ldb, n0, p.N, kbeg + (kt) * BK, st_ + 64 * BK, wave, lane);        \
  } while (0)
  G3_ISSUE(0);
  if (NS == 3 && nk > 1) G3_ISSUE(1);
  for (int kt = 0; kt < nk; ++kt) {
    if (NS == 3 && kt + 1 < nk)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NS - 1 < nk) G3_ISSUE(kt + NS - 1);
    const float* As = smem + (kt % NS) * G3_STAGE_FLOATS;
    const float* Bs = As + 64 * BK;
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    float fa[BK / 2], fb[BK / 2];
    frags_f32<A_KC>(As, ar, kh, fa);
    frags_f32<B_KC>(Bs, bc, kh, fb);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc, 0, 0, 0);
    if (do_colsum && tid < 64) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) csum += As[k * 64 + tid];
    }
  }
#undef G3_ISSUE

  const int gn = n0 + wn * 32 + (lane & 31);
  const float bias = (p.bias != nullptr && gn < p.N) ? p.bias[gn] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int gm = m0 + wm * 32 + row;
    if (gm < p.M && gn < p.N) {
      float v = acc[r] + bias;
      if (p.relu) v = fmaxf(v, 0.f);
      if (p.mask != nullptr) v = p.mask[(size_t)gm * p.ldmask + gn] > 0.f ? v : 0.f;
      float* c = p.C + (size_t)gm * p.ldc + gn;
      if (p.atomic)
        atomic_add_f32(c, v);
      else
        *c = v;
    }
  }
  if (do_colsum && tid < 64 && m0 + tid < p.M) atomic_add_f32(p.colsum + m0 + tid, csum);
}

template <bool A_KC, bool B_KC, int NS = 3>
__global__ __launch_bounds__(256) void gemm_f32_glds_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[NS * G3_STAGE_FLOATS];  // 96 / 64 KB, one LDS object
  gemm_f32_glds_body<A_KC, B_KC, NS>(p, blockIdx.z, smem);
}

// Batched form (stage-2 attention: one problem per (image, head)): blockIdx.z = b0 * nb1 + b1 selects
// the operand bases, no k-split.
struct BatchDims {
  int nb1;
  long long sA0, sA1, sB0, sB1, sC0, sC1;
};
template <bool A_KC, bool B_KC, int NS = 3>
__global__ __launch_bounds__(256) void gemm_f32_glds_batched_kernel(GemmArgs p, BatchDims d) {
  __shared__ __attribute__((aligned(16))) float smem[NS * G3_STAGE_FLOATS];
  const int b0 = blockIdx.z / d.nb1, b1 = blockIdx.z - b0 * d.nb1;
  GemmArgs q = p;
  q.A = p.A + b0 * d.sA0 + b1 * d.sA1;
  q.B = p.B + b0 * d.sB0 + b1 * d.sB1;
  q.C = p.C + b0 * d.sC0 + b1 * d.sC1;
  gemm_f32_glds_body<A_KC, B_KC, NS>(q, 0, smem);
}

// Single-k-tile problems (q k^T and dO v^T of the stage-2 attention: K = 64) gain nothing from the 3-stage ring, and
// its 96 KB of LDS hold a CU to ONE workgroup whose only tile is pure load -> MFMA -> store latency (measured:
// 27 TF/s).  The register-staged body needs 35 KB: four workgroups per CU overlap each other's latencies.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_batched_small_k_kernel(GemmArgs p, BatchDims d) {
  __shared__ __attribute__((aligned(16))) float As[Tile<A_KC, 64, 256, 64>::LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[Tile<B_KC, 64, 256, 64>::LDS_FLOATS];
  const int b0 = blockIdx.z / d.nb1, b1 = blockIdx.z - b0 * d.nb1;
  GemmArgs q = p;
  q.A = p.A + b0 * d.sA0 + b1 * d.sA1;
  q.B = p.B + b0 * d.sB0 + b1 * d.sB1;
  q.C = p.C + b0 * d.sC0 + b1 * d.sC1;
  if (p.smul != nullptr) {
    q.smul = p.smul + b0 * d.sC0 + b1 * d.sC1;
    q.rowsub = p.rowsub + (size_t)blockIdx.z * p.M;
  }
  gemm_f32_body<A_KC, B_KC, 2, 2, 64>(q, blockIdx.x, blockIdx.y, 0, As, Bs);
}

// ---- 128 x 128 x 32 tile for LARGE-m forward layers (the fp32 extractor: m = views x 1408 tokens) ----------------------
// Round 5 ("fp32 extractor GEMM 109 -> >= 125 TF/s").  The 64 x 64 tile above gives every wave ONE 32 x 32
// accumulator: per MFMA (64 cycles) a wave reads two fragment values per lane from LDS and a workgroup moves (64 + 64) rows
// through L2 -> LDS per 64 x 64 outputs.  Here a wave owns 64 x 64 outputs = 2 x 2 accumulators (64 VGPRs): every A / B
// fragment value feeds TWO MFMAs, four independent 64-cycle accumulation chains keep the pipe issuing back to back, and the
// workgroup's operand traffic per flop halves.  BK = 32 keeps the LDS rows at 128 B (whole lines per DMA row piece), the
// stage at 32 KB and TWO stages at 64 KB, i.e. TWO workgroups per CU (one wave each per SIMD) that fill each other's
// barrier / fragment-read gaps, as in the 64 x 64 kernel.  LDS image: lane-linear rows of 32 floats, 16-B chunk c of row r at
// position c ^ ((r >> 1) & 7) (applied at the DMA source and at the b128 fragment reads: the 16 lanes of a read group cover
// all 64 banks).  Both operands k-contiguous, M % 128 == N % 128 == K % 32 == 0 (the caller falls back otherwise).
constexpr int GB_BK = 32;
constexpr int GB_TILE_FLOATS = 128 * GB_BK;        // one operand tile: 16 KB
constexpr int GB_STAGE_FLOATS = 2 * GB_TILE_FLOATS;  // A + B: 32 KB

__device__ __forceinline__ void stage_big(const float* __restrict__ X, int ld, int r0, int k0, float* lds, int wave, int lane) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = it * 256 + wave * 64 + lane;  // 16-B slot of the 16-KB tile: row s >> 3, position s & 7
    const int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    glds16f(X + (size_t)(r0 + row) * ld + k0 + c * 4, lds + (it * 256 + wave * 64) * 4);
  }
}

// 16-B piece q (k-values 4 q .. 4 q + 3 of k half `kh`) of row `row`: what one lane feeds into MFMA steps 4 q .. 4 q + 3
__device__ __forceinline__ float4 frag4_big(const float* __restrict__ S, int row, int kh, int q) {
  return *reinterpret_cast<const float4*>(S + row * GB_BK + (((kh * (GB_BK / 8) + q) ^ ((row >> 1) & 7)) << 2));
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_big_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * GB_STAGE_FLOATS];  // 64 KB, one LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
  const int nk = p.K / GB_BK;
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#define GB_ISSUE(kt)                                                             \
  do {                                                                           \
    float* st_ = smem + ((kt) & 1) * GB_STAGE_FLOATS;                            \
    stage_big(p.A, p.lda, m0, (kt) * GB_BK, st_, wave, lane);                    \
    stage_big(p.B, p.ldb, n0, (kt) * GB_BK, st_ + GB_TILE_FLOATS, wave, lane);   \
  } while (0)
  GB_ISSUE(0);
  const int l31 = lane & 31, kh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (own DMAs), everyone's has, and everyone is done reading the other buffer (tile kt - 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) GB_ISSUE(kt + 1);
    const float* As = smem + (kt & 1) * GB_STAGE_FLOATS;
    const float* Bs = As + GB_TILE_FLOATS;
    // reads in the order the MFMA steps need them (piece q of all four fragments, then q + 1): the first 16 MFMAs can
    // start behind four reads instead of thirteen
    float4 fa[2][GB_BK / 8], fb[2][GB_BK / 8];
#pragma unroll
    for (int q = 0; q < GB_BK / 8; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i][q] = frag4_big(As, wm * 64 + i * 32 + l31, kh, q);
        fb[i][q] = frag4_big(Bs, wn * 64 + i * 32 + l31, kh, q);
      }
#pragma unroll
    for (int q = 0; q < GB_BK / 8; ++q) {
#define GB_STEP(E)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q].E, fb[j][q].E, acc[i][j], 0, 0, 0);
      GB_STEP(x) GB_STEP(y) GB_STEP(z) GB_STEP(w)
#undef GB_STEP
    }
  }
#undef GB_ISSUE
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): a half-wave stores one
  // 128-B row piece per instruction
  // EPI (round 5, the fp32 extractor): the exact-erf GELU of fc1 and the LayerScale + residual of proj / fc2 happen HERE, on
  // the accumulators, instead of in a separate pass that re-read and re-wrote the fp32 [tokens, 3072] / [tokens, 768] tensors
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gn = n0 + wn * 64 + j * 32 + l31;
    const float bias = p.bias != nullptr ? p.bias[gn] : 0.f;
    const float gm_n = EPI == 2 ? p.gamma[gn] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float v = acc[i][j][r] + bias;
        float* c = p.C + (size_t)gm * p.ldc + gn;
        if constexpr (EPI == 1) {
          v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));  // nn.GELU(): exact erf form (as gelu_f32_kernel)
        } else if constexpr (EPI == 2) {
          v = *c + gm_n * v;  // (as resid_f32_kernel: o += g * v)
        } else {
          if (p.relu) v = fmaxf(v, 0.f);
        }
        *c = v;
      }
  }
}

// ---- the same 128 x 128 tile for the WEIGHT-GRADIENT product of the stage-2 trainer (round 6): C[M][N] (+)= sum_k A(m, k) B(k, n) with
// BOTH operands stored reduction-index-major -- A_mem[k][m] = dy[row][out feature], B_mem[k][n] = x[row][in feature]; the
// reduction runs over the 45 056 token rows of a batch, split over blockIdx.z with one atomic add per split and element.
// A stage is 32 rows of each operand, [row][128] floats as they lie in memory (whole 512-B row pieces per LDS-DMA
// instruction, lane-linear: no swizzle needed -- an MFMA 32x32x2 operand is ONE float per lane, lane i of a half-wave reads
// column i of row 2 s + (lane >> 5): 32 consecutive dwords, conflict-free ds_read_b32).  Per stage and wave 64 reads feed 64
// MFMAs (2 x 2 accumulator blocks, four independent chains).  colsum[m] += sum_k A(m, k) (the bias gradient) by the n-tile-0
// blocks from the fragments they read anyway.  Two 32-KB stages = two workgroups per CU, as the forward kernel.
__device__ __forceinline__ void stage_rows_big(const float* __restrict__ X, int ld, int k0, int c0, float* lds, int wave, int lane) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = it * 256 + wave * 64 + lane;  // 16-B slot: row s >> 5, columns 4 (s & 31) ..
    glds16f(X + (size_t)(k0 + (s >> 5)) * ld + c0 + (s & 31) * 4, lds + (it * 256 + wave * 64) * 4);
  }
}

__global__ __launch_bounds__(256) void gemm_f32_big_tn_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * GB_STAGE_FLOATS];  // 64 KB, one LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
  const int kbeg = blockIdx.z * p.kchunk, kend = kbeg + p.kchunk < p.K ? kbeg + p.kchunk : p.K;
  const int nk = (kend - kbeg) / GB_BK;
  const bool do_colsum = p.colsum != nullptr && blockIdx.x == 0 && wn == 0;
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float csum[2] = {0.f, 0.f};
#define GT_ISSUE(kt)                                                                       \
  do {                                                                                     \
    float* st_ = smem + ((kt) & 1) * GB_STAGE_FLOATS;                                      \
    stage_rows_big(p.A, p.lda, kbeg + (kt) * GB_BK, m0, st_, wave, lane);                  \
    stage_rows_big(p.B, p.ldb, kbeg + (kt) * GB_BK, n0, st_ + GB_TILE_FLOATS, wave, lane); \
  } while (0)
  if (nk > 0) GT_ISSUE(0);
  const int l31 = lane & 31, kh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) GT_ISSUE(kt + 1);
    const float* As = smem + (kt & 1) * GB_STAGE_FLOATS + kh * 128 + wm * 64 + l31;
    const float* Bs = smem + (kt & 1) * GB_STAGE_FLOATS + GB_TILE_FLOATS + kh * 128 + wn * 64 + l31;
#pragma unroll
    for (int s = 0; s < GB_BK / 2; ++s) {  // rows 2 s + kh of the stage
      const float a0 = As[s * 256], a1 = As[s * 256 + 32], b0 = Bs[s * 256], b1 = Bs[s * 256 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      if (do_colsum) {
        csum[0] += a0;
        csum[1] += a1;
      }
    }
  }
#undef GT_ISSUE
  // C/D layout of the 32x32 MFMA: col = lane & 31 (n), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (m)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gn = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float* c = p.C + (size_t)gm * p.ldc + gn;
        if (p.atomic)
          atomic_add_f32(c, acc[i][j][r]);
        else
          *c = acc[i][j][r];
      }
  }
  if (do_colsum) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
      if (kh == 0) atomic_add_f32(p.colsum + m0 + wm * 64 + i * 32 + l31, v);
    }
  }
}

int g_f32_big = 1;  // dvt_tune_set(4, 10 / 11): the 128 x 128 kernel of dvt_linear_fwd_big off / on (A/B; results differ in summation order only)
int g_f32_glds = 1;  // 0: always the register-staged kernel
int g_f32_ex_stages = 2;  // LDS stages of the stage-2 GEMMs (dvt_gemm_f32_ex): 2 or 3
extern int g_cfg_override;

// eligibility of the LDS-DMA kernel: whole k-tiles, and row-contiguous operands whose 64-wide
// tile never runs past the row end (k-contiguous operands are clamped per row instead)
template <bool A_KC, bool B_KC>
bool glds_ok(const GemmArgs& a) {
  if (!g_f32_glds || (a.K % BK) || (a.kchunk % BK) || g_cfg_override > 0) return false;
  if (!A_KC && (a.M % 64)) return false;
  if (!B_KC && (a.N % 64)) return false;
  return a.M >= 1 && a.N >= 1;
}

// Tile configurations: 0 = 64x64 (4 waves), 1 = 32x64 (2 waves), 2 = 32x32 (1 wave), 3 = 64x32.
int g_cfg_override = -1;

struct TileCfg {
  int bm, bn;
};
constexpr TileCfg kCfgs[4] = {{64, 64}, {32, 64}, {32, 32}, {64, 32}};

// The batch is small (2048 rows): pick the largest tile that still yields >= ~2 workgroups
// per CU worth of waves, so that barrier / load phases of one wave hide behind another's MFMAs.
int pick_cfg(int M, int N, int ksplits) {
  // Measured at the fit's shapes (M = 2048, N, K <= 768): the per-tile time is set by the
  // staging latency, not by the number of workgroups -- 64x64 wins everywhere (322.6 us/step vs
  // 340.2 / 333.1 for 32x64 / 32x32), so it is the default; smaller tiles only for tiny outputs.
  if (g_cfg_override >= 0 && g_cfg_override < 4) return g_cfg_override;
  (void)ksplits;
  if (M <= 32 || N <= 32) return 2;
  return 0;
}

// k-depth of the register-staged kernel: 64, 32 or 16.  Default 32 (17 KB of LDS per workgroup):
// in the pipelined driver the fit shares every CU with a 136-KB ViT GEMM workgroup, and a fit
// kernel that does not fit beside it has to wait for CUs to drain (measured: 1.55 -> 1.72 images/s;
// the serial fit time is unchanged, 0.345 s).  64 re-enables the 96-KB LDS-DMA kernel.
int g_f32_bk = 32;

template <bool A_KC, bool B_KC, int WM, int WN, int BKT>
int launch_cfg(const GemmArgs& a, int ksplits, hipStream_t s) {
  dim3 grid(dvt_cdiv(a.N, 32 * WN), dvt_cdiv(a.M, 32 * WM), ksplits);
  DvtProbeScope probe(DVT_PROBE_FIT_GEMM, s, 2.0 * a.M * a.N * a.K);
  hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, WM, WN, BKT>), grid, dim3(64 * WM * WN), 0, s, a);
  DVT_CHECK_LAUNCH();
  return 0;
}

template <bool A_KC, bool B_KC>
int launch(const GemmArgs& a, int ksplits, hipStream_t s) {
  if (g_f32_bk == 64 && glds_ok<A_KC, B_KC>(a)) {
    dim3 grid(dvt_cdiv(a.N, 64), dvt_cdiv(a.M, 64), ksplits);
    DvtProbeScope probe(DVT_PROBE_FIT_GEMM, s, 2.0 * a.M * a.N * a.K);
    hipLaunchKernelGGL((gemm_f32_glds_kernel<A_KC, B_KC>), grid, dim3(256), 0, s, a);
    DVT_CHECK_LAUNCH();
    return 0;
  }
  const int cfg = pick_cfg(a.M, a.N, ksplits);
  if (g_f32_bk == 16) return launch_cfg<A_KC, B_KC, 2, 2, 16>(a, ksplits, s);
  if (g_f32_bk == 32) return launch_cfg<A_KC, B_KC, 2, 2, 32>(a, ksplits, s);
  switch (cfg) {
    case 0: return launch_cfg<A_KC, B_KC, 2, 2, 64>(a, ksplits, s);
    case 1: return launch_cfg<A_KC, B_KC, 1, 2, 64>(a, ksplits, s);
    case 3: return launch_cfg<A_KC, B_KC, 2, 1, 64>(a, ksplits, s);
    default: return launch_cfg<A_KC, B_KC, 1, 1, 64>(a, ksplits, s);
  }
}

// ---- GemmArgs builders for the three contractions of a linear layer ----
GemmArgs make_fwd(const float* x, const float* w, const float* b, float* y, int m, int n, int k,
                  int relu) {
  GemmArgs a{};
  a.A = x; a.B = w; a.C = y;
  a.M = m; a.N = n; a.K = k;
  a.lda = k; a.ldb = k; a.ldc = n;
  a.bias = b; a.relu = relu;
  a.kchunk = (k + 63) / 64 * 64;
  return a;
}

// dw[n,k] += sum_b dy[b,n] * x[b,k]; the batch reduction is split so that ~384 workgroups exist
GemmArgs make_wgrad(const float* dy, const float* x, float* dw, float* db, int m, int n, int k,
                    int* splits_out) {
  GemmArgs a{};
  a.A = dy; a.B = x; a.C = dw;
  a.M = n; a.N = k; a.K = m;
  a.lda = n; a.ldb = k; a.ldc = k;
  a.colsum = db;
  a.atomic = 1;
  const int tiles = dvt_cdiv(n, 64) * dvt_cdiv(k, 64);
  const int ktiles = dvt_cdiv(m, 64);
  int splits = dvt_cdiv(384, tiles);  // every split costs one fp32 atomic per output element
  if (splits > ktiles / 2) splits = ktiles / 2;
  if (splits < 1) splits = 1;
  a.kchunk = dvt_cdiv(ktiles, splits) * 64;
  *splits_out = dvt_cdiv(m, a.kchunk);
  return a;
}

GemmArgs make_dgrad(const float* dy, const float* w, float* dx, const float* relu_mask, int m,
                    int n, int k) {
  GemmArgs a{};
  a.A = dy; a.B = w; a.C = dx;
  a.M = m; a.N = k; a.K = n;
  a.lda = n; a.ldb = k; a.ldc = k;
  a.mask = relu_mask; a.ldmask = k;
  a.kchunk = (n + 63) / 64 * 64;
  return a;
}

}  // namespace

int dvt_linear_group(const DvtLinearOp* ops, int n_ops, hipStream_t s, int bf16_operands) {
  if (!ops || n_ops < 1 || n_ops > MULTI_MAX) return DVT_E_BADARG;
  MultiArgs m{};
  m.n = n_ops;
  int blocks = 0;
  double flops = 0.0;
  for (int i = 0; i < n_ops; ++i) {
    const DvtLinearOp& o = ops[i];
    if (o.m <= 0 || o.n <= 0 || o.k <= 0 || (o.n & 3) || (o.k & 3)) return DVT_E_BADARG;
    int splits = 1;
    if (o.kind == 0) {
      m.g[i] = make_fwd(o.x, o.w, o.b, o.y, o.m, o.n, o.k, o.relu);
      m.layout[i] = 0;
    } else if (o.kind == 1) {
      m.g[i] = make_wgrad(o.dy, o.x, o.dw, o.db, o.m, o.n, o.k, &splits);
      m.layout[i] = 2;
    } else if (o.kind == 2) {
      m.g[i] = make_dgrad(o.dy, o.w, o.dx, o.relu_mask, o.m, o.n, o.k);
      m.layout[i] = 1;
    } else {
      return DVT_E_BADARG;
    }
    m.gx[i] = dvt_cdiv(m.g[i].N, 64);
    m.gy[i] = dvt_cdiv(m.g[i].M, 64);
    m.blk0[i] = blocks;
    blocks += m.gx[i] * m.gy[i] * splits;
    flops += 2.0 * o.m * o.n * o.k;
  }
  m.blk0[n_ops] = blocks;
  DvtProbeScope probe(DVT_PROBE_FIT_GEMM, s, flops);
  if (bf16_operands) {
    // BK = 32 (80 VGPRs, 10 KB of LDS): fits beside the extractor's workgroups; BK = 64 (146 VGPRs)
    // is equally fast alone (214 vs 215 us/step) and slower in the pipeline (2.09 vs 2.12 images/s)
    if (g_f32_bk == 64)
      hipLaunchKernelGGL(gemm_bf16op_multi_kernel<64>, dim3(blocks), dim3(256), 0, s, m);
    else
      hipLaunchKernelGGL(gemm_bf16op_multi_kernel<32>, dim3(blocks), dim3(256), 0, s, m);
  } else if (g_f32_bk == 16)
    hipLaunchKernelGGL(gemm_f32_multi_kernel<16>, dim3(blocks), dim3(256), 0, s, m);
  else if (g_f32_bk == 32)
    hipLaunchKernelGGL(gemm_f32_multi_kernel<32>, dim3(blocks), dim3(256), 0, s, m);
  else
    hipLaunchKernelGGL(gemm_f32_multi_kernel<64>, dim3(blocks), dim3(256), 0, s, m);
  DVT_CHECK_LAUNCH();
  return 0;
}

namespace {
extern int g_cfg_override;
}
extern int g_fit_fused_enable;
extern int g_fit_fused_f32;
extern int g_fit_sorted_grid;
extern int g_fit_lazy_adam;
extern int g_fit_lazy_refresh;
extern int g_fit_lazy_exact;
extern int g_fit_lazy_merge;
extern int g_fit_shadow_in_adam;
extern int g_adam_pingpong;
extern int g_fit_rows32;
extern int g_fit_small_wg;
extern int g_fit_xcd_affinity;
#ifdef DVT_LAB
extern int g_fit_skip_mask;
#endif

extern "C" int dvt_tune_set(int key, int value) {
  if (key == 0) {
    g_cfg_override = value;
    return 0;
  }
  if (key == 4) {
    if (value == 2 || value == 3)
      g_f32_ex_stages = value;  // stage-2 GEMMs: LDS stages
    else if (value == 10 || value == 11)
      g_f32_big = value == 11;  // fp32 extractor: 128 x 128 x 32 tile (11, default) / the 64 x 64 x 64 LDS-DMA kernel (10)
    else
      g_f32_glds = value;
    return 0;
  }
  if (key == 5) {
    if (value != 16 && value != 32 && value != 64) return DVT_E_BADARG;
    g_f32_bk = value;
    return 0;
  }
  if (key == 8) {
    g_adam_pingpong = value != 0;
    return 0;
  }
  if (key == 6) {
    if (value == 2 || value == 3)
      g_fit_fused_f32 = value == 3;  // fp32-operand fused step off / on (default on)
    else
      g_fit_fused_enable = value != 0;
    return 0;
  }
  if (key == 7) {
    g_fit_sorted_grid = value != 0;
    return 0;
  }
  if (key == 10) {
    g_fit_lazy_exact = value != 0;
    return 0;
  }
  if (key == 11) {
    g_fit_lazy_merge = value != 0;
    return 0;
  }
  if (key == 12) {
    g_fit_shadow_in_adam = value != 0;
    return 0;
  }
  if (key == 13) {
    if (value < 0 || value > 2) return DVT_E_BADARG;
    g_fit_rows32 = value;
    return 0;
  }
  if (key == 14) {
    g_fit_small_wg = value != 0;
    return 0;
  }
  if (key == 16) {
    g_fit_xcd_affinity = value != 0;
    return 0;
  }
#ifdef DVT_LAB
  if (key == 15) {  // timing-only ablation mask of the fused fit step (dvt_fit.hip)
    g_fit_skip_mask = value & 7;
    return 0;
  }
#endif
  if (key == 9) {
    g_fit_lazy_adam = value != 0;
    if (value >= 2) g_fit_lazy_refresh = value;
    return 0;
  }
  if (key == 18) return dvt_s2_tune(value);
  if (key == 1) return dvt_vit_tune(value);
  if (key == 2) return dvt_grid_tune(value);
  if (key == 3) return dvt_adam_tune(value);
  return DVT_E_BADARG;
}

// General exact-fp32 GEMM for the stage-2 trainer (dvt_stage2.hip): any of the three operand layouts with
// explicit leading dimensions, optional (image, head) batching, optional atomic accumulation with a k-split.
// Always the 3-stage LDS-DMA kernel (there is no extractor to share the CUs with in stage 2), so the shapes
// must be whole 64-tiles where that kernel needs them (checked).
int dvt_gemm_f32_ex(const DvtGemmEx* g, hipStream_t s) {
  if (!g || !g->A || !g->B || !g->C || g->M <= 0 || g->N <= 0 || g->K <= 0) return DVT_E_BADARG;
  GemmArgs a{};
  a.A = g->A; a.B = g->B; a.C = g->C;
  a.M = g->M; a.N = g->N; a.K = g->K;
  a.lda = g->lda; a.ldb = g->ldb; a.ldc = g->ldc;
  a.bias = g->bias;
  a.colsum = g->layout == 2 ? g->colsum : nullptr;
  a.atomic = g->accumulate;
  if (g->smul != nullptr) {  // softmax-backward epilogue: the batched small-k kernel only
    const int nb_ = (g->nb0 > 0 ? g->nb0 : 1) * (g->nb1 > 0 ? g->nb1 : 1);
    if (!g->rowsub || g->layout != 0 || nb_ <= 1 || g->K > BK || g->accumulate) return DVT_E_BADARG;
    a.smul = g->smul;
    a.rowsub = g->rowsub;
    a.oscale = g->oscale;
  }
  if (a.K % BK) return DVT_E_BADARG;
  const bool a_kc = g->layout != 2, b_kc = g->layout == 0;
  if ((!a_kc && (a.M % 64)) || (!b_kc && (a.N % 64)) || (a.lda & 3) || (a.ldb & 3)) return DVT_E_BADARG;
  const int nb = (g->nb0 > 0 ? g->nb0 : 1) * (g->nb1 > 0 ? g->nb1 : 1);
  int splits = 1;
  a.kchunk = a.K;
  if (g->accumulate && nb == 1) {  // split the reduction so that ~1024 workgroups exist (one atomic per split)
    const int tiles = dvt_cdiv(a.M, 64) * dvt_cdiv(a.N, 64), ktiles = a.K / BK;
    splits = dvt_cdiv(1024, tiles);
    if (splits > ktiles / 4) splits = ktiles / 4;
    if (splits < 1) splits = 1;
    a.kchunk = dvt_cdiv(ktiles, splits) * BK;
    splits = dvt_cdiv(a.K, a.kchunk);
  }
  if (nb > 65535 || splits > 65535) return DVT_E_BADARG;
  dim3 grid(dvt_cdiv(a.N, 64), dvt_cdiv(a.M, 64), nb > 1 ? nb : splits);
  DvtProbeScope probe(DVT_PROBE_FIT_GEMM, s, 2.0 * a.M * a.N * a.K * nb);
#define EX_LAUNCH(NS)                                                                                     \
  do {                                                                                                    \
    if (nb > 1) {                                                                                         \
      BatchDims d{g->nb1 > 0 ? g->nb1 : 1, g->sA0, g->sA1, g->sB0, g->sB1, g->sC0, g->sC1};               \
      if (a.K <= BK && g->layout == 0)                                                                    \
        hipLaunchKernelGGL((gemm_f32_batched_small_k_kernel<true, true>), grid, dim3(256), 0, s, a, d);   \
      else if (g->layout == 0)                                                                            \
        hipLaunchKernelGGL((gemm_f32_glds_batched_kernel<true, true, NS>), grid, dim3(256), 0, s, a, d);  \
      else if (g->layout == 1)                                                                            \
        hipLaunchKernelGGL((gemm_f32_glds_batched_kernel<true, false, NS>), grid, dim3(256), 0, s, a, d); \
      else                                                                                                \
        hipLaunchKernelGGL((gemm_f32_glds_batched_kernel<false, false, NS>), grid, dim3(256), 0, s, a, d);\
    } else {                                                                                              \
      if (g->layout == 0)                                                                                 \
        hipLaunchKernelGGL((gemm_f32_glds_kernel<true, true, NS>), grid, dim3(256), 0, s, a);             \
      else if (g->layout == 1)                                                                            \
        hipLaunchKernelGGL((gemm_f32_glds_kernel<true, false, NS>), grid, dim3(256), 0, s, a);            \
      else                                                                                                \
        hipLaunchKernelGGL((gemm_f32_glds_kernel<false, false, NS>), grid, dim3(256), 0, s, a);           \
    }                                                                                                     \
  } while (0)
  if (g_f32_ex_stages == 2)
    EX_LAUNCH(2);
  else
    EX_LAUNCH(3);
#undef EX_LAUNCH
  DVT_CHECK_LAUNCH();
  return 0;
}

// the 128 x 128 x 32 kernel takes this shape (and with it the fused GELU / residual epilogues below)
bool dvt_linear_big_ok(int m, int n, int k) { return g_f32_big && m > 0 && m % 128 == 0 && n % 128 == 0 && k % GB_BK == 0; }

// dw[n][k] (+)= sum_r dy[r][n] x[r][k], db[n] += sum_r dy[r][n] on the 128 x 128 tile (gemm_f32_big_tn_kernel); n, k multiples of
// 128, rows of 32; `accumulate`: atomic adds into dw (the reduction is split so that ~1024 workgroups exist), else ONE split
bool dvt_linear_wgrad_big_ok(int rows, int n, int k) { return g_f32_big && rows > 0 && rows % GB_BK == 0 && n % 128 == 0 && k % 128 == 0; }
int dvt_linear_wgrad_big(const float* dy, const float* x, float* dw, float* db, int rows, int n, int k, int accumulate,
                         hipStream_t s) {
  if (!dy || !x || !dw || !dvt_linear_wgrad_big_ok(rows, n, k)) return DVT_E_BADARG;
  GemmArgs a{};
  a.A = dy; a.B = x; a.C = dw;
  a.M = n; a.N = k; a.K = rows;
  a.lda = n; a.ldb = k; a.ldc = k;
  a.colsum = db;
  a.atomic = accumulate;
  const int tiles = (n / 128) * (k / 128), ktiles = rows / GB_BK;
  int splits = 1;
  if (accumulate) {
    splits = dvt_cdiv(1024, tiles);
    if (splits > ktiles / 8) splits = ktiles / 8;
    if (splits < 1) splits = 1;
  }
  a.kchunk = dvt_cdiv(ktiles, splits) * GB_BK;
  splits = dvt_cdiv(rows, a.kchunk);
  DvtProbeScope probe(DVT_PROBE_FIT_GEMM, s, 2.0 * a.M * a.N * a.K);
  hipLaunchKernelGGL(gemm_f32_big_tn_kernel, dim3(k / 128, n / 128, splits), dim3(256), 0, s, a);
  DVT_CHECK_LAUNCH();
  return 0;
}

// y = epi(x . w^T + b) on the 128 x 128 x 32 kernel ONLY (dvt_linear_big_ok): epi 1 = exact-erf GELU, epi 2 = y += gamma (.) (.)
int dvt_linear_fwd_big_epi(const float* x, const float* w, const float* b, float* y, int m, int n, int k, int epi,
                           const float* gamma, hipStream_t s) {
  if (!x || !w || !y || !dvt_linear_big_ok(m, n, k) || (epi != 1 && epi != 2) || (epi == 2 && !gamma)) return DVT_E_BADARG;
  GemmArgs a = make_fwd(x, w, b, y, m, n, k, 0);
  a.epi = epi;
  a.gamma = gamma;
  DvtProbeScope probe(DVT_PROBE_FIT_GEMM, s, 2.0 * a.M * a.N * a.K);
  if (epi == 1) hipLaunchKernelGGL(gemm_f32_big_kernel<1>, dim3(a.N / 128, a.M / 128), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gemm_f32_big_kernel<2>, dim3(a.N / 128, a.M / 128), dim3(256), 0, s, a);
  DVT_CHECK_LAUNCH();
  return 0;
}

// Forward linear for LARGE m (the fp32 extractor: m = views x 1408 tokens): the 128 x 128 x 32 kernel where the shape is
// eligible, else the 64 x 64 LDS-DMA kernel, whatever the co-residency knob of the fit says.
int dvt_linear_fwd_big(const float* x, const float* w, const float* b, float* y, int m, int n, int k, hipStream_t s) {
  if (!x || !w || !y || m <= 0 || n <= 0 || k <= 0 || (k & 3) || (n & 3)) return DVT_E_BADARG;
  const GemmArgs a = make_fwd(x, w, b, y, m, n, k, 0);
  if (dvt_linear_big_ok(m, n, k)) {
    DvtProbeScope probe(DVT_PROBE_FIT_GEMM, s, 2.0 * a.M * a.N * a.K);
    hipLaunchKernelGGL(gemm_f32_big_kernel<0>, dim3(a.N / 128, a.M / 128), dim3(256), 0, s, a);
    DVT_CHECK_LAUNCH();
    return 0;
  }
  if ((a.K % BK) == 0 && (a.kchunk % BK) == 0) {
    dim3 grid(dvt_cdiv(a.N, 64), dvt_cdiv(a.M, 64), 1);
    DvtProbeScope probe(DVT_PROBE_FIT_GEMM, s, 2.0 * a.M * a.N * a.K);
    // 2 LDS stages (64 KB): two workgroups share a CU and fill each other's barrier / fragment-read gaps -- the stage-2
    // trainer measured 81 -> 97 TF/s with it; round 3 gives the fp32 extractor the same ring (dvt_tune_set(4, 3)
    // restores the 3-stage ring for A/B).  Same arithmetic order: results are bit-identical.
    if (g_f32_ex_stages == 2)
      hipLaunchKernelGGL((gemm_f32_glds_kernel<true, true, 2>), grid, dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((gemm_f32_glds_kernel<true, true, 3>), grid, dim3(256), 0, s, a);
    DVT_CHECK_LAUNCH();
    return 0;
  }
  return launch<true, true>(a, 1, s);
}

extern "C" int dvt_linear_fwd(const float* x, const float* w, const float* b, float* y, int m,
                              int n, int k, int relu, void* stream) {
  if (!x || !w || !y || m < 0 || n <= 0 || k <= 0 || (k & 3) || (n & 3)) return DVT_E_BADARG;
  if (m == 0) return 0;
  return launch<true, true>(make_fwd(x, w, b, y, m, n, k, relu), 1, (hipStream_t)stream);
}

extern "C" int dvt_linear_bwd(const float* dy, const float* x, const float* w, float* dw,
                              float* db, float* dx, const float* relu_mask, int m, int n, int k,
                              void* stream) {
  if (!dy || m < 0 || n <= 0 || k <= 0 || (k & 3) || (n & 3)) return DVT_E_BADARG;
  if (m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dw != nullptr) {
    if (!x) return DVT_E_BADARG;
    int splits = 1;
    const GemmArgs a = make_wgrad(dy, x, dw, db, m, n, k, &splits);
    int rc = launch<false, false>(a, splits, s);
    if (rc) return rc;
  } else if (db != nullptr) {
    return DVT_E_BADARG;  // db is produced by the wgrad launch
  }
  if (dx != nullptr) {
    if (!w) return DVT_E_BADARG;
    int rc = launch<true, false>(make_dgrad(dy, w, dx, relu_mask, m, n, k), 1, s);
    if (rc) return rc;
  }
  return 0;
}
