// The fused ROW kernel of the fit step: everything of one Adam step that is local to a sampled row
// runs in ONE launch -- raw-row gather, hash-grid forward, the field MLP and the residual predictor
// forward, the loss with all its gradients, and the data-gradient (dgrad) chain back to the encoding.
//
// Reference: the body of the inner loop, main_img_denoising.py:73-88, through
// dvt/models/neural_feature_field.py:46-49 (enc -> Linear/ReLU/Linear) and
// dvt/models/offline_denoiser.py:96-140 (G lookup, residual predictor, losses) and autograd's
// backward of the same.  Two operand precisions (template parameter F32 of everything below):
//   bf16 (DvtFitConfig.mlp_bf16, the reference's `--dtype bfloat16`): operands rounded to bf16, v_mfma_f32_16x16x32_bf16,
//        accumulation / bias / ReLU / losses fp32;
//   fp32 (round 5; the reference's default `--dtype float32`): fp32 operands end to end on v_mfma_f32_16x16x4_f32 -- the
//        exact-fp32 matrix instruction, an fmaf chain per output -- read FOUR steps at a time, so that LDS images, weight
//        fragments and transposed copies keep the bf16 path's geometry: 16 bytes per lane and step group, 1 KB per
//        (tile, step group); a step group covers 16 k instead of 32 (dvt_common.h: dvt_frag_off32).  C <= 768 (the
//        phase-2 LDS images of C = 1024 are 206 KB); until round 4 this mode ran five layer-GEMM launches per step.
//
// Why this shape.  With B = 2048 sampled rows the step used to be 5 dependent grouped-GEMM launches +
// the loss, each latency-bound (65 TF/s = 2.6 % of the bf16 MFMA peak) with every activation
// round-tripping HBM.  Forward, loss and dgrad are ROW-LOCAL, only the weight gradients reduce over
// rows.  So a workgroup owns 16 rows through the whole chain:
//   * activations never leave the CU between layers: bf16 [16][K] images in LDS (+16 B row padding),
//     read as MFMA A fragments (v_mfma_f32_16x16x32_bf16, M = 16 rows);
//   * the 8 waves split every layer's OUTPUT columns (tile t -> wave t % 8), so no two waves share a
//     weight element: each B fragment goes straight from global/L2 into VGPRs as one 16-byte load of a
//     k-contiguous bf16 shadow copy (dvt_common.h: DvtShadowLayout) -- an LDS stage would be pure
//     overhead (operand streamed once per workgroup, the CDNA guide's M <= 16 rule) -- software
//     pipelined PD k-steps ahead in registers;
//   * what the weight-gradient GEMMs, the grid backward and Adam need (enc, h1, dF, dh1, denc, ...) is
//     written once as fp32 rows.
// Per workgroup the bound is the weight stream through one CU (1.3 MB phase 1 / 2.3 MB phase 2 of L2
// hits); 128 workgroups cover B = 2048.
#include "dvt_common.h"
#include "dvt_grid_dev.h"
#include "dvt_loss_row.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

int g_fit_rows32 = 1;  // dvt_tune_set(13, v), see launch_rows
int g_fit_xcd_affinity = 1;  // dvt_tune_set(16, v): the row kernel's fit -> XCD map for 2 / 4 fits per launch (fit_rows_kernel) on / off
int g_fit_small_wg = 0;  // dvt_tune_set(14, v): 1 = 4-wave fit_rows + 8-wave fit_backward workgroups (same arithmetic, same results)

namespace {

constexpr int FR0 = 16;  // rows per workgroup of the round-2 kernel = M of the MFMA; the kernel template takes FR = 16 or 32
constexpr int FW8 = 8;   // waves per workgroup (template parameter FW of the kernels below: 8, or 4 = the small-footprint
                         // shape of dvt_tune_set(14, 1): one wave per SIMD, so that a workgroup fits where ONE attention
                         // workgroup of the extractor has left -- see profiles/r04/r04p_pipeline_timeline.txt)
constexpr int FE = 128;  // encoding width: 16 levels x 8 features

// operand element: bytes, and k values per 16-byte fragment piece of ONE lane group (= k per step group / 4)
template <bool F32>
struct Op {
  static constexpr int ES = F32 ? 4 : 2;    // bytes per element
  static constexpr int KS = F32 ? 16 : 32;  // k per step group (four lane groups x 16 B)
};
typedef unsigned u32x4f __attribute__((ext_vector_type(4)));  // one 16-byte fragment piece, either element type

// LDS row pitch of a [16][K] activation image: +16 B so that the 16 rows of an A-fragment read
// (ds_read_b128, 16 B per lane) fall on distinct 16-byte bank slots
template <bool F32>
__host__ __device__ constexpr int apitch(int K) { return K * Op<F32>::ES + 16; }

struct FusedArgs {
  DvtGridTable T;
  DvtShadowLayout S;
  DvtTLayout TL;
  int n, lattice;
  int xcd_group;  // XCDs per fit of the row kernel's block -> (fit, row block) map: 8 / k for k = 2, 4 fits per launch, else 0 (see fit_rows_kernel)
  float grad_scale;
  long long off_grid, off_b1, off_b2, off_G, off_bh1, off_bh2, off_bh3;
  DvtFusedFit f[DVT_FIT_BATCH_MAX];
};

__device__ __forceinline__ uint16_t bf16_of(float v) { return (uint16_t)(dvt_pack_bf16x2(v, 0.f) & 0xffffu); }

// One [FR][NC] LDS image (pitch apitch(NC)) -> its slice of the transposed, fragment-major operand copy
// dstT = [NC][B]: for every column the FR batch rows of this workgroup are 16-byte pieces (8 rows each in bf16, 4 in fp32)
// of the 1-KB fragment block (tile col / 16, step group b0 / 32 or b0 / 16).  Adjacent threads take adjacent columns.
template <int NC, int FR, int FW, bool F32>
__device__ __forceinline__ void store_T(const char* img, void* __restrict__ dstTv, int B, int b0, int tid) {
  constexpr int RPP = F32 ? 4 : 8;  // batch rows per 16-byte piece
  for (int item = tid; item < NC * (FR / RPP); item += 64 * FW) {
    const int grp = item / NC, col = item - grp * NC;
    const int brow = b0 + grp * RPP;
    const char* src = img + grp * RPP * apitch<F32>(NC) + col * Op<F32>::ES;
    if constexpr (F32) {
      const int sg = brow >> 4, g = (brow >> 2) & 3;
      float4 v;
      v.x = *reinterpret_cast<const float*>(src);
      v.y = *reinterpret_cast<const float*>(src + apitch<F32>(NC));
      v.z = *reinterpret_cast<const float*>(src + 2 * apitch<F32>(NC));
      v.w = *reinterpret_cast<const float*>(src + 3 * apitch<F32>(NC));
      const long long off = ((long long)(col >> 4) * (B >> 4) + sg) * 256 + (((g << 4) + (col & 15)) << 2);
      *reinterpret_cast<float4*>(static_cast<float*>(dstTv) + off) = v;
    } else {
      const int kstep = brow >> 5, g = (brow >> 3) & 3;
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t lo = *reinterpret_cast<const uint16_t*>(src + (2 * j) * apitch<F32>(NC));
        const uint32_t hi = *reinterpret_cast<const uint16_t*>(src + (2 * j + 1) * apitch<F32>(NC));
        w[j] = lo | (hi << 16);
      }
      const long long off = ((long long)(col >> 4) * (B >> 5) + kstep) * 512 + (((g << 4) + (col & 15)) << 3);
      *reinterpret_cast<uint4*>(static_cast<uint16_t*>(dstTv) + off) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

// out[16][N] = act(A[16][K] . W[N][K]^T + bias)  (optionally masked by mask[16][N] > 0)
//   actA    LDS image [16][K] (bf16 or fp32), pitch apitch(K)
//   W       global shadow copy of the [N][K] weight matrix in fragment-major order (dvt_frag_off / dvt_frag_off32):
//           1 KB per (tile, step group), 16 B per lane, either element type
//   act_out LDS image [16][N] for the next layer (may alias `mask`: every element is read, then
//           written, by the one lane that owns it), or nullptr
//   gout    global fp32 [16][N] (this workgroup's rows), or nullptr
template <int K, int N, bool RELU, bool MASK, int RB, int FW, bool F32>
__device__ __forceinline__ void mlp_layer(const char* actA, const void* __restrict__ Wv,
                                          const float* __restrict__ bias, char* act_out,
                                          float* __restrict__ gout, const char* mask, int wave, int lane) {
  // RB = row blocks of 16 (FR / 16): every weight fragment fetched from L2 is used for RB MFMAs (x 4 sub-steps in fp32)
  constexpr int KS = Op<F32>::KS, ES = Op<F32>::ES;
  constexpr int NTILES = N / 16, NT = (NTILES + FW - 1) / FW, S = K / KS;
  // step groups of weights in flight per wave: ~24 x 1 KB.  The weights are L2 hits at best and memory-side
  // cache hits on first touch (every kernel starts with a cold L2), i.e. 0.3-2 us of latency: with 6 loads
  // in flight per wave the first version of this kernel streamed its 1.3 MB at 24 GB/s per CU (54 us).
  constexpr int INFL = FW == 8 ? 24 : 16;  // (4-wave shape: 16, which keeps the phase-2 kernel at C = 768 under the 272 VGPRs one
                                           // attention workgroup leaves per SIMD; a CU's L2 fill rate saturates far below either)
  constexpr int PD0 = INFL / NT > 16 ? 16 : (INFL / NT < 1 ? 1 : INFL / NT);
  constexpr int PD = PD0 < S ? PD0 : S;
  static_assert(K % KS == 0 && N % 16 == 0, "layer shape");
  const int lc = lane & 15, g = lane >> 4;
  f32x4 acc[RB][NT];
  const char* wp[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int t = wave + FW * i;
    const int tt = (NTILES % FW == 0 || t < NTILES) ? t : 0;  // surplus tiles compute tile 0 again, never stored
    wp[i] = static_cast<const char*>(Wv) + (size_t)tt * S * 1024 + lane * 16;  // 1 KB per (tile, step group), lane-linear
  }
  u32x4f b[PD][NT];
#pragma unroll
  for (int p = 0; p < PD; ++p)
#pragma unroll
    for (int i = 0; i < NT; ++i) b[p][i] = *reinterpret_cast<const u32x4f*>(wp[i] + 1024 * p);
  // The order below is pinned with sched_barrier: left alone, the scheduler sinks every prefetch to just
  // before its use (register pressure) and the kernel runs with ~6 loads in flight per wave instead of PD*NT.
  __builtin_amdgcn_sched_barrier(0);
  const char* ap = actA + lc * apitch<F32>(K) + g * 16;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    u32x4f av[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) av[rb] = *reinterpret_cast<const u32x4f*>(ap + rb * 16 * apitch<F32>(K) + s * 64);
    if constexpr (F32) {
      // four exact-fp32 steps: component j of both pieces is sub-step j's operand value.  Sub-step outermost: consecutive
      // MFMAs go to DIFFERENT accumulators (v_mfma_f32_16x16x4_f32: 32 cycles of issue, 40 of dependent latency)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
            acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[rb][j]), __uint_as_float(b[s % PD][i][j]),
                                                              acc[rb][i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
          acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[rb]),
                                                               __builtin_bit_cast(bf16x8, b[s % PD][i]), acc[rb][i], 0, 0, 0);
    }
    if (s + PD < S) {
#pragma unroll
      for (int i = 0; i < NT; ++i) b[s % PD][i] = *reinterpret_cast<const u32x4f*>(wp[i] + 1024 * (s + PD));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // C/D layout of the 16x16 MFMAs (both): col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int t = wave + FW * i;
    if (NTILES % FW != 0 && t >= NTILES) continue;  // wave-uniform
    const int n = t * 16 + lc;
    const float bv = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb * 16 + 4 * g + r;
        float v = acc[rb][i][r] + bv;
        if (RELU) v = fmaxf(v, 0.f);
        if (MASK) {  // ReLU output > 0
          if constexpr (F32) {
            v = *reinterpret_cast<const float*>(mask + row * apitch<F32>(N) + n * ES) > 0.f ? v : 0.f;
          } else {
            const uint16_t m = *reinterpret_cast<const uint16_t*>(mask + row * apitch<F32>(N) + n * ES);
            v = (m != 0 && !(m & 0x8000u)) ? v : 0.f;
          }
        }
        if (act_out != nullptr) {
          if constexpr (F32) *reinterpret_cast<float*>(act_out + row * apitch<F32>(N) + n * ES) = v;
          else *reinterpret_cast<uint16_t*>(act_out + row * apitch<F32>(N) + n * ES) = bf16_of(v);
        }
        if (gout != nullptr) gout[(size_t)row * N + n] = v;
      }
  }
}

template <int C, bool PH2, int FR, bool F32 = false>
struct FusedLds {
  static constexpr int H = C / 2, R = C / 4;
  static constexpr int O_ENC = 0;
  static constexpr int O_H1 = O_ENC + FR * apitch<F32>(FE);   // h1, later dh1 in place
  static constexpr int O_DF = O_H1 + FR * apitch<F32>(H);     // d(pred)
  static constexpr int O_RAW = O_DF + FR * apitch<F32>(C);    // phase 2: raw rows, later d(Hres)
  static constexpr int O_R1 = O_RAW + FR * apitch<F32>(C);    // phase 2: r1
  static constexpr int O_R2 = O_R1 + FR * apitch<F32>(R);     // phase 2: r2, later dr2 in place
  static constexpr int TOTAL = PH2 ? O_R2 + FR * apitch<F32>(R) : O_RAW;
};
// the fp32-operand kernel exists where BOTH phases' images fit a CU's LDS at 16 rows per workgroup: C = 384, 768
template <int C>
constexpr bool fused_f32_fits() { return FusedLds<C, true, 16, true>::TOTAL <= 160 * 1024; }

template <int C, bool PH2, int FR, int FW, bool F32 = false>
__global__ __launch_bounds__(64 * FW) void fit_rows_kernel(FusedArgs a) {
  using L = FusedLds<C, PH2, FR, F32>;
  constexpr int RB = FR / 16;
  constexpr int ES = Op<F32>::ES;
  static_assert(FR == 16 || FR == 32, "rows per workgroup");
  static_assert(FW == 8 || (FW == 4 && FR == 16), "waves per workgroup");
  static_assert(!F32 || (FR == 16 && FW == 8), "fp32 operands: 16 rows, 8 waves");
  static_assert(L::TOTAL <= 160 * 1024, "LDS images of the row kernel");
  constexpr int H = C / 2, R = C / 4, E = FE, cq = C / 4;
  __shared__ __attribute__((aligned(16))) char smem[L::TOTAL];
  // Block -> (fit, row block).  Workgroups go round-robin to the 8 XCDs (private 4-MB L2s) and every workgroup streams its fit's
  // whole weight shadow (2.4-4.8 MB) from L2.  With k fits per launch and the plain (x, y) map every XCD hosts row blocks of
  // ALL k fits: 10-19 MB of weights through each 4-MB L2 -- they come from the memory side instead.  Round 6: fit f owns the
  // XCDs [f * 8 / k, (f + 1) * 8 / k) (k = 2, 4): an L2 sees ONE fit's weights, 16-32 workgroups re-use every line.  Placement
  // (workgroup b on XCD b % 8) is an assumption for speed only; the map is a bijection whatever the placement.
  int fit_, rb_;
  if (a.xcd_group > 0) {
    const int lin = (int)blockIdx.x + (int)blockIdx.y * (int)gridDim.x, xcd = lin & 7, j = lin >> 3;
    fit_ = xcd / a.xcd_group;
    rb_ = j * a.xcd_group + xcd % a.xcd_group;
  } else {
    fit_ = (int)blockIdx.y;
    rb_ = (int)blockIdx.x;
  }
  const DvtFusedFit& f = a.f[fit_];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = rb_ * FR;
  const char* __restrict__ sh = static_cast<const char*>(f.shadow);  // byte pointer: element offsets below are scaled by ES
  const float* __restrict__ P = f.params;
  const float4* __restrict__ feat4 = reinterpret_cast<const float4*>(f.feat);

  // ---- cooperative L2 warm-up.  Every launch starts with a cold L2 (the previous kernel's write-back /
  // invalidate), and the 16 workgroups that share an XCD walk the SAME weight lines in lock step, so
  // without this every one of them pays the fabric latency for every line.  Here each workgroup touches
  // a different 1/16 of the shadow weights once (one dword per 128-B line); the loads stay in flight
  // under the prologue and whatever the k-loops read afterwards is an L2 hit.  (Placement assumption --
  // workgroup b runs on XCD b % 8 -- is for speed only.)
  uint32_t warm[5] = {0u, 0u, 0u, 0u, 0u};
  {  // the rows the loss stage will read much later: this wave's sampled feature rows (random HBM rows) and
     // their lattice rows of G, one dword per 128-B line, so that their latency hides under the forward pass
    constexpr int LPR = (C * 4 + 127) / 128;  // lines per row
    const int j = lane % 32;  // 2 rows per pass (lane / 32), <= 32 lines each (C <= 1024)
#pragma unroll
    for (int rr = lane / 32; rr < FR / FW; rr += 2) {
      if (j < LPR) {
        const int ri = f.ridx[row0 + wave + FW * rr];
        warm[4] ^= __float_as_uint(f.feat[(size_t)ri * C + j * 32]) ^
                   __float_as_uint(P[a.off_G + (size_t)(ri % a.lattice) * C + j * 32]);
      }
    }
  }
  {
    // (the workgroups of one XCD that walk the same fit's weights share the warm-up: 8 per fit and XCD with the plain map at 16
    // rows, gridDim.x / xcd_group with the fit -> XCD map)
    const int per_xcd = a.xcd_group > 0 ? (int)gridDim.x / a.xcd_group : ((int)gridDim.x + 7) / 8;
    const int nslot = per_xcd >= 16 ? 16 : per_xcd;
    const int slot = (a.xcd_group > 0 ? rb_ / a.xcd_group : (rb_ >> 3)) % nslot;
    const long long lines = ((PH2 ? a.S.total : a.S.direct[2]) * ES + 127) / 128;
    const long long per = (lines + nslot - 1) / nslot;
    const long long l0 = slot * per, l1 = l0 + per < lines ? l0 + per : lines;
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(sh);
#pragma unroll
    for (int j = 0; j < 4 * RB * (8 / FW); ++j) {
      const long long li = l0 + tid + (long long)j * 64 * FW;
      if (li < l1) warm[j & 3] ^= w32[li * 32];
    }
  }

  // ---- prologue: hash-grid forward of the 16 rows (one (row, level) pair per thread, tcnn semantics in
  //      dvt_grid.hip) and, in phase 2, the raw rows as the residual predictor's input
  if (tid < FR * 16) {
    const int r = tid >> 4, l = tid & 15;
    const float2 p = reinterpret_cast<const float2*>(f.xy)[f.ridx[row0 + r]];
    uint32_t idx[4];
    float w[4];
    corners2d(a.T, l, p.x, p.y, idx, w);
    const float4* __restrict__ grid = reinterpret_cast<const float4*>(P + a.off_grid);
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 u = grid[(size_t)idx[c] * 2], v = grid[(size_t)idx[c] * 2 + 1];
      lo.x = fmaf(w[c], u.x, lo.x);
      lo.y = fmaf(w[c], u.y, lo.y);
      lo.z = fmaf(w[c], u.z, lo.z);
      lo.w = fmaf(w[c], u.w, lo.w);
      hi.x = fmaf(w[c], v.x, hi.x);
      hi.y = fmaf(w[c], v.y, hi.y);
      hi.z = fmaf(w[c], v.z, hi.z);
      hi.w = fmaf(w[c], v.w, hi.w);
    }
    if constexpr (F32) {
      *reinterpret_cast<float4*>(smem + L::O_ENC + r * apitch<F32>(E) + l * 32) = lo;
      *reinterpret_cast<float4*>(smem + L::O_ENC + r * apitch<F32>(E) + l * 32 + 16) = hi;
    } else {
      *reinterpret_cast<uint4*>(smem + L::O_ENC + r * apitch<F32>(E) + l * 16) =
          make_uint4(dvt_pack_bf16x2(lo.x, lo.y), dvt_pack_bf16x2(lo.z, lo.w), dvt_pack_bf16x2(hi.x, hi.y),
                     dvt_pack_bf16x2(hi.z, hi.w));
    }
  }
  if (PH2) {
    for (int i = tid; i < FR * cq; i += 64 * FW) {
      const int r = i / cq, q = i - r * cq;
      const float4 v = feat4[(size_t)f.ridx[row0 + r] * cq + q];
      if constexpr (F32)
        *reinterpret_cast<float4*>(smem + L::O_RAW + r * apitch<F32>(C) + q * 16) = v;
      else
        *reinterpret_cast<uint2*>(smem + L::O_RAW + r * apitch<F32>(C) + q * 8) =
            make_uint2(dvt_pack_bf16x2(v.x, v.y), dvt_pack_bf16x2(v.z, v.w));
    }
  }
  __syncthreads();

  // ---- forward: field MLP (neural_feature_field.py:40-44, :49), residual predictor (offline_denoiser.py:107)
  const int B = a.n;
  char* __restrict__ T = static_cast<char*>(f.T);  // byte pointer, like `sh`
#define SH(off) (sh + (size_t)(off) * ES)
#define TT(idx) (T + (size_t)a.TL.off[idx] * ES)
  mlp_layer<E, H, true, false, RB, FW, F32>(smem + L::O_ENC, SH(a.S.direct[0]), P + a.off_b1, smem + L::O_H1, nullptr,
                               nullptr, wave, lane);
  if (PH2)
    mlp_layer<C, R, true, false, RB, FW, F32>(smem + L::O_RAW, SH(a.S.direct[2]), P + a.off_bh1, smem + L::O_R1, nullptr,
                                 nullptr, wave, lane);
  __syncthreads();
  mlp_layer<H, C, false, false, RB, FW, F32>(smem + L::O_H1, SH(a.S.direct[1]), P + a.off_b2, nullptr,
                                f.F + (size_t)row0 * C, nullptr, wave, lane);
  // operands of the weight gradients leave as transposed bf16 copies while their LDS images are stable
  store_T<H, FR, FW, F32>(smem + L::O_H1, TT(DVT_T_H1), B, row0, tid);
  store_T<E, FR, FW, F32>(smem + L::O_ENC, TT(DVT_T_ENC), B, row0, tid);
  if (PH2) {
    mlp_layer<R, R, true, false, RB, FW, F32>(smem + L::O_R1, SH(a.S.direct[3]), P + a.off_bh2, smem + L::O_R2, nullptr,
                                 nullptr, wave, lane);
    store_T<C, FR, FW, F32>(smem + L::O_RAW, TT(DVT_T_RAW), B, row0, tid);
    store_T<R, FR, FW, F32>(smem + L::O_R1, TT(DVT_T_R1), B, row0, tid);
    __syncthreads();
    mlp_layer<R, C, false, false, RB, FW, F32>(smem + L::O_R2, SH(a.S.direct[4]), P + a.off_bh3, nullptr,
                                  f.Hres + (size_t)row0 * C, nullptr, wave, lane);
    store_T<R, FR, FW, F32>(smem + L::O_R2, TT(DVT_T_R2), B, row0, tid);
  }
  __syncthreads();  // F (and Hres) rows of this workgroup are visible to all of its waves

  // ---- loss + gradients (offline_denoiser.py:113-140), one wave per row, two rows per wave; the
  //      gradient of G is gathered inside Adam from the d(pred) rows written here
  static_assert(C <= 1024, "row warm-up above assumes <= 32 lines per row");
  {
    constexpr int NR = FR / FW;
    constexpr int NB = (C <= 768 && !(FW == 4 && PH2)) ? 2 : 1;  // rows whose loads are in flight together (register budget)
    DvtLossRowRegs<PH2> lr[NB];
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += NB) {
#pragma unroll
    for (int rr = r0; rr < r0 + NB; ++rr) {  // the rows' loads in flight before the first reduction
      const int gr = row0 + wave + FW * rr;
      const int ri = f.ridx[gr];
      dvt_loss_row_load<PH2>(lr[rr - r0], reinterpret_cast<const float4*>(f.F) + (size_t)gr * cq,
                             reinterpret_cast<const float4*>(P + a.off_G) + (size_t)(ri % a.lattice) * cq,
                             PH2 ? reinterpret_cast<const float4*>(f.Hres) + (size_t)gr * cq : nullptr,
                             feat4 + (size_t)ri * cq, cq, lane);
    }
#pragma unroll
    for (int rr = r0; rr < r0 + NB; ++rr) {
      const int row = wave + FW * rr, gr = row0 + row;
      char* const img_p = smem + L::O_DF + row * apitch<F32>(C);                     // d(pred) image of this row ...
      char* const img_h = PH2 ? smem + L::O_RAW + row * apitch<F32>(C) : nullptr;  // ... d(Hres) over the raw row
      dvt_loss_row_compute<PH2>(lr[rr - r0], reinterpret_cast<float4*>(f.dF) + (size_t)gr * cq,
                                nullptr, nullptr,
                                f.rows + (size_t)gr * 8, a.n, cq, a.grad_scale, lane,
                                F32 ? nullptr : reinterpret_cast<uint2*>(img_p), F32 ? nullptr : reinterpret_cast<uint2*>(img_h),
                                F32 ? reinterpret_cast<float4*>(img_p) : nullptr, F32 ? reinterpret_cast<float4*>(img_h) : nullptr);
    }
    }
  }
  __syncthreads();

  // ---- data gradients: dh1 = (dF . W2) * (h1 > 0), denc = dh1 . W1; dr2 = (dH . Wh3) * (r2 > 0),
  //      dr1 = (dr2 . Wh2) * (r1 > 0)  (the [K][N] shadow copies make these k-contiguous as well)
  mlp_layer<C, H, false, true, RB, FW, F32>(smem + L::O_DF, SH(a.S.transp[1]), nullptr, smem + L::O_H1, nullptr,
                               smem + L::O_H1, wave, lane);
  store_T<C, FR, FW, F32>(smem + L::O_DF, TT(DVT_T_DF), B, row0, tid);
  if (PH2) {
    mlp_layer<C, R, false, true, RB, FW, F32>(smem + L::O_RAW, SH(a.S.transp[4]), nullptr, smem + L::O_R2, nullptr,
                                 smem + L::O_R2, wave, lane);
    store_T<C, FR, FW, F32>(smem + L::O_RAW, TT(DVT_T_DH), B, row0, tid);
  }
  __syncthreads();
  mlp_layer<H, E, false, false, RB, FW, F32>(smem + L::O_H1, SH(a.S.transp[0]), nullptr, nullptr,
                                f.denc + (size_t)row0 * E, nullptr, wave, lane);
  store_T<H, FR, FW, F32>(smem + L::O_H1, TT(DVT_T_DH1), B, row0, tid);
  if (PH2) {
    mlp_layer<R, R, false, true, RB, FW, F32>(smem + L::O_R2, SH(a.S.transp[3]), nullptr, smem + L::O_R1, nullptr,
                                 smem + L::O_R1, wave, lane);
    store_T<R, FR, FW, F32>(smem + L::O_R2, TT(DVT_T_DR2), B, row0, tid);
    __syncthreads();
    store_T<R, FR, FW, F32>(smem + L::O_R1, TT(DVT_T_DR1), B, row0, tid);
  }
#undef SH
#undef TT
  // keeps the warm-up loads alive (a.n is never negative)
  if (a.n < 0) f.rows[tid] = __uint_as_float(warm[0] ^ warm[1] ^ warm[2] ^ warm[3] ^ warm[4]);
}

// one float4 of the arena per thread -> its bf16 shadow copies
__global__ __launch_bounds__(256) void shadow_build_kernel(DvtShadowLayout L, const float* const p0,
                                                           const float* const p1, const float* const p2,
                                                           const float* const p3, void* s0, void* s1,
                                                           void* s2, void* s3, long long q_lo, long long q_hi) {
  const float* p = blockIdx.y == 0 ? p0 : (blockIdx.y == 1 ? p1 : (blockIdx.y == 2 ? p2 : p3));
  void* sh = blockIdx.y == 0 ? s0 : (blockIdx.y == 1 ? s1 : (blockIdx.y == 2 ? s2 : s3));
  const long long q = q_lo + (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= q_hi) return;
  dvt_shadow_store(L, sh, q * 4, reinterpret_cast<const float4*>(p)[q]);
}


// ======================================================================================================
// Weight gradients of the step: dW[m][n] += sum_b dY[b][m] * X[b][n] for every layer, one launch.
// Both operands arrive as [cols][batch] fragment-major copies (store_T above; bf16, or fp32 in the fp32-operand mode), so A
// and B fragments of v_mfma_f32_16x16x32_bf16 (four steps of v_mfma_f32_16x16x4_f32) are single coalesced 16-byte-per-lane
// loads, straight to VGPRs: every WAVE owns
// a 32 x 32 block of one dW and one quarter of the batch (16 loads in flight); the four quarter sums of a
// block meet in LDS and ONE wave stores the total with plain stores into the zeroed gradient arena -- no
// atomics, deterministic sums (see wgrad_block; round 1 used 4 fp32 atomics per element).
// Bias gradients (column sums of dY, i.e. of the A operand) ride along in the waves of the first n-block.
// (Round 3 tried 64 x 64 blocks per wave in their own launch: correct, +1 % at 4 concurrent fits, -9 % for one
// fit; profiles/r03/r03d_wgrad64_experiment.txt.)
// ======================================================================================================
constexpr int WG_MAX_PROB = 5 * DVT_FIT_BATCH_MAX;
struct WgradProb {
  const char* AT;  // dY^T [M][B]: fragment-major, 1 KB per (tile, step group) in either element type
  const char* BT;  // X^T  [N][B]
  float* dW;           // [M][N] fp32, += (atomics)
  float* db;           // [M] or nullptr
  int M, N;
};
struct WgradGather {  // gradient of the shared-artifact map G: row r = sum of the d(pred) rows of its samples
  const int32_t* offs;  // [lattice + 1]
  const uint16_t* perm; // [B], sorted inside a list (deterministic sums)
  const float4* rows;   // d(pred) [B][C] fp32
  float4* dG;           // gradient arena at G, [lattice][C]
};
struct WgradArgs {
  int n_prob, B, ksplit, units_total;
  int n_gather, lattice, cq;  // fits with a G gradient this step (0 in phase 2), rows of G, float4 per row
  WgradGather gg[DVT_FIT_BATCH_MAX];
  int unit0[WG_MAX_PROB + 1];
  WgradProb p[WG_MAX_PROB];
};

// One 1024-thread block = 4 output blocks x 4 batch quarters: wave w works on unit (4 * blk + w / 4), quarter
// w % 4; the four partial 32 x 32 sums of a unit meet in LDS and ONE wave stores the total with plain stores
// (the gradient arena is zero between steps) -- no atomics, deterministic sums.  (With fp32 atomics, 4 per
// element, this half and the grid backward were both bound by the L2 atomic rate: side by side in one launch
// they took exactly the sum of their stand-alone times.)
constexpr int WG_PART_FLOATS = 32 * 32 + 32;  // a wave's partial block + its bias partials
template <int SLOTS, bool F32>  // units per workgroup: 4 (16 waves) or 2 (8 waves, dvt_tune_set(14, 1)); operand element type
__device__ __forceinline__ void wgrad_block(const WgradArgs& a, int blk, int wave, int lane, float* red) {
  constexpr int PD = 4;
  const int slot = wave >> 2, ks = wave & 3;
  const int unit = blk * SLOTS + slot;
  float* mine = red + (slot * 4 + ks) * WG_PART_FLOATS;
  const bool is_w = unit < a.units_total;
  const int gu = unit - a.units_total;  // gradient-of-G rows ride in the surplus units (4 rows per unit)
  float* out_w = nullptr;  // (plain values, not a pointer into the kernel arguments: that forces a stack copy)
  float* out_b = nullptr;
  int out_n = 0;
  int mb = 0, nb = 0;
  bool do_bias = false;
  if (is_w) {
    int pi = 0;
#pragma unroll
    for (int j = 1; j < WG_MAX_PROB; ++j)
      if (j < a.n_prob && unit >= a.unit0[j]) pi = j;
    const WgradProb& p = a.p[pi];
    out_w = p.dW;
    out_b = p.db;
    out_n = p.N;
    const int local = unit - a.unit0[pi], nbn = p.N >> 5;
    mb = local / nbn;
    nb = local - mb * nbn;
    const int ksteps = a.B / Op<F32>::KS, S = ksteps >> 2, s_begin = ks * S;  // step groups of the batch; S % PD == 0 (host)
    const char* ap[2];
    const char* bp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ap[i] = p.AT + ((size_t)(mb * 2 + i) * ksteps + s_begin) * 1024 + lane * 16;
      bp[i] = p.BT + ((size_t)(nb * 2 + i) * ksteps + s_begin) * 1024 + lane * 16;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    do_bias = p.db != nullptr && nb == 0;  // wave-uniform
    float bsum[2] = {0.f, 0.f};
    u32x4f fa[PD][2], fb[PD][2];
#pragma unroll
    for (int q = 0; q < PD; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[q][i] = *reinterpret_cast<const u32x4f*>(ap[i] + 1024 * q);
        fb[q][i] = *reinterpret_cast<const u32x4f*>(bp[i] + 1024 * q);
      }
    __builtin_amdgcn_sched_barrier(0);
    for (int s0 = 0; s0 < S; s0 += PD) {
#pragma unroll
      for (int q = 0; q < PD; ++q) {
        if constexpr (F32) {  // sub-step outermost: four independent accumulators in rotation (see mlp_layer)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fa[q][i][e]), __uint_as_float(fb[q][j][e]),
                                                                 acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[q][i]),
                                                                  __builtin_bit_cast(bf16x8, fb[q][j]), acc[i][j], 0, 0, 0);
        }
        if (do_bias) {  // column sums of dY: every batch row of the step group sits in exactly one lane group's piece
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if constexpr (F32) {
#pragma unroll
              for (int e = 0; e < 4; ++e) bsum[i] += __uint_as_float(fa[q][i][e]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {  // (element order as in rounds 2-4: the sums stay bit-identical)
                bsum[i] += __uint_as_float(fa[q][i][e] << 16);
                bsum[i] += __uint_as_float(fa[q][i][e] & 0xffff0000u);
              }
            }
          }
        }
        if (s0 + q + PD < S) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            fa[q][i] = *reinterpret_cast<const u32x4f*>(ap[i] + 1024 * (s0 + q + PD));
            fb[q][i] = *reinterpret_cast<const u32x4f*>(bp[i] + 1024 * (s0 + q + PD));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // partial block -> LDS, element (i, j, r) of lane l at ((i * 2 + j) * 4 + r) * 64 + l (conflict-free)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[((i * 2 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {  // lane (lc, g) holds 8 of a k-step's 32 batch rows for column lc: fold the 4 groups
        float v = bsum[i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16) mine[1024 + i * 16 + lane] = v;
      }
    }
  }
  __syncthreads();
  if (is_w) {
    if (ks != 0) return;
    const float* base = red + slot * 4 * WG_PART_FLOATS;
    // D[row = m][col = n]: col = lane & 15, row = 4 * (lane >> 4) + r
    const int lc = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = ((i * 2 + j) * 4 + r) * 64 + lane;
          const float v = (base[o] + base[WG_PART_FLOATS + o]) + (base[2 * WG_PART_FLOATS + o] + base[3 * WG_PART_FLOATS + o]);
          out_w[(size_t)(mb * 32 + i * 16 + 4 * g + r) * out_n + nb * 32 + j * 16 + lc] = v;
        }
    if (do_bias && lane < 32) {
      const int o = 1024 + lane;
      out_b[mb * 32 + lane] = (base[o] + base[WG_PART_FLOATS + o]) + (base[2 * WG_PART_FLOATS + o] + base[3 * WG_PART_FLOATS + o]);
    }
    return;
  }
  // ---- surplus units: the gradient of G, one wave per lattice row (lists average 1.5 samples).  Plain stores
  // into the zeroed gradient arena; Adam then reads G like any dense-gradient tensor -- which keeps the
  // gather's registers out of the Adam kernel (see adam_kernel).
  const int u = gu * 4 + ks;
  if (u >= a.n_gather * a.lattice) return;
  const int fi = u / a.lattice, r = u - fi * a.lattice;
  const WgradGather& gg = a.gg[fi];
  const int o0 = gg.offs[r], o1 = gg.offs[r + 1];
  if (o1 == o0) return;
  for (int q = lane; q < a.cq; q += 64) {
    float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int o = o0; o < o1; ++o) {
      const float4 d = gg.rows[(size_t)gg.perm[o] * a.cq + q];
      acc4.x += d.x;
      acc4.y += d.y;
      acc4.z += d.z;
      acc4.w += d.w;
    }
    gg.dG[(size_t)r * a.cq + q] = acc4;
  }
}

// ONE launch for the two independent halves of the backward pass that reduce over rows: the hash-grid
// backward (blocks [0, grid_blocks): LDS-accumulated coarse levels + atomics for the fine ones, dvt_grid.hip)
// and the weight gradients (the rest: 16 independent waves per block).  Back to back they cost 27 + 17 us and
// a launch boundary; side by side the longer one.
struct BackwardArgs {
  DvtGridTable T;
  GridBwdPlan plan;
  GridBwdPtrs gp;
  GridSortedPtrs gs;  // gs.nt > 0: gather from the sorted lists instead of scattering with atomics
  int n, grid_blocks_per_fit, k, wg_blocks;
  WgradArgs w;
};
template <int WAVES, bool F32 = false>
__global__ __launch_bounds__(64 * WAVES) void fit_backward_kernel(BackwardArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[WAVES * WG_PART_FLOATS];  // 66 KB at 16 waves: grid half uses the first 8.2 KB
  // weight-gradient blocks FIRST: the grid half alone is more blocks than the chip holds at once, behind
  // it the other half would only start when it drains (measured: the sum of the two, not the maximum)
  if ((int)blockIdx.x >= a.wg_blocks) {
    const int b = (int)blockIdx.x - a.wg_blocks;
    const int fy = b / a.grid_blocks_per_fit, bx = b - fy * a.grid_blocks_per_fit;
    if (a.gs.nt > 0) {
      const int parts = a.gs.nt / (64 * WAVES);
      grid_gather_body(a.T, a.gs, fy, bx / parts, bx % parts, a.gp.d_enc[fy], a.gp.d_params[fy], a.gp.touched[fy], 64 * WAVES);
      return;
    }
    if constexpr (WAVES == 16)  // (the scatter path is written for 1024 threads; the host only picks 8 waves with sorted lists)
      grid_bwd_body<256>(a.T, a.plan, a.gp, a.n, bx, fy, smem, reinterpret_cast<uint32_t*>(smem + 256 * 8));
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  wgrad_block<WAVES / 4, F32>(a.w, (int)blockIdx.x, wave, threadIdx.x & 63, smem);
}

// FR = 32 rows per workgroup halves the weight stream per row (every fragment feeds two MFMAs) and the number of
// workgroups.  Measured (tools/bench_fit_batch.py, C = 768, us per step and fit, 16 -> 32 rows): one fit 96.7 -> 107.8
// (latency-bound on 128 workgroups, worse on 64), two fits 74.1 -> 79.1 (128 workgroups leave half the chip empty),
// four fits 71.8 -> 66.1, eight 66.9 -> 63.5.  So: 32 rows when k >= 4 fits share the launch and the LDS images fit
// (158.7 KB at C = 768 in phase 2; C = 1024 only in phase 1) -- dvt_tune_set(13, v): 1 auto, 0 = always 16, 2 = 32
// wherever it fits (the parity tests run both).
template <int C>
int launch_rows(const FusedArgs& a, int k, bool phase2, hipStream_t s) {
  if (a.S.f32) {  // fp32 operands: 16 rows, 8 waves (the shapes the LDS images fit: dvt_fit_fused_ok)
    if constexpr (fused_f32_fits<C>()) {
      dim3 grid(a.n / 16, k), block(64 * FW8);
      if (phase2) hipLaunchKernelGGL((fit_rows_kernel<C, true, 16, FW8, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((fit_rows_kernel<C, false, 16, FW8, true>), grid, block, 0, s, a);
      DVT_CHECK_LAUNCH();
      return 0;
    }
    return DVT_E_BADARG;
  }
  const bool fits32 = phase2 ? (FusedLds<C, true, 32>::TOTAL <= 160 * 1024) : (FusedLds<C, false, 32>::TOTAL <= 160 * 1024);
  const bool r32 = fits32 && a.n % 32 == 0 && (g_fit_rows32 == 2 || (g_fit_rows32 == 1 && k >= 4));
  const bool small = g_fit_small_wg && !r32;
  dim3 grid(a.n / (r32 ? 32 : 16), k), block(64 * (small ? 4 : FW8));
  FusedArgs ax = a;  // (the fit -> XCD map needs whole groups of 8 / k row blocks)
  ax.xcd_group = (g_fit_xcd_affinity && (k == 2 || k == 4) && grid.x % (8 / k) == 0 && (grid.x * k) % 8 == 0) ? 8 / k : 0;
  const FusedArgs& a_ = ax;
  if (r32) {
    if (phase2) {
      if constexpr (FusedLds<C, true, 32>::TOTAL <= 160 * 1024) hipLaunchKernelGGL((fit_rows_kernel<C, true, 32, FW8>), grid, block, 0, s, a_);
    } else {
      if constexpr (FusedLds<C, false, 32>::TOTAL <= 160 * 1024) hipLaunchKernelGGL((fit_rows_kernel<C, false, 32, FW8>), grid, block, 0, s, a_);
    }
  } else if (small) {
    if (phase2) hipLaunchKernelGGL((fit_rows_kernel<C, true, 16, 4>), grid, block, 0, s, a_);
    else hipLaunchKernelGGL((fit_rows_kernel<C, false, 16, 4>), grid, block, 0, s, a_);
  } else if (phase2) {
    hipLaunchKernelGGL((fit_rows_kernel<C, true, 16, FW8>), grid, block, 0, s, a_);
  } else {
    hipLaunchKernelGGL((fit_rows_kernel<C, false, 16, FW8>), grid, block, 0, s, a_);
  }
  DVT_CHECK_LAUNCH();
  return 0;
}

}  // namespace

int dvt_shadow_layout(const DvtFitConfig* c, DvtShadowLayout* L) {
  if (!c || !L) return DVT_E_BADARG;
  const int C = c->feat_dim, H = c->hidden, R = c->res_hidden;
  const int E = c->grid.n_levels * c->grid.n_features;
  const int Ns[DVT_SHADOW_MATS] = {H, C, R, R, C}, Ks[DVT_SHADOW_MATS] = {E, H, C, R, R};
  const long long begins[DVT_SHADOW_MATS] = {c->off_w1, c->off_w2, c->off_wh1, c->off_wh2, c->off_wh3};
  const bool tr[DVT_SHADOW_MATS] = {true, true, false, true, true};  // no data gradient flows into `raw`
  long long o = 0;
  L->n = DVT_SHADOW_MATS;
  L->f32 = c->mlp_bf16 ? 0 : 1;  // element type of the copies = the operand precision of the step
  L->lo = begins[0];
  L->hi = 0;
  for (int i = 0; i < DVT_SHADOW_MATS; ++i) {
    if (Ks[i] % 32 || Ns[i] % 16 || (tr[i] && (Ns[i] % 32 || Ks[i] % 16))) return DVT_E_BADARG;
    L->N[i] = Ns[i];
    L->K[i] = Ks[i];
    L->begin[i] = begins[i];
    const long long sz = (long long)Ns[i] * Ks[i];
    L->direct[i] = o;
    o += sz;
    L->transp[i] = tr[i] ? o : -1;
    if (tr[i]) o += sz;
    if (begins[i] < L->lo) L->lo = begins[i];
    if (begins[i] + sz > L->hi) L->hi = begins[i] + sz;
  }
  L->total = o;
  return 0;
}

int dvt_shadow_build_k(const DvtShadowLayout* L, int k, const float* const* params, void* const* shadow,
                       long long lo, long long hi, hipStream_t s) {
  if (!L || k < 1 || k > DVT_FIT_BATCH_MAX || L->n <= 0) return DVT_E_BADARG;
  const float* p[4] = {nullptr, nullptr, nullptr, nullptr};
  void* sh[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int f = 0; f < k; ++f) {
    if (!params[f] || !shadow[f]) return DVT_E_BADARG;
    p[f] = params[f];
    sh[f] = shadow[f];
  }
  const long long q_lo = (lo > L->lo ? lo : L->lo) / 4, q_hi = (hi < L->hi ? hi : L->hi) / 4;
  if (q_hi <= q_lo) return 0;
  hipLaunchKernelGGL(shadow_build_kernel, dim3(dvt_cdiv(q_hi - q_lo, 256), k), dim3(256), 0, s, *L, p[0], p[1],
                     p[2], p[3], sh[0], sh[1], sh[2], sh[3], q_lo, q_hi);
  DVT_CHECK_LAUNCH();
  return 0;
}

int g_fit_fused_enable = 1;  // dvt_tune_set(6, 0): the unfused launch sequence (same results, A/B timing + parity)
int g_fit_sorted_grid = 1;  // 0: the fused step scatters the grid gradient with atomics (round-2a path)

int g_fit_fused_f32 = 1;  // dvt_tune_set(6, 2 / 3): fp32-operand fused step off / on (A/B against the layer-by-layer launches)

bool dvt_fit_fused_ok(const DvtFitConfig* c) {
  if (!g_fit_fused_enable || !c || !dvt_fit_fused_shapes_ok(c)) return false;
  if (c->mlp_bf16) return true;
  // fp32 operands (round 5): where the fp32 LDS images of both phases fit a CU at 16 rows per workgroup
  return g_fit_fused_f32 && (c->feat_dim == 384 || c->feat_dim == 768);
}

bool dvt_fit_fused_shapes_ok(const DvtFitConfig* c) {
  if (!c) return false;
  if (c->grid.n_levels != 16 || c->grid.n_features != 8) return false;
  const int C = c->feat_dim;
  if (C != 384 && C != 768 && C != 1024) return false;
  if (c->hidden != C / 2 || c->res_hidden != C / 4) return false;
  // the weight-gradient kernel splits the batch into 4 quarters of whole 4-deep prefetch groups of 32-row k-steps
  if (c->batch % 512 || c->batch <= 0) return false;
  return c->lattice <= 8192 && c->batch <= 65535;  // G gradient through Adam's row lists
}

int dvt_fit_rows_k(const DvtFitConfig* c, const DvtShadowLayout* L, int k, const DvtFusedFit* fits, bool phase2,
                   hipStream_t s) {
  if (!dvt_fit_fused_ok(c) || !L || !fits || k < 1 || k > DVT_FIT_BATCH_MAX) return DVT_E_BADARG;
  FusedArgs a{};
  a.T = c->grid;
  a.S = *L;
  dvt_t_layout(c, &a.TL);
  a.n = c->batch;
  a.lattice = c->lattice;
  a.grad_scale = (float)c->grad_scale;
  a.off_grid = c->off_grid;
  a.off_b1 = c->off_b1;
  a.off_b2 = c->off_b2;
  a.off_G = c->off_G;
  a.off_bh1 = c->off_bh1;
  a.off_bh2 = c->off_bh2;
  a.off_bh3 = c->off_bh3;
  for (int f = 0; f < k; ++f) a.f[f] = fits[f];
  const double C = c->feat_dim, H = c->hidden, R = c->res_hidden, E = FE, B = c->batch;
  // forward + dgrad flops of the row chain (the wgrad half of the step runs in the grouped GEMM launch)
  const double flops = 2.0 * B * (2.0 * (E * H + H * C) + (phase2 ? (C * R + 2.0 * R * R + 2.0 * R * C) : 0.0));
  DvtProbeScope probe(DVT_PROBE_FIT_ROWS, s, flops * k);
  switch (c->feat_dim) {
    case 384: return launch_rows<384>(a, k, phase2, s);
    case 768: return launch_rows<768>(a, k, phase2, s);
    default: return launch_rows<1024>(a, k, phase2, s);
  }
}

void dvt_t_layout(const DvtFitConfig* c, DvtTLayout* L) {
  const int C = c->feat_dim, H = c->hidden, R = c->res_hidden, E = c->grid.n_levels * c->grid.n_features;
  const int cols[DVT_T_COUNT] = {C, H, H, E, C, R, R, C, R, R};
  long long o = 0;
  for (int i = 0; i < DVT_T_COUNT; ++i) {
    L->off[i] = o;
    L->cols[i] = cols[i];
    o += (long long)cols[i] * c->batch;
  }
  L->total = o;
}

int dvt_fit_backward_k(const DvtFitConfig* c, int k, const DvtFusedFit* fits, bool phase2, hipStream_t s) {
  if (!dvt_fit_fused_ok(c) || !fits || k < 1 || k > DVT_FIT_BATCH_MAX) return DVT_E_BADARG;
  DvtTLayout TL;
  dvt_t_layout(c, &TL);
  const int C = c->feat_dim, H = c->hidden, R = c->res_hidden, E = c->grid.n_levels * c->grid.n_features;
  BackwardArgs ba{};
  WgradArgs& a = ba.w;
  a.B = c->batch;
  const int ksteps = a.B / 32;
  a.ksplit = 4;  // batch quarters = the 4 waves of a unit; S = ksteps / 4 must be a multiple of the prefetch depth 4
  if (ksteps % 16) return DVT_E_BADARG;  // (fp32 operands: step groups of 16 rows, twice as many: the same condition)
  int units = 0;
  double flops = 0.0;
  const bool f32 = !c->mlp_bf16;
  const size_t es = f32 ? 4 : 2;  // bytes per element of the transposed operand copies
  auto add = [&](const DvtFusedFit& f, int tA, int tB, long long ow, long long ob, int M, int N) {
    WgradProb& p = a.p[a.n_prob];
    p.AT = static_cast<const char*>(f.T) + (size_t)TL.off[tA] * es;
    p.BT = static_cast<const char*>(f.T) + (size_t)TL.off[tB] * es;
    p.dW = f.grads + ow;
    p.db = f.grads + ob;
    p.M = M;
    p.N = N;
    a.unit0[a.n_prob++] = units;
    units += (M / 32) * (N / 32);
    flops += 2.0 * M * N * a.B;
  };
  for (int f = 0; f < k; ++f) {
    add(fits[f], DVT_T_DF, DVT_T_H1, c->off_w2, c->off_b2, C, H);    // dW2 = dF^T . h1
    add(fits[f], DVT_T_DH1, DVT_T_ENC, c->off_w1, c->off_b1, H, E);  // dW1 = dh1^T . enc
    if (phase2) {
      add(fits[f], DVT_T_DH, DVT_T_R2, c->off_wh3, c->off_bh3, C, R);    // dWh3 = dH^T . r2
      add(fits[f], DVT_T_DR2, DVT_T_R1, c->off_wh2, c->off_bh2, R, R);   // dWh2 = dr2^T . r1
      add(fits[f], DVT_T_DR1, DVT_T_RAW, c->off_wh1, c->off_bh1, R, C);  // dWh1 = dr1^T . raw
    }
  }
  a.unit0[a.n_prob] = units;
  a.units_total = units;
  a.lattice = c->lattice;
  a.cq = C / 4;
  if (!phase2) {
    for (int f = 0; f < k; ++f) {
      if (!fits[f].g_offs || !fits[f].g_perm) return DVT_E_BADARG;
      a.gg[f] = WgradGather{fits[f].g_offs, fits[f].g_perm, reinterpret_cast<const float4*>(fits[f].dF),
                            reinterpret_cast<float4*>(fits[f].grads + c->off_G)};
    }
    a.n_gather = k;
  }
  // hash-grid backward half (same plan / body as dvt_grid_bwd_k, 256-entry LDS chunks)
  ba.T = c->grid;
  ba.n = c->batch;
  ba.k = k;
  const int saved_chunk = g_grid_lds_chunk;
  g_grid_lds_chunk = 256;
  dvt_grid_bwd_plan(c->grid, c->batch, &ba.plan);
  g_grid_lds_chunk = saved_chunk;
  const long long direct_threads = (long long)c->batch * (c->grid.n_levels - ba.plan.first_direct_level) * 8;
  ba.grid_blocks_per_fit = ba.plan.n_lds_blocks + dvt_cdiv(direct_threads, 1024);
  bool sorted = g_fit_sorted_grid && dvt_grid_sorted_ok(&c->grid, c->batch);
  for (int f = 0; f < k; ++f) sorted = sorted && fits[f].gs_keys && fits[f].gs_pay && fits[f].gs_w;
  const bool small = g_fit_small_wg && sorted && !f32;  // 8-wave workgroups, 2 units each
  if (sorted) {
    ba.gs.nt = 4 * c->batch;
    ba.gs.bitmap_end = fits[0].gs_bitmap_end;
    ba.grid_blocks_per_fit = c->grid.n_levels * (ba.gs.nt / (small ? 512 : 1024));
    for (int f = 0; f < k; ++f) {
      ba.gs.keys[f] = fits[f].gs_keys;
      ba.gs.pay[f] = fits[f].gs_pay;
      ba.gs.w[f] = fits[f].gs_w;
    }
  }
  for (int f = 0; f < k; ++f) {
    ba.gp.xy[f] = reinterpret_cast<const float2*>(fits[f].xy);
    ba.gp.ridx[f] = fits[f].ridx;
    ba.gp.d_enc[f] = fits[f].denc;
    ba.gp.d_params[f] = fits[f].grads + c->off_grid;
    ba.gp.touched[f] = fits[f].touched;
  }
  const int wg_blocks = dvt_cdiv((long long)units + dvt_cdiv((long long)a.n_gather * a.lattice, 4), small ? 2 : 4);
  ba.wg_blocks = wg_blocks;
  DvtProbeScope probe(DVT_PROBE_FIT_GEMM, s, flops);
  if (f32)
    hipLaunchKernelGGL((fit_backward_kernel<16, true>), dim3(ba.grid_blocks_per_fit * k + wg_blocks), dim3(1024), 0, s, ba);
  else if (small)
    hipLaunchKernelGGL(fit_backward_kernel<8>, dim3(ba.grid_blocks_per_fit * k + wg_blocks), dim3(512), 0, s, ba);
  else
    hipLaunchKernelGGL(fit_backward_kernel<16>, dim3(ba.grid_blocks_per_fit * k + wg_blocks), dim3(1024), 0, s, ba);
  DVT_CHECK_LAUNCH();
  return 0;
}
