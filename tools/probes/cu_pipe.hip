// Developer probe: what ONE workgroup per CU (the GEMM's occupancy: 8 waves, >80 KB of LDS) can pull from L2 and push to
// memory, by access pattern.  Answers two questions of profiles/r04: is the ~48 GB/s per CU that the GEMM's LDS-DMA stream,
// fit_rows and fit_backward all land on a property of the pattern (128-B row pieces at a 1.5-6 KB pitch) or of the CU; and
// would a tile-major (contiguous) layout of the intermediates make the epilogue's stores faster than 512-B row pieces at a
// 4.6-KB pitch.
//   build: hipcc --offload-arch=gfx950 -O3 cu_pipe.hip -o cu_pipe        run: ./cu_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                      \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int LDS_BYTES = 96 * 1024;  // one workgroup per CU

// ---- loads.  Every iteration a workgroup fetches PIECES x 8 KB: 512 threads x 16 B per piece.
//   ROWB   bytes of one contiguous row piece (128 = a GEMM operand row of one k-tile; 8192 = fully contiguous)
//   pitch  distance between consecutive row pieces
//   The region a workgroup walks is `span` bytes starting at (blockIdx % nreg) * span: with nreg * span <= a few MB every
//   fetch is an L2 hit after the first pass and never an L1 hit (32 KB L1, 8 KB per piece, span >> 32 KB).
template <bool DMA, int ROWB>
__global__ __launch_bounds__(512) void load_kernel(const char* __restrict__ src, size_t pitch, size_t span, int nreg,
                                                   int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int t = threadIdx.x;
  constexpr int TPR = ROWB / 16;  // threads per row piece
  const size_t lane_off = (size_t)(t / TPR) * pitch + (size_t)(t % TPR) * 16;
  const size_t piece = (size_t)(512 / TPR) * pitch;  // bytes of address space one 8-KB piece covers
  const char* base = src + (size_t)(blockIdx.x % nreg) * span;
  size_t pos = ((size_t)blockIdx.x * 7919u * piece) % span;  // de-phase the workgroups that share a region
  if (pos + piece > span) pos = 0;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const char* g = base + pos + lane_off;
      if constexpr (DMA) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (lds_ptr_t)(smem + ((it * 8 + p) % 8) * 8192 + (t >> 6) * 1024), 16, 0, 0);
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(g);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      }
      pos += piece;
      if (pos + piece > span) pos = 0;
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  if constexpr (DMA) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc.x = *reinterpret_cast<unsigned*>(smem + t * 16);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

// ---- stores.  A workgroup writes `tiles` output tiles of 256 rows x 512 B (a 256 x 256 bf16 GEMM tile), 16 B per lane.
//   BLOCKED = false: row-major [M][N]: row pieces of 512 B at pitch n_bytes (qkv: 4608, fc1: 6144)
//   BLOCKED = true:  the tile is one contiguous 128-KB burst
template <bool BLOCKED>
__global__ __launch_bounds__(512) void store_kernel(char* __restrict__ dst, size_t n_bytes, int n_tiles_n, int tiles_total) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int t = threadIdx.x;
  smem[t] = (char)t;
  __syncthreads();
  const uint4 v = make_uint4(t, smem[(t * 7) & 511], blockIdx.x, 42);
  for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
    const int tm = tile / n_tiles_n, tn = tile % n_tiles_n;
    char* base = BLOCKED ? dst + (size_t)tile * 131072 : dst + (size_t)tm * 256 * n_bytes + (size_t)tn * 512;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      char* p = BLOCKED ? base + it * 8192 + t * 16 : base + (size_t)(it * 16 + (t >> 5)) * n_bytes + (t & 31) * 16;
      *reinterpret_cast<uint4*>(p) = v;
    }
  }
}

// ---- the GEMM's operand stream, alone.  One workgroup = one 256 x 256 tile of y = x W^T (x [M][K], W [N][K] bf16, row-major):
// per k-tile (64 k) eight 8-KB pieces -- 4 of A (64 rows x 128 B at pitch 2K), 4 of W -- by LDS-DMA, D pieces in flight, nothing
// else (no MFMA, no LDS reads, no stores).  Tile order = dvt_vit.hip's map_tile (every XCD a contiguous run of the order
// (n group, block of `mblock` M panels, n, m in block)).  Knobs:
//   rot      0: every tile walks k = 0 .. nk-1 (the kernel today: the tiles that share an A panel / W slice ask for the same
//               lines at the same time); 1: tile (m, n) starts at k-tile ((n * mblock + m % mblock) * nk) / (group * mblock)
//               and wraps; 2: start = (n * nk) / group (only the sharers of an A panel are spread)
//   a_panels > 0: A is read from panels (m % a_panels) only (an L2-resident A: the upper bound with no first-touch misses)
struct TileMap { int m, n; };
__device__ __forceinline__ TileMap map_tile(int bid, int nwg, int mt, int nt, int group, int mblock) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int per_group = group * mt;
  int g = id / per_group;
  const int full = nt / group;
  int width = group;
  if (g >= full) { g = full; width = nt - full * group; }
  const int rem = id - g * per_group;
  TileMap t;
  if (mblock <= 1) { t.m = rem / width; t.n = g * group + rem % width; }
  else {
    const int per_block = mblock * width;
    const int mb = rem / per_block, r2 = rem - mb * per_block;
    const int left = mt - mb * mblock, hb = left < mblock ? left : mblock;
    t.n = g * group + r2 / hb;
    t.m = mb * mblock + r2 % hb;
  }
  return t;
}

template <int D>
__global__ __launch_bounds__(512) void gemm_stream_kernel(const char* __restrict__ A, const char* __restrict__ W, int mt, int nt,
                                                          int K, int group, int mblock, int rot, int a_panels) {
  __shared__ __attribute__((aligned(16))) char smem[(D + 1) * 8192 > LDS_BYTES ? (D + 1) * 8192 : LDS_BYTES];
  const int t = threadIdx.x;
  const TileMap tm = map_tile(blockIdx.x, gridDim.x, mt, nt, group, mblock);
  const int nk = K / 64;
  const size_t pitch = (size_t)K * 2;
  const int am = a_panels > 0 ? tm.m % a_panels : tm.m;
  const char* a0 = A + (size_t)am * 256 * pitch + (size_t)(t >> 3) * pitch + (t & 7) * 16;
  const char* w0 = W + (size_t)tm.n * 256 * pitch + (size_t)(t >> 3) * pitch + (t & 7) * 16;
  int kt = 0;
  if (rot == 1) kt = ((tm.n % group) * mblock + tm.m % mblock) * nk / (group * mblock);
  if (rot == 2) kt = (tm.n % group) * nk / group;
  int slot = 0;
  for (int s = 0; s < nk; ++s) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const char* g = ((p & 2) ? w0 : a0) + (size_t)((p >> 2) * 2 + (p & 1)) * 64 * pitch + (size_t)kt * 128;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (lds_ptr_t)(smem + slot * 8192 + (t >> 6) * 1024), 16, 0, 0);
      slot = slot == D ? 0 : slot + 1;
      if constexpr (D == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else if constexpr (D == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    }
    kt = kt + 1 == nk ? 0 : kt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- does LDS / LDS-DMA traffic of ONE wave slow the MFMAs of the OTHER wave on the same SIMD?  (the 8p GEMM's premise is that
// it does not: one wave group issues MFMAs while the other reads fragments and issues the DMA.)  512 threads: waves 0-3 (one per
// SIMD) run a pure MFMA loop on register operands and time it with s_memtime; waves 4-7 (their SIMD partners) run `partner`:
//   0 nothing, 1 ds_read_b128 stream (the GEMM's swizzled fragment pattern), 2 LDS-DMA stream from an L2-resident buffer,
//   3 both interleaved (12 reads : 2 DMA, the mix of a GEMM load segment), 4 the same with the MFMA waves at s_setprio 1
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mix_kernel(const char* __restrict__ src, int partner, int n_mfma_iters, int n_partner_iters,
                                                  unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < LDS_BYTES / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(i, i * 3, i * 5, i * 7);
  __syncthreads();
  if (wave < 4) {
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(smem + (lane + 64 * i) * 16);
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const bf16x8*>(smem + (lane + 64 * (4 + i)) * 16);
    f32x4 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (partner == 4) __builtin_amdgcn_s_setprio(1);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n_mfma_iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][3];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 4 + wave] = (t1 - t0) | (sum == 123.f ? 1ull << 63 : 0ull);
  } else if (partner != 0) {
    const int w = wave - 4;
    const int row = w * 16 + (lane & 15), cg = lane >> 4;
    const int o0 = row * 128 + (((0 + cg) ^ (row & 7)) << 4), o1 = row * 128 + (((4 + cg) ^ (row & 7)) << 4);
    const char* g = src + (size_t)(blockIdx.x & 7) * (2 << 20) + (size_t)t * 16;
    uint4 x = make_uint4(0, 0, 0, 0);
    const unsigned long long p0 = __builtin_readcyclecounter();
    for (int it = 0; it < n_partner_iters; ++it) {
      asm volatile("" ::: "memory");  // the reads below are loop-invariant otherwise
      if (partner == 1 || partner >= 3) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          uint4 v0, v1;
          asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"((unsigned)(unsigned long)(lds_ptr_t)(smem + r * 8192 + o0)) : "memory");
          asm volatile("ds_read_b128 %0, %1" : "=v"(v1) : "v"((unsigned)(unsigned long)(lds_ptr_t)(smem + r * 8192 + o1)) : "memory");
          asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          x.x ^= v0.x ^ v1.y; x.y ^= v0.z ^ v1.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (partner >= 2) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((it * 2 + r) & 127) * 8192),
                                           (lds_ptr_t)(smem + 49152 + r * 8192 + w * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long p1 = __builtin_readcyclecounter();
    if (lane == 0) out[(gridDim.x + blockIdx.x) * 4 + w] = p1 - p0;
    if ((x.x ^ x.y) == 0x1234567u) out[0] = 1;
  }
}

// ---- the 8p k-loop in miniature: which ingredient stretches it from the pipe's 2048 cycles per k-tile to the kernel's ~3400?
// 8 waves, two groups half a phase apart, four phases per k-tile {fragment reads; barrier; 16 MFMAs; barrier} with 8p's read
// pattern (12 / 4 / 8 / 0 ds_read_b128 per wave) and 8p's fragment-register reuse (loads TIED to the registers the previous
// segments read).  FLAGS: 1 = the fragments of consecutive k-tiles alternate between two full register sets instead;
// 2 = 8p's LDS-DMA staging (2 x global_load_lds per phase and wave into the ring, counted vmcnt(8) in phases 1, 2, 4; L2-resident
// source); 4 = 32 accumulators per wave in 8p's quadrant order (instead of 8 reused by every phase); 8 = s_setprio 1 around the
// MFMA segments; 16 = the LDS image holds random bf16 values (instead of a counter pattern); 32 = the fragment-read addresses are
// COMPUTED in each load segment (one VALU op per address register from a run-time buffer offset, as the kernel's k-loop does)
// instead of being immediates.
template <int FLAGS>
__device__ __forceinline__ void kloop_body(const char* __restrict__ src, int n_tiles, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) char smem[131072 + 8192];
  constexpr bool DB = FLAGS & 1, DMA = FLAGS & 2, ACC32 = FLAGS & 4, PRIO = FLAGS & 8, RND = FLAGS & 16, VADDR = FLAGS & 32;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 2, wn = wave & 3;
  for (int i = t; i < 131072 / 16; i += 512) {
    unsigned h = i * 2654435761u + blockIdx.x * 40503u;
    uint4 v = make_uint4(i, i * 3, i * 5, i * 7);
    if (RND) {  // bf16 pairs in (-2, 2): exponent bits 0x3f80 / 0xbf80 + random mantissa
      unsigned w[4];
      for (int q = 0; q < 4; ++q) { h = h * 1664525u + 1013904223u; w[q] = (h & 0x807f807fu) | 0x3f803f80u; }
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    reinterpret_cast<uint4*>(smem)[i] = v;
  }
  __syncthreads();
  const int cg = lane >> 4;
  const int ra = wm * 64 + (lane & 15), rb = wn * 32 + (lane & 15);
  const int oa0_ = ra * 128 + (((0 + cg) ^ (ra & 7)) << 4), oa1_ = ra * 128 + (((4 + cg) ^ (ra & 7)) << 4);
  const int ob0_ = rb * 128 + (((0 + cg) ^ (rb & 7)) << 4), ob1_ = rb * 128 + (((4 + cg) ^ (rb & 7)) << 4);
  const int rt_par = n_tiles < 0 ? 16 : 0;
  constexpr int NS = DB ? 2 : 1;
  bf16x8 a0[NS][4][2], a1[NS][4][2], b0[NS][2][2], b1[NS][2][2];
  for (int n = 0; n < NS; ++n)
    for (int k = 0; k < 2; ++k) {
      for (int i = 0; i < 4; ++i) a0[n][i][k] = a1[n][i][k] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      for (int j = 0; j < 2; ++j) b0[n][j][k] = b1[n][j][k] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  constexpr int NA = ACC32 ? 8 : 4, NB = ACC32 ? 4 : 2;
  f32x4 acc[NA][NB];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // DMA sources: as in 8p, two passes of 512 threads x 16 B per half-tile; a 2-MB region per XCD keeps them L2 hits
  const char* g0 = src + (size_t)(blockIdx.x & 7) * (2 << 20) + (size_t)((blockIdx.x >> 3) & 15) * 65536 + (size_t)t * 16;
  char* const ldsw = smem + wave * 1024;
// the destination is TIED to the variable's current register ("+v"): the new fragment lands where the old one lived
#define KL_LD(dst, off) asm volatile("ds_read_b128 %0, %1" : "+v"(dst) : "v"((unsigned)(unsigned long)(lds_ptr_t)(smem + (off))) : "memory")
#define KL_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define KL_STAGE(slot_off, kofs)                                                                              \
  do {                                                                                                        \
    if constexpr (DMA) {                                                                                      \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g0 + (kofs)), (lds_ptr_t)(ldsw + (slot_off)), 16, 0, 0);        \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g0 + (kofs) + 8192), (lds_ptr_t)(ldsw + (slot_off) + 8192), 16, 0, 0); \
    }                                                                                                         \
  } while (0)
#define KL_WAIT() do { if constexpr (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); } while (0)
#define KL_MFMA(IB, JB, AF, BF)                                                                               \
  do {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)            \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
        acc[(ACC32 ? IB : 0) + i][(ACC32 ? JB : 0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(            \
            AF[i][ks], BF[j][ks], acc[(ACC32 ? IB : 0) + i][(ACC32 ? JB : 0) + j], 0, 0, 0);                  \
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);                                                        \
  } while (0)
  // ring: parity buffers of 64 KB: A0 +0, A1 +16384, B0 +32768, B1 +49152
#define KL_TILE(S)                                                                                            \
  do {                                                                                                        \
    constexpr int bo_ = (S) * 65536, bn_ = bo_ ^ 65536;                                                       \
    int rt_ = 0;                                                                                              \
    if constexpr (VADDR) { rt_ = rt_par; asm volatile("" : "+s"(rt_)); } /* opaque run-time 0: the adds below are real VALU ops */ \
    const int oa0 = oa0_ + rt_, oa1 = oa1_ + rt_, ob0 = ob0_ + rt_, ob1 = ob1_ + rt_;                          \
    auto& A0 = a0[DB ? (S) : 0];                                                                              \
    auto& A1 = DB ? a1[(S)] : a0[0]; /* reuse: A1 lands in A0's registers, as in 8p */                        \
    auto& B0 = b0[DB ? (S) : 0];                                                                              \
    auto& B1 = b1[DB ? (S) : 0];                                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) { KL_LD(B0[j][0], bo_ + 32768 + j * 2048 + ob0); KL_LD(B0[j][1], bo_ + 32768 + j * 2048 + ob1); } \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { KL_LD(A0[i][0], bo_ + i * 2048 + oa0); KL_LD(A0[i][1], bo_ + i * 2048 + oa1); }                 \
    KL_STAGE(bn_ + 49152, kq * 16384); KL_WAIT();                                                             \
    KL_BAR(); KL_MFMA(0, 0, A0, B0); KL_BAR();                                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) { KL_LD(B1[j][0], bo_ + 49152 + j * 2048 + ob0); KL_LD(B1[j][1], bo_ + 49152 + j * 2048 + ob1); } \
    KL_STAGE(bn_ + 16384, kq * 16384 + 16384 * 64); KL_WAIT();                                                \
    KL_BAR(); KL_MFMA(0, 2, A0, B1); KL_BAR();                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { KL_LD(A1[i][0], bo_ + 16384 + i * 2048 + oa0); KL_LD(A1[i][1], bo_ + 16384 + i * 2048 + oa1); } \
    KL_STAGE(bo_ + 0, kq * 16384 + 16384 * 128);                                                              \
    KL_BAR(); KL_MFMA(4, 2, A1, B1); KL_BAR();                                                                \
    KL_STAGE(bo_ + 32768, kq * 16384 + 16384 * 192); KL_WAIT();                                               \
    KL_BAR(); KL_MFMA(4, 0, A1, B0); KL_BAR();                                                                \
    kq = (kq + 1) & 3;                                                                                        \
  } while (0)
  int kq = 0;
  if (wm == 1) KL_BAR();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n_tiles; it += 2) {
    KL_TILE(0);
    KL_TILE(1);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wm == 0) KL_BAR();
  float sum = 0.f;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) sum += acc[i][j][0] + acc[i][j][2];
  if (lane == 0) out[blockIdx.x * 8 + wave] = (t1 - t0) | (sum == 123.f ? 1ull << 63 : 0ull);
#undef KL_TILE
#undef KL_MFMA
#undef KL_WAIT
#undef KL_STAGE
#undef KL_BAR
#undef KL_LD
}
template <int FLAGS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void kloop_kernel(const char* src, int n_tiles, unsigned long long* out) {
  kloop_body<FLAGS>(src, n_tiles, out);
}

// ---- what does one s_memtime tick last?  (one wave spins until the counter has advanced by `ticks`)
__global__ void tick_kernel(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long t1 = t0;
  while (t1 - t0 < ticks) t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

// ---- does s_memtime keep counting while the only wave of a CU is stalled on memory?  (a dependent chain of HBM loads: the wave
// spends its life in s_waitcnt.)  If ticks / wall time stays at the idle clock, tick counts under load measure the real clock.
__global__ void chase_kernel(const unsigned* __restrict__ next, int hops, unsigned long long* out) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned i = threadIdx.x;
  for (int h = 0; h < hops; ++h) i = next[i];
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}

template <typename F>
static float time_ms(F&& launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs; one 512-thread workgroup per CU (96 KB LDS each)\n", prop.name, cus);
  const size_t big = (size_t)2 << 30;
  char* buf;
  unsigned* sink;
  CK(hipMalloc(&buf, big));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, big));
  const int iters = 400;  // x 8 pieces x 8 KB = 25.6 MB per workgroup
  const double bytes = (double)cus * iters * 8 * 8192;
  struct Case { const char* name; size_t pitch; size_t span; int nreg; int rowb; };
  const Case cases[] = {
      {"128-B rows, pitch 1536 (K = 768), 2 MB shared per 8 regions [L2 hits]", 1536, (size_t)2 << 20, 8, 128},
      {"128-B rows, pitch 6144 (K = 3072), 2 MB shared per 8 regions [L2 hits]", 6144, (size_t)2 << 20, 8, 128},
      {"contiguous 8 KB pieces, 2 MB shared per 8 regions [L2 hits]", 128, (size_t)2 << 20, 8, 8192},
      {"128-B rows, pitch 1536, 6 MB private per workgroup [HBM / MALL]", 1536, (size_t)6 << 20, 256, 128},
      {"contiguous 8 KB pieces, 6 MB private per workgroup [HBM / MALL]", 128, (size_t)6 << 20, 256, 8192},
      {"64-B rows, pitch 1536, 2 MB shared per 8 regions [L2 hits]", 1536, (size_t)2 << 20, 8, 64},
  };
  for (const Case& c : cases) {
    for (int dma = 0; dma < 2; ++dma) {
      auto launch = [&]() {
        if (c.rowb == 128) {
          if (dma) hipLaunchKernelGGL((load_kernel<true, 128>), dim3(cus), dim3(512), 0, 0, buf, c.pitch, c.span, c.nreg, iters, sink);
          else hipLaunchKernelGGL((load_kernel<false, 128>), dim3(cus), dim3(512), 0, 0, buf, c.pitch, c.span, c.nreg, iters, sink);
        } else if (c.rowb == 64) {
          if (dma) hipLaunchKernelGGL((load_kernel<true, 64>), dim3(cus), dim3(512), 0, 0, buf, c.pitch, c.span, c.nreg, iters, sink);
          else hipLaunchKernelGGL((load_kernel<false, 64>), dim3(cus), dim3(512), 0, 0, buf, c.pitch, c.span, c.nreg, iters, sink);
        } else {
          if (dma) hipLaunchKernelGGL((load_kernel<true, 8192>), dim3(cus), dim3(512), 0, 0, buf, (size_t)8192, c.span, c.nreg, iters, sink);
          else hipLaunchKernelGGL((load_kernel<false, 8192>), dim3(cus), dim3(512), 0, 0, buf, (size_t)8192, c.span, c.nreg, iters, sink);
        }
      };
      const float ms = time_ms(launch, 5);
      printf("load  %-8s %-78s %7.1f GB/s per CU  %6.2f TB/s\n", dma ? "lds-dma" : "to-vgpr", c.name, bytes / cus / (ms * 1e6),
             bytes / (ms * 1e9));
    }
  }
  {  // s_memtime while a lone wave is stalled on dependent HBM loads
    const size_t n = (size_t)64 << 20;  // 256 MB of indices: every hop is a miss
    unsigned* nx;
    unsigned long long* co;
    CK(hipMalloc(&nx, n * 4));
    CK(hipMalloc(&co, 16));
    std::vector<unsigned> hnx(n);
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hnx[i] = (unsigned)(x % n); }
    CK(hipMemcpy(nx, hnx.data(), n * 4, hipMemcpyHostToDevice));
    const int hops = 4000;
    hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(64), 0, 0, nx, 10, co);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(64), 0, 0, nx, hops, co);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hco[2];
    CK(hipMemcpy(hco, co, 16, hipMemcpyDeviceToHost));
    printf("s_memtime under stalls: %d dependent HBM loads, %llu ticks in %.3f ms = %.1f MHz (%.0f ns per hop): the counter %s while the wave waits\n",
           hops, hco[0], ms, hco[0] / (ms * 1e3), ms * 1e6 / hops, hco[0] / (ms * 1e3) > 2000.0 ? "keeps running" : "SLOWS DOWN");
  }
  {  // s_memtime calibration
    unsigned long long* to;
    CK(hipMalloc(&to, 8));
    const unsigned long long ticks = 200000000ull;
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, 0, 1000ull, to);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, 0, ticks, to);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("s_memtime: %llu ticks in %.3f ms = %.1f MHz (an idle chip: one wave spinning)\n", ticks, ms, ticks / (ms * 1e3));
  }
  {  // the 8p k-loop in miniature, ingredient by ingredient
    unsigned long long* ko;
    CK(hipMalloc(&ko, (size_t)cus * 8 * 8));
    std::vector<unsigned long long> h(cus * 8);
    const int n_tiles = 2000;
    auto run = [&](int flags, const char* name) {
      for (int rep = 0; rep < 2; ++rep) {
        switch (flags) {
#define KCASE(F) case F: hipLaunchKernelGGL(kloop_kernel<F>, dim3(cus), dim3(512), 0, 0, buf, n_tiles, ko); break;
          KCASE(0) KCASE(1) KCASE(2) KCASE(4) KCASE(6) KCASE(14) KCASE(30) KCASE(16) KCASE(20) KCASE(32) KCASE(62)
#undef KCASE
        }
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(h.data(), ko, h.size() * 8, hipMemcpyDeviceToHost));
      double sum = 0;
      for (auto v : h) sum += (double)(v & ~(1ull << 63));
      printf("k-loop   %-78s %7.0f cycles per k-tile (2048 = the pipe's rate)\n", name, sum / h.size() / n_tiles);
    };
    run(0, "8p's phases, reads and register reuse; no DMA, 8 accumulators, pattern data");
    run(1, "+ fresh fragment registers (two sets)");
    run(16, "+ random bf16 data in LDS");
    run(4, "+ 32 accumulators (8p's quadrants)");
    run(20, "+ 32 accumulators + random data");
    run(2, "+ LDS-DMA staging (2 per phase, vmcnt(8) in phases 1, 2, 4)");
    run(6, "+ DMA + 32 accumulators");
    run(14, "+ DMA + 32 accumulators + s_setprio");
    run(30, "+ DMA + 32 accumulators + s_setprio + random data  (= 8p's k-loop)");
    run(32, "first line + fragment-read addresses computed by VALU ops in every load segment");
    run(62, "8p's k-loop + those VALU ops");
  }
  {  // MFMA waves vs LDS / DMA partner waves
    unsigned long long* mo;
    CK(hipMalloc(&mo, (size_t)cus * 8 * 8));
    CK(hipMemset(mo, 0, (size_t)cus * 8 * 8));
    std::vector<unsigned long long> h(cus * 8);
    const int n_it = 4000;  // x 16 MFMAs
    const char* names[] = {"alone", "partner: ds_read_b128 stream", "partner: LDS-DMA stream", "partner: 12 reads : 2 DMA",
                           "partner: 12 reads : 2 DMA, MFMA waves at s_setprio 1"};
    for (int partner = 0; partner < 5; ++partner) {
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(mix_kernel, dim3(cus), dim3(512), 0, 0, buf, partner, n_it, 1 << 20 >> (partner == 2 ? 2 : 4), mo);
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(h.data(), mo, h.size() * 8, hipMemcpyDeviceToHost));
      double sum = 0, psum = 0;
      for (int i = 0; i < cus * 4; ++i) sum += (double)(h[i] & ~(1ull << 63));
      for (int i = cus * 4; i < cus * 8; ++i) psum += (double)h[i];
      const int pit = 1 << 20 >> (partner == 2 ? 2 : 4);
      const double pbytes = (double)pit * ((partner == 1 || partner >= 3 ? 12 * 1024.0 : 0.0) + (partner >= 2 ? 2 * 1024.0 : 0.0));
      printf("mfma-mix %-56s %6.2f cycles per v_mfma_f32_16x16x32_bf16 (16.0 = the pipe's rate); partner waves: %6.1f LDS bytes per cycle and CU over %.1fx the MFMA loop's duration\n",
             names[partner], sum / (cus * 4) / (n_it * 16.0), partner ? 4.0 * pbytes / (psum / (cus * 4)) : 0.0,
             partner ? (psum / (cus * 4)) / (sum / (cus * 4)) : 0.0);
    }
  }
  // the GEMM's operand stream alone: qkv (N = 2304, K = 768), fc1 (3072, 768), fc2 (768, 3072), proj (768, 768) at 110 views
  {
    const int mt = 605;
    struct G { const char* name; int nt, K, group, mblock; };
    const G gs[] = {{"qkv  N=2304 K=768 ", 9, 768, 9, 4}, {"fc1  N=3072 K=768 ", 12, 768, 12, 4}, {"proj N=768  K=768 ", 3, 768, 3, 1},
                    {"fc2  N=768  K=3072", 3, 3072, 3, 1}};
    char* Wb = buf + ((size_t)1 << 30);  // A at buf (605 * 256 * K * 2 <= 952 MB), W behind it
    for (const G& g : gs) {
      const int tiles = mt * g.nt, nk = g.K / 64;
      const double rounds = (double)tiles / cus;
      struct V { const char* name; int D, rot, a_panels; };
      const V vs[] = {{"as the kernel walks it (8 pieces in flight)", 8, 0, 0}, {"16 pieces in flight", 16, 0, 0}, {"4 pieces in flight", 4, 0, 0},
                      {"A L2-resident (8 panels only)", 8, 0, 8}, {"k start rotated per (n, m in block)", 8, 1, 0},
                      {"k start rotated per n", 8, 2, 0}, {"rotated per (n, m), 16 in flight", 16, 1, 0}};
      for (const V& v : vs) {
        auto launch = [&]() {
          if (v.D == 8) hipLaunchKernelGGL((gemm_stream_kernel<8>), dim3(tiles), dim3(512), 0, 0, buf, Wb, mt, g.nt, g.K, g.group, g.mblock, v.rot, v.a_panels);
          else if (v.D == 16) hipLaunchKernelGGL((gemm_stream_kernel<16>), dim3(tiles), dim3(512), 0, 0, buf, Wb, mt, g.nt, g.K, g.group, g.mblock, v.rot, v.a_panels);
          else hipLaunchKernelGGL((gemm_stream_kernel<4>), dim3(tiles), dim3(512), 0, 0, buf, Wb, mt, g.nt, g.K, g.group, g.mblock, v.rot, v.a_panels);
        };
        const float ms = time_ms(launch, 3);
        printf("gemm-stream %s %-46s %7.1f us per launch  %5.2f us per k-tile and CU  %6.1f GB/s per CU\n", g.name, v.name, ms * 1e3,
               ms * 1e3 / (rounds * nk), 65536.0 / (ms * 1e3 / (rounds * nk)) / 1e3);
      }
    }
  }
  // stores: the qkv output of a 110-view launch (154880 x 2304 bf16 = 605 x 9 tiles) and the fc1 output (x 3072 = 12 tiles)
  for (int nt : {9, 12}) {
    const int tiles = 605 * nt;
    const size_t n_bytes = (size_t)nt * 512;
    const double sb = (double)tiles * 131072;
    for (int blocked = 0; blocked < 2; ++blocked) {
      auto launch = [&]() {
        if (blocked) hipLaunchKernelGGL((store_kernel<true>), dim3(cus), dim3(512), 0, 0, buf, n_bytes, nt, tiles);
        else hipLaunchKernelGGL((store_kernel<false>), dim3(cus), dim3(512), 0, 0, buf, n_bytes, nt, tiles);
      };
      const float ms = time_ms(launch, 5);
      printf("store %-8s 605 x %2d tiles of 256 x 512 B (%s)%*s %7.1f GB/s per CU  %6.2f TB/s  (%.0f us)\n",
             blocked ? "blocked" : "rowmajor", nt, blocked ? "one 128-KB burst per tile" : "512-B pieces at the row pitch",
             blocked ? 9 : 5, "", sb / cus / (ms * 1e6), sb / (ms * 1e9), ms * 1e3);
    }
  }
  return 0;
}
