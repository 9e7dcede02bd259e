// Developer probe: power / clock of a pure-MFMA loop, v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16, random operands,
// 8 waves per CU x 2 (the GEMM's occupancy).  Build: hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ src, float* out, int iters) {
  const int t = threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(blockIdx.x * 512 + t) * 8 + i]; b[i] = src[(blockIdx.x * 512 + t) * 8 + 4 + i]; }
  float s = 0.f;
  if constexpr (KIND == 0) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i * 2 + ks], b[j * 2 + ks], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
  }
  out[blockIdx.x * 512 + t] = s;
}

int main() {
  const int nb = 512, n = nb * 512 * 8;
  bf16x8* src; float* out;
  hipMalloc(&src, n * sizeof(bf16x8)); hipMalloc(&out, nb * 512 * 4);
  unsigned short* h = (unsigned short*)malloc(n * 16);
  srand(1);
  for (int i = 0; i < n * 8; ++i) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  hipMemcpy(src, h, n * 16, hipMemcpyHostToDevice);
  for (int kind = 0; kind < 2; ++kind) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      const int iters = rep ? 5000000 : 100000;
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(512), 0, 0, src, out, iters);
      else hipLaunchKernelGGL(k<1>, dim3(nb), dim3(512), 0, 0, src, out, iters);
      hipEventRecord(e1);
      if (rep == 1) {
        std::this_thread::sleep_for(std::chrono::milliseconds(1200));
        printf("kind %d mid-run: ", kind); fflush(stdout);
        system("rocm-smi --showclocks --showpower | grep -E 'sclk|Socket' | tr '\\n' ' '; echo");
      }
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)nb * 8 * iters * 16.0 * 16 * 16 * 32 * 2;
      printf("kind %d (%s): %.1f ms, %.1f TF/s\n", kind, kind ? "32x32x16" : "16x16x32", ms, fl / ms / 1e9);
    }
  }
  return 0;
}
