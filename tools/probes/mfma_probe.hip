// Micro-probe: cycles per v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x32_bf16 for 1, 2, 4 independent
// accumulator chains per wave, and the effective shader clock (s_memtime ticks vs wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int CH>
__global__ void k_f32(float* out, long long* cyc, int n) {
  floatx16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH>
__global__ void k_bf16(float* out, long long* cyc, int n) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x); b[i] = (short)(0x3f80 + i); }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <typename K>
void run(const char* name, K kern, int ch, int blocks, int threads, int n, double flops_per_mfma) {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, threads>>>(out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0); kern<<<blocks, threads>>>(out, cyc, n); hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double waves = (double)blocks * threads / 64;
  double tf = flops_per_mfma * ch * (double)n * waves / (ms * 1e-3) / 1e12;
  printf("%-6s chains=%d blocks=%4d thr=%4d: %8.1f memtime-ticks/MFMA, wall %.3f ms, %.1f TF/s, tick rate %.1f MHz\n",
         name, ch, blocks, threads, (double)c / ((double)n * ch), ms, tf, (double)c / (ms * 1e-3) / 1e6);
  hipFree(out); hipFree(cyc);
}
int main() {
  const double F32 = 2.0 * 32 * 32 * 2, BF = 2.0 * 16 * 16 * 32;
  for (int blocks : {256, 512}) {
    run("f32", k_f32<1>, 1, blocks, 256, 20000, F32);
    run("f32", k_f32<2>, 2, blocks, 256, 10000, F32);
    run("f32", k_f32<4>, 4, blocks, 256, 5000, F32);
  }
  run("f32", k_f32<1>, 1, 256, 256, 400, F32);   // short kernel (~15 us): clock ramp?
  run("f32", k_f32<1>, 1, 192, 256, 400, F32);
  for (int blocks : {256, 512}) {
    run("bf16", k_bf16<1>, 1, blocks, 256, 40000, BF);
    run("bf16", k_bf16<4>, 4, blocks, 256, 10000, BF);
    run("bf16", k_bf16<16>, 16, blocks, 256, 2500, BF);
  }
  run("bf16", k_bf16<16>, 16, 256, 512, 2500, BF);
  return 0;
}
