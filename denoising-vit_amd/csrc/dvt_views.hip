// View synthesis for stage 1: crop + resize (bicubic, antialias) + horizontal
// flip of one image into V views, entirely on the device.
//
// Reference: dvt/dataset/transform.py:48-52 `F.resized_crop(img, i, j, h, w, size, BICUBIC,
// antialias=True)` and :70 `F.hflip`, fed by single_image_dataset.py:33-38 (base resize).  The
// resampler is torchvision/ATen's separable anti-aliased filter (third party; restated from
// aten/src/ATen/native/UpSampleKernel `_compute_indices_min_size_weights_aa`, align_corners=False):
//   scale   = in / out                      support = 2 * max(scale, 1)      (bicubic: 4 taps wide)
//   center  = scale * (o + 0.5)             invscale = scale >= 1 ? 1/scale : 1
//   xmin    = max(int(center - support + 0.5), 0)
//   xsize   = min(int(center + support + 0.5), in) - xmin
//   w_j     = cubic_{a=-0.5}((j + xmin - center + 0.5) * invscale),  normalised to sum 1
// The window is TRUNCATED at the borders (not clamped).  Random crops of scale (0.1, 0.5) of a 518^2
// image are always up-sampled (<= 5 taps per axis); down-scaling up to 3.5x (16 taps) is supported
// for the base resize.  One thread per output pixel, all three channels; the source image
// (3.2 MB) stays L2-resident while the 769 views stream out (2.5 GB of writes = the HBM floor).
#include "dvt_common.h"

namespace {

constexpr int MAX_TAPS = 16;

__device__ __forceinline__ float cubic_aa(float x) {
  const float a = -0.5f;
  x = fabsf(x);
  if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
  if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
  return 0.0f;
}

// window + normalised weights of output index o along one axis
__device__ __forceinline__ void aa_window(int o, int in_size, int out_size, int& xmin, int& xsize,
                                          float w[MAX_TAPS]) {
  const float scale = (float)in_size / (float)out_size;
  const float support = 2.0f * fmaxf(scale, 1.0f);
  const float center = scale * ((float)o + 0.5f);
  const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  xmin = max((int)(center - support + 0.5f), 0);
  xsize = min(min((int)(center + support + 0.5f), in_size) - xmin, MAX_TAPS);
  float total = 0.f;
#pragma unroll
  for (int j = 0; j < MAX_TAPS; ++j) {
    w[j] = j < xsize ? cubic_aa(((float)(j + xmin) - center + 0.5f) * invscale) : 0.f;
    total += w[j];
  }
  const float inv = total != 0.f ? 1.0f / total : 0.f;
#pragma unroll
  for (int j = 0; j < MAX_TAPS; ++j) w[j] *= inv;
}

__global__ __launch_bounds__(256) void render_views_kernel(const float* __restrict__ img, int H, int W,
                                                           const int32_t* __restrict__ boxes,
                                                           float* __restrict__ out, int V, int OH,
                                                           int OW) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)V * OH * OW) return;
  const int ox = (int)(t % OW), oy = (int)((t / OW) % OH), v = (int)(t / ((long long)OW * OH));
  const int32_t* b = boxes + (size_t)v * 5;
  const int top = b[0], left = b[1], ch = b[2], cw = b[3], flip = b[4];
  const int sx = flip ? OW - 1 - ox : ox;  // hflip of the resized crop (transform.py:70)
  int ymin, ysz, xmin, xsz;
  float wy[MAX_TAPS], wx[MAX_TAPS];
  aa_window(oy, ch, OH, ymin, ysz, wy);
  aa_window(sx, cw, OW, xmin, xsz, wx);
  const size_t plane = (size_t)H * W;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  for (int jy = 0; jy < ysz; ++jy) {
    const float* row = img + (size_t)(top + ymin + jy) * W + left + xmin;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    for (int jx = 0; jx < xsz; ++jx) {
      const float w = wx[jx];
      r0 = fmaf(w, row[jx], r0);
      r1 = fmaf(w, row[plane + jx], r1);
      r2 = fmaf(w, row[2 * plane + jx], r2);
    }
    acc0 = fmaf(wy[jy], r0, acc0);
    acc1 = fmaf(wy[jy], r1, acc1);
    acc2 = fmaf(wy[jy], r2, acc2);
  }
  const size_t oplane = (size_t)OH * OW;
  float* o = out + (size_t)v * 3 * oplane + (size_t)oy * OW + ox;
  o[0] = acc0;
  o[oplane] = acc1;
  o[2 * oplane] = acc2;
}

}  // namespace

extern "C" int dvt_render_views(const float* img, int H, int W, const int32_t* boxes, float* out,
                                int V, int OH, int OW, void* stream) {
  if (!img || !boxes || !out || H <= 0 || W <= 0 || V < 0 || OH <= 0 || OW <= 0) return DVT_E_BADARG;
  if (V == 0) return 0;
  const long long n = (long long)V * OH * OW;
  hipLaunchKernelGGL(render_views_kernel, dim3(dvt_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     img, H, W, boxes, out, V, OH, OW);
  DVT_CHECK_LAUNCH();
  return 0;
}
