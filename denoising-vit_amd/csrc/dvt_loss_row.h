// Per-row loss forward + backward shared by the stand-alone loss kernel (dvt_loss.hip) and the fused
// row kernel of the fit (dvt_fit_fused.hip): ONE arithmetic, two launch shapes.
// Reference: dvt/models/offline_denoiser.py:113-140 (+ main_img_denoising.py:88 loss * 1024).
// One wave (64 lanes) owns one row of C <= 1024 channels held in registers as float4.
#pragma once
#include "dvt_common.h"

constexpr int DVT_LOSS_MAXQ = 4;  // float4 slots per lane: C <= 64*4*4 = 1024

__device__ __forceinline__ float dvt_sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// Registers of one row, loaded first (so that a caller can put several rows' loads in flight) ...
template <bool HAS_RES>
struct DvtLossRowRegs {
  float4 f[DVT_LOSS_MAXQ], g[DVT_LOSS_MAXQ], r[DVT_LOSS_MAXQ], h[HAS_RES ? DVT_LOSS_MAXQ : 1];
};

template <bool HAS_RES>
__device__ __forceinline__ void dvt_loss_row_load(DvtLossRowRegs<HAS_RES>& x, const float4* __restrict__ F,
                                                  const float4* __restrict__ G, const float4* __restrict__ Hres,
                                                  const float4* __restrict__ raw, int cq, int lane) {
#pragma unroll
  for (int s = 0; s < DVT_LOSS_MAXQ; ++s) {
    const int q = lane + 64 * s;
    if (q < cq) {
      x.f[s] = F[q];
      x.g[s] = G[q];
      x.r[s] = raw[q];
      if (HAS_RES) x.h[s] = Hres[q];
    }
  }
}

// ... then reduced and differentiated.  Outputs (each optional):
//   d_pred / d_hres : global fp32 gradient rows (grad_scale * dloss/dpred, .../dh)
//   d_G             : fp32 atomics into the lattice row of the G gradient
//   row_sums        : {sse, cos, res_sse, res_abs, ...} of this row (lane 0)
//   b_pred / b_hres : the same gradient rows rounded to bf16 (4 values = 8 B per float4), e.g. in LDS
//   f_pred / f_hres : the same gradient rows as fp32 (e.g. the LDS images of the fp32-operand row kernel)
template <bool HAS_RES>
__device__ __forceinline__ void dvt_loss_row_compute(const DvtLossRowRegs<HAS_RES>& x, float4* __restrict__ d_pred,
                                                     float4* __restrict__ d_hres, float* __restrict__ d_G,
                                                     float* __restrict__ row_sums, int n, int cq, float grad_scale,
                                                     int lane, uint2* b_pred, uint2* b_hres, float4* f_pred = nullptr,
                                                     float4* f_hres = nullptr) {
  float4 vp[DVT_LOSS_MAXQ], vr[DVT_LOSS_MAXQ], vh[DVT_LOSS_MAXQ], vfg[DVT_LOSS_MAXQ];
  float sse = 0.f, dot = 0.f, np = 0.f, nr = 0.f, rsse = 0.f, rabs = 0.f;
#pragma unroll
  for (int s = 0; s < DVT_LOSS_MAXQ; ++s) {
    const int q = lane + 64 * s;
    if (q < cq) {
      const float4 f = x.f[s], g = x.g[s], r = x.r[s];
      float4 fg = make_float4(f.x + g.x, f.y + g.y, f.z + g.z, f.w + g.w);
      float4 p = fg;
      if (HAS_RES) {
        const float4 h = x.h[s];
        vh[s] = h;
        p = make_float4(fg.x + h.x, fg.y + h.y, fg.z + h.z, fg.w + h.w);
        // gt_residual = raw - F - G ; residual terms use (h - gt)
        const float ex = h.x - (r.x - fg.x), ey = h.y - (r.y - fg.y), ez = h.z - (r.z - fg.z),
                    ew = h.w - (r.w - fg.w);
        rsse += ex * ex + ey * ey + ez * ez + ew * ew;
        rabs += fabsf(h.x) + fabsf(h.y) + fabsf(h.z) + fabsf(h.w);
      }
      vp[s] = p;
      vr[s] = r;
      vfg[s] = fg;
      const float dx = p.x - r.x, dy = p.y - r.y, dz = p.z - r.z, dw = p.w - r.w;
      sse += dx * dx + dy * dy + dz * dz + dw * dw;
      dot += p.x * r.x + p.y * r.y + p.z * r.z + p.w * r.w;
      np += p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w;
      nr += r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
    }
  }
  sse = wave_sum(sse);
  dot = wave_sum(dot);
  np = wave_sum(np);
  nr = wave_sum(nr);
  if (HAS_RES) {
    rsse = wave_sum(rsse);
    rabs = wave_sum(rabs);
  }
  // torch (ATen cosine_similarity): sum(x/max(|x|,eps) * y/max(|y|,eps)), eps = 1e-8
  const float n1 = sqrtf(np), n2 = sqrtf(nr);
  const bool clamped = n1 < 1e-8f;
  const float denom = fmaxf(n1, 1e-8f) * fmaxf(n2, 1e-8f);
  const float cosv = dot / denom;
  if (lane == 0 && row_sums != nullptr) {
    row_sums[0] = sse;
    row_sums[1] = cosv;
    row_sums[2] = rsse;
    row_sums[3] = rabs;
  }
  if (d_pred == nullptr && b_pred == nullptr && f_pred == nullptr) return;
  const float inv_nc = 1.0f / ((float)n * (float)(cq * 4));
  const float inv_n = 1.0f / (float)n;
  // d/dp [mse] = 2 (p - r) / (n c);  d/dp [1 - mean cos] = -(1/n) (r/denom - cos * p / |p|^2)
  // (when the clamp is active the denominator is constant: gradient = -(1/n) r / denom)
  const float a_r = -inv_n / denom;
  const float a_p = clamped ? 0.f : inv_n * cosv / np;
  const float c_mse = 2.0f * inv_nc;
#pragma unroll
  for (int s = 0; s < DVT_LOSS_MAXQ; ++s) {
    const int q = lane + 64 * s;
    if (q < cq) {
      const float4 p = vp[s], r = vr[s];
      float4 d;
      d.x = grad_scale * (c_mse * (p.x - r.x) + a_r * r.x + a_p * p.x);
      d.y = grad_scale * (c_mse * (p.y - r.y) + a_r * r.y + a_p * p.y);
      d.z = grad_scale * (c_mse * (p.z - r.z) + a_r * r.z + a_p * p.z);
      d.w = grad_scale * (c_mse * (p.w - r.w) + a_r * r.w + a_p * p.w);
      if (d_pred != nullptr) d_pred[q] = d;
      if (b_pred != nullptr) b_pred[q] = make_uint2(dvt_pack_bf16x2(d.x, d.y), dvt_pack_bf16x2(d.z, d.w));
      if (f_pred != nullptr) f_pred[q] = d;
      if (d_G != nullptr) {
        float* g = d_G + q * 4;
        atomic_add_f32(g + 0, d.x);
        atomic_add_f32(g + 1, d.y);
        atomic_add_f32(g + 2, d.z);
        atomic_add_f32(g + 3, d.w);
      }
      if (HAS_RES && (d_hres != nullptr || b_hres != nullptr || f_hres != nullptr)) {
        const float4 h = vh[s], fg = vfg[s];
        const float c_res = 0.1f * 2.0f * inv_nc, c_abs = 0.02f * inv_nc;
        float4 e;
        e.x = grad_scale * (c_res * (h.x - (r.x - fg.x)) + c_abs * dvt_sgn(h.x));
        e.y = grad_scale * (c_res * (h.y - (r.y - fg.y)) + c_abs * dvt_sgn(h.y));
        e.z = grad_scale * (c_res * (h.z - (r.z - fg.z)) + c_abs * dvt_sgn(h.z));
        e.w = grad_scale * (c_res * (h.w - (r.w - fg.w)) + c_abs * dvt_sgn(h.w));
        if (d_hres != nullptr) d_hres[q] = e;
        if (b_hres != nullptr) b_hres[q] = make_uint2(dvt_pack_bf16x2(e.x, e.y), dvt_pack_bf16x2(e.z, e.w));
        if (f_hres != nullptr) f_hres[q] = e;
      }
    }
  }
}

// F, G, Hres, raw: this row's data (G already offset to its lattice row); load + compute in one go.
template <bool HAS_RES>
__device__ __forceinline__ void dvt_loss_row(const float4* __restrict__ F, const float4* __restrict__ G,
                                             const float4* __restrict__ Hres, const float4* __restrict__ raw,
                                             float4* __restrict__ d_pred, float4* __restrict__ d_hres,
                                             float* __restrict__ d_G, float* __restrict__ row_sums, int n,
                                             int cq, float grad_scale, int lane, uint2* b_pred, uint2* b_hres) {
  DvtLossRowRegs<HAS_RES> x;
  dvt_loss_row_load<HAS_RES>(x, F, G, Hres, raw, cq, lane);
  dvt_loss_row_compute<HAS_RES>(x, d_pred, d_hres, d_G, row_sums, n, cq, grad_scale, lane, b_pred, b_hres);
}
