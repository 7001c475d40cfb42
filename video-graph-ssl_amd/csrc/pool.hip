// Max / average pooling for NCDHW clips on gfx950 (HBM-bound gathers, lanes along W).
// Reference call sites: see include/gca_hip.h (Pooling section).
#include "gca_common.h"
#include <math.h>

namespace {

// One thread per output element.  Tie break = first maximum in (d,h,w) scan order and NaN
// propagates, exactly as ATen's max_pool3d_with_indices (val > max || isnan(val)).
__global__ __launch_bounds__(256) void maxpool3d_fwd_kernel(gca_pool_geom g, const float* __restrict__ x,
                                                            float* __restrict__ y, int* __restrict__ argmax,
                                                            long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int OHW = g.OH * g.OW, OSP = g.OD * OHW;
  const long long plane = i / OSP;
  const int o = (int)(i - plane * OSP);
  const int od = o / OHW, r = o - od * OHW, oh = r / g.OW, ow = r - oh * g.OW;
  int d0 = od * g.sd - g.pd, h0 = oh * g.sh - g.ph, w0 = ow * g.sw - g.pw;
  const int d1 = min(d0 + g.kd, g.D), h1 = min(h0 + g.kh, g.H), w1 = min(w0 + g.kw, g.W);
  d0 = max(d0, 0); h0 = max(h0, 0); w0 = max(w0, 0);
  const float* xp = x + plane * ((long long)g.D * g.H * g.W);
  float best = -INFINITY;
  int bi = (d0 * g.H + h0) * g.W + w0;
  for (int d = d0; d < d1; ++d)
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) {
        const int idx = (d * g.H + h) * g.W + w;
        const float v = xp[idx];
        if (v > best || isnan(v)) { best = v; bi = idx; }
      }
  y[i] = best;
  if (argmax) argmax[i] = bi;
}

// One thread per INPUT element: gathers from the (few) windows that cover it, no atomics.
__global__ __launch_bounds__(256) void maxpool3d_bwd_kernel(gca_pool_geom g, const float* __restrict__ dy,
                                                            const int* __restrict__ argmax, float* __restrict__ dx,
                                                            long long total, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int HW = g.H * g.W, SP = g.D * HW;
  const long long plane = i / SP;
  const int s = (int)(i - plane * SP);
  const int d = s / HW, r = s - d * HW, h = r / g.W, w = r - h * g.W;
  // outputs o with o*s - p <= d < o*s - p + k
  const int od_lo = max(0, (d + g.pd - g.kd + g.sd) / g.sd), od_hi = min(g.OD - 1, (d + g.pd) / g.sd);
  const int oh_lo = max(0, (h + g.ph - g.kh + g.sh) / g.sh), oh_hi = min(g.OH - 1, (h + g.ph) / g.sh);
  const int ow_lo = max(0, (w + g.pw - g.kw + g.sw) / g.sw), ow_hi = min(g.OW - 1, (w + g.pw) / g.sw);
  const int OHW = g.OH * g.OW;
  const long long obase = plane * ((long long)g.OD * OHW);
  float acc = 0.f;
  for (int od = od_lo; od <= od_hi; ++od)
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const long long o = obase + (long long)od * OHW + oh * g.OW + ow;
        if (argmax[o] == s) acc += dy[o];
      }
  dx[i] = accumulate ? dx[i] + acc : acc;
}

// y[p] = norm * sum_d wt[d] * sum_hw x[p,d,hw]; one wave per (n,c) plane.
__global__ __launch_bounds__(256) void wavgpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                           float norm, long long NC, int D, int HW,
                                                           float* __restrict__ y) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= NC) return;
  const int lane = threadIdx.x & 63;
  const float* xp = x + p * ((long long)D * HW);
  float s = 0.f;
  for (int d = 0; d < D; ++d) {
    float t = 0.f;
    for (int i = lane; i < HW; i += 64) t += xp[(long long)d * HW + i];
    s += (wt ? wt[d] : 1.f) * t;
  }
  s = gca_wave_sum(s);
  if (lane == 0) y[p] = s * norm;
}

__global__ __launch_bounds__(256) void wavgpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ wt,
                                                           float norm, long long total, int D, int HW,
                                                           float* __restrict__ dx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long p = i / ((long long)D * HW);
  const int d = (int)((i / HW) % D);
  dx[i] = dy[p] * norm * (wt ? wt[d] : 1.f);
}

inline bool pool_ok(const gca_pool_geom* g) {
  if (!g || g->N <= 0 || g->C <= 0 || g->D <= 0 || g->H <= 0 || g->W <= 0) return false;
  if (g->kd <= 0 || g->kh <= 0 || g->kw <= 0 || g->sd <= 0 || g->sh <= 0 || g->sw <= 0) return false;
  if (g->pd < 0 || g->ph < 0 || g->pw < 0 || 2 * g->pd > g->kd || 2 * g->ph > g->kh || 2 * g->pw > g->kw) return false;
  // floor mode (ceil_mode=False everywhere in the reference)
  if (g->OD != (g->D + 2 * g->pd - g->kd) / g->sd + 1 || g->OH != (g->H + 2 * g->ph - g->kh) / g->sh + 1 ||
      g->OW != (g->W + 2 * g->pw - g->kw) / g->sw + 1)
    return false;
  return g->OD > 0 && g->OH > 0 && g->OW > 0;
}

}  // namespace

extern "C" {

int gca_maxpool3d_fwd(const gca_pool_geom* g, const float* x, float* y, int32_t* argmax, void* stream) {
  if (!pool_ok(g) || !x || !y) return GCA_EINVAL;
  const long long total = (long long)g->N * g->C * g->OD * g->OH * g->OW;
  hipLaunchKernelGGL(maxpool3d_fwd_kernel, dim3((unsigned)gca_ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, *g, x, y, argmax, total);
  return gca_launch_status();
}

int gca_maxpool3d_bwd(const gca_pool_geom* g, const float* dy, const int32_t* argmax, float* dx,
                      int accumulate, void* stream) {
  if (!pool_ok(g) || !dy || !argmax || !dx) return GCA_EINVAL;
  const long long total = (long long)g->N * g->C * g->D * g->H * g->W;
  hipLaunchKernelGGL(maxpool3d_bwd_kernel, dim3((unsigned)gca_ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, *g, dy, argmax, dx, total, accumulate ? 1 : 0);
  return gca_launch_status();
}

int gca_wavgpool_fwd(const float* x, const float* wt, float norm, int64_t NC, int64_t D, int64_t HW,
                     float* y, void* stream) {
  if (!x || !y || NC <= 0 || D <= 0 || HW <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(wavgpool_fwd_kernel, dim3((unsigned)gca_ceil_div(NC, 4)), dim3(256), 0, (hipStream_t)stream,
                     x, wt, norm, (long long)NC, (int)D, (int)HW, y);
  return gca_launch_status();
}

int gca_wavgpool_bwd(const float* dy, const float* wt, float norm, int64_t NC, int64_t D, int64_t HW,
                     float* dx, void* stream) {
  if (!dy || !dx || NC <= 0 || D <= 0 || HW <= 0) return GCA_EINVAL;
  const long long total = (long long)NC * D * HW;
  hipLaunchKernelGGL(wavgpool_bwd_kernel, dim3((unsigned)gca_ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, wt, norm, total, (int)D, (int)HW, dx);
  return gca_launch_status();
}

}  // extern "C"
