// Max / average pooling for NCDHW clips on gfx950 (HBM-bound gathers, lanes along W).
// Reference call sites: see include/gca_hip.h (Pooling section).
#include "gca_common.h"
#include <math.h>
#include <cstdlib>

namespace {

struct PoolMagic {
  gca_magic osp, ohw, ow;       // output decomposition
  gca_magic sp, hw, w;          // input decomposition
  gca_magic sd, sh, sw;         // strides
  gca_magic c;                  // channels (plane -> channel for the fused BN+ReLU producer)
};

// One thread per output element.  Tie break = first maximum in (d,h,w) scan order and NaN propagates, exactly
// as ATen's max_pool3d_with_indices (val > max || isnan(val)).  KD/KH/KW > 0: window fully unrolled with
// clamped (always in-bounds) loads, so all taps are in flight at once; 0 = run-time window.
template <typename T, int KD, int KH, int KW>
__global__ __launch_bounds__(256) void maxpool3d_fwd_kernel(gca_pool_geom g, PoolMagic q, const T* __restrict__ x,
                                                            T* __restrict__ y, int* __restrict__ argmax,
                                                            unsigned total, const float* __restrict__ scale,
                                                            const float* __restrict__ shift) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned plane = gca_fdiv(i, q.osp), o = i - plane * q.osp.d;
  // optional fused producer: the pooled tensor is relu(x*scale[c] + shift[c]) (BatchNorm + ReLU of the conv output x),
  // evaluated on the fly exactly as bn_apply would, so that tensor is never written or read
  float sc = 1.f, sf = 0.f;
  if (scale) { const unsigned c = plane - gca_fdiv(plane, q.c) * q.c.d; sc = scale[c]; sf = shift[c]; }
#define GCA_POOL_VAL(v) (scale ? fmaxf((v) * sc + sf, 0.f) : (v))
  const int od = (int)gca_fdiv(o, q.ohw), r = (int)(o - (unsigned)od * q.ohw.d);
  const int oh = (int)gca_fdiv((unsigned)r, q.ow), ow = r - oh * g.OW;
  const int d0 = od * g.sd - g.pd, h0 = oh * g.sh - g.ph, w0 = ow * g.sw - g.pw;
  const T* xp = x + (long long)plane * ((long long)g.D * g.H * g.W);
  float best = -INFINITY;
  int bi = (max(d0, 0) * g.H + max(h0, 0)) * g.W + max(w0, 0);
  if (KD > 0) {
#pragma unroll
    for (int a = 0; a < KD; ++a)
#pragma unroll
      for (int b = 0; b < KH; ++b)
#pragma unroll
        for (int c = 0; c < KW; ++c) {
          const int d = d0 + a, h = h0 + b, w = w0 + c;
          const bool ok = (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
          const int idx = (d * g.H + h) * g.W + w;
          const float v = GCA_POOL_VAL((float)xp[ok ? idx : bi]);
          if (ok && (v > best || isnan(v))) { best = v; bi = idx; }
        }
  } else {
    const int d1 = min(d0 + g.kd, g.D), h1 = min(h0 + g.kh, g.H), w1 = min(w0 + g.kw, g.W);
    for (int d = max(d0, 0); d < d1; ++d)
      for (int h = max(h0, 0); h < h1; ++h)
        for (int w = max(w0, 0); w < w1; ++w) {
          const int idx = (d * g.H + h) * g.W + w;
          const float v = GCA_POOL_VAL((float)xp[idx]);
          if (v > best || isnan(v)) { best = v; bi = idx; }
        }
  }
#undef GCA_POOL_VAL
  y[i] = (T)best;
  if (argmax) argmax[i] = bi;
}

// One thread per INPUT element: gathers from the (few) windows that cover it, no atomics (deterministic, fixed
// summation order od,oh,ow).  CD/CH/CW = ceil(k/s) per axis is the most windows that can cover one element;
// > 0: fully unrolled with clamped loads (argmax and dy of all candidates in flight together); 0 = run-time.
template <typename T, int CD, int CH, int CW>
__global__ __launch_bounds__(256) void maxpool3d_bwd_kernel(gca_pool_geom g, PoolMagic q, const T* __restrict__ dy,
                                                            const int* __restrict__ argmax, T* __restrict__ dx,
                                                            unsigned total, int accumulate) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned plane = gca_fdiv(i, q.sp);
  const int s = (int)(i - plane * q.sp.d);
  const int d = (int)gca_fdiv((unsigned)s, q.hw), r = s - d * (int)q.hw.d;
  const int h = (int)gca_fdiv((unsigned)r, q.w), w = r - h * g.W;
  // outputs o with o*s - p <= v < o*s - p + k   <=>   ceil((v + p - k + 1)/s) <= o <= floor((v + p)/s)
  const int td = d + g.pd - g.kd + g.sd, th = h + g.ph - g.kh + g.sh, tw = w + g.pw - g.kw + g.sw;
  const int od_lo = td > 0 ? (int)gca_fdiv((unsigned)td, q.sd) : 0, od_hi = min(g.OD - 1, (int)gca_fdiv((unsigned)(d + g.pd), q.sd));
  const int oh_lo = th > 0 ? (int)gca_fdiv((unsigned)th, q.sh) : 0, oh_hi = min(g.OH - 1, (int)gca_fdiv((unsigned)(h + g.ph), q.sh));
  const int ow_lo = tw > 0 ? (int)gca_fdiv((unsigned)tw, q.sw) : 0, ow_hi = min(g.OW - 1, (int)gca_fdiv((unsigned)(w + g.pw), q.sw));
  const int OHW = g.OH * g.OW;
  const long long obase = (long long)plane * ((long long)g.OD * OHW);
  float acc = 0.f;
  if (CD > 0) {
    constexpr int NE = CD * CH * CW > 0 ? CD * CH * CW : 1;
    int am[NE];
    float gv[NE];
    bool okv[NE];
#pragma unroll
    for (int a = 0; a < CD; ++a)
#pragma unroll
      for (int b = 0; b < CH; ++b)
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          const int od = od_lo + a, oh = oh_lo + b, ow = ow_lo + c;
          const bool ok = od <= od_hi && oh <= oh_hi && ow <= ow_hi;
          const long long o = obase + (ok ? (long long)od * OHW + oh * g.OW + ow : 0);
          const int e = (a * CH + b) * CW + c;
          okv[e] = ok; am[e] = argmax[o]; gv[e] = (float)dy[o];
        }
#pragma unroll
    for (int e = 0; e < CD * CH * CW; ++e) acc += (okv[e] && am[e] == s) ? gv[e] : 0.f;
  } else {
    for (int od = od_lo; od <= od_hi; ++od)
      for (int oh = oh_lo; oh <= oh_hi; ++oh)
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
          const long long o = obase + (long long)od * OHW + oh * g.OW + ow;
          if (argmax[o] == s) acc += (float)dy[o];
        }
  }
  dx[i] = (T)(accumulate ? (float)dx[i] + acc : acc);
}

// LDS-tiled backward of the 3x3x3 / stride 2 / pad 1 pools behind the R(2+1)D and 3D-ResNet stems (resnet2p1d.py:178,
// resnet.py:127): a block owns a box of 4 x 16 x 16 INPUT elements of one plane; the <= 3 x 9 x 9
// windows that can cover them are staged once as (argmax, dy) pairs -- half a load per input element instead of sixteen --
// and every thread finishes four consecutive-w elements (one 16-byte store), summing the matching windows in the same
// (od, oh, ow) order as maxpool3d_bwd_kernel (deterministic, no atomics): 0.32 vs 0.48 ms on the (32,64,16,56,56) stem map.
// (The same tiling of the FORWARD pool -- 5 x 17 x 33 inputs staged per 2 x 8 x 16 outputs, producer evaluated once per
// staged input -- was measured and not kept: 0.35 vs 0.32 ms.)
constexpr int PB_D = 4, PB_H = 16, PB_W = 16;
constexpr int PB_OD = PB_D / 2 + 1, PB_OH = PB_H / 2 + 1, PB_OW = PB_W / 2 + 1;
template <typename T>
__global__ __launch_bounds__(256) void maxpool3d_bwd_tiled333s2_kernel(gca_pool_geom g, const T* __restrict__ dy,
                                                                     const int* __restrict__ argmax, T* __restrict__ dx,
                                                                     int accumulate, int nbd, int nbh, int nbw) {
  __shared__ int am_s[PB_OD][PB_OH][PB_OW];
  __shared__ float dy_s[PB_OD][PB_OH][PB_OW];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tbw = b % nbw; b /= nbw;
  const int tbh = b % nbh; b /= nbh;
  const int tbd = b % nbd;
  const int plane = b / nbd;
  const int d0 = tbd * PB_D, h0 = tbh * PB_H, w0 = tbw * PB_W;
  const int od0 = d0 / 2, oh0 = h0 / 2, ow0 = w0 / 2;             // first window that can cover the box (window o covers 2o-1 .. 2o+1)
  const long long obase = (long long)plane * ((long long)g.OD * g.OH * g.OW);
  if (tid < PB_OD * PB_OH * PB_OW) {
    const int zd = tid / (PB_OH * PB_OW), r = tid - zd * (PB_OH * PB_OW), zh = r / PB_OW, zw = r - zh * PB_OW;
    const int od = od0 + zd, oh = oh0 + zh, ow = ow0 + zw;
    int a = -1;
    float v = 0.f;
    if (od < g.OD && oh < g.OH && ow < g.OW) {
      const long long o = obase + ((long long)od * g.OH + oh) * g.OW + ow;
      a = argmax[o];
      v = (float)dy[o];
    }
    am_s[zd][zh][zw] = a;
    dy_s[zd][zh][zw] = v;
  }
  __syncthreads();
  const int w4 = (tid & 3) * 4, lh = (tid >> 2) & 15, ld = tid >> 6;
  const int d = d0 + ld, h = h0 + lh, w = w0 + w4;
  if (d >= g.D || h >= g.H || w >= g.W) return;                  // (W % 4 == 0: a group of four is all inside or all outside)
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // window o covers i  <=>  2o - 1 <= i <= 2o + 1:  i even -> o = i/2 only, i odd -> o in {(i-1)/2, (i+1)/2}
  const int od_first = (d + 1) / 2 - (d & 1), od_last = (d + 1) / 2;
  const int oh_first = (h + 1) / 2 - (h & 1), oh_last = (h + 1) / 2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int wi = w + e;
    const int s_idx = (d * g.H + h) * g.W + wi;
    const int ow_first = (wi + 1) / 2 - (wi & 1), ow_last = (wi + 1) / 2;
    float a = 0.f;
    for (int od = od_first; od <= od_last; ++od)
      for (int oh = oh_first; oh <= oh_last; ++oh)
        for (int ow = ow_first; ow <= ow_last; ++ow) {
          if (od >= g.OD || oh >= g.OH || ow >= g.OW) continue;
          if (am_s[od - od0][oh - oh0][ow - ow0] == s_idx) a += dy_s[od - od0][oh - oh0][ow - ow0];
        }
    acc[e] = a;
  }
  T* out = dx + (long long)plane * ((long long)g.D * g.H * g.W) + (long long)(d * g.H + h) * g.W + w;
  if (accumulate) {
    const float4 o = gca_act<T>::ld4(out);
    acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
  }
  gca_act<T>::st4(out, make_float4(acc[0], acc[1], acc[2], acc[3]));
}

// y[p] = norm * sum_d wt[d] * sum_hw x[p,d,hw]; one wave per (n,c) plane.
template <typename T>
__global__ __launch_bounds__(256) void wavgpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ wt,
                                                           float norm, long long NC, int D, int HW,
                                                           float* __restrict__ y) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= NC) return;
  const int lane = threadIdx.x & 63;
  const T* xp = x + p * ((long long)D * HW);
  float s = 0.f;
  for (int d = 0; d < D; ++d) {
    float t = 0.f;
    for (int i = lane; i < HW; i += 64) t += (float)xp[(long long)d * HW + i];
    s += (wt ? wt[d] : 1.f) * t;
  }
  s = gca_wave_sum(s);
  if (lane == 0) y[p] = s * norm;
}

template <typename T>
__global__ __launch_bounds__(256) void wavgpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ wt,
                                                           float norm, long long total, int D, int HW,
                                                           T* __restrict__ dx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long p = i / ((long long)D * HW);
  const int d = (int)((i / HW) % D);
  dx[i] = (T)(dy[p] * norm * (wt ? wt[d] : 1.f));
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

// GCA_POOL_TILED=0: the per-output kernels for every pool (A/B runs)
static bool pool_tiled_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("GCA_POOL_TILED"); on = (e && e[0] == '0') ? 0 : 1; }
  return on != 0;
}

static PoolMagic pool_magic(const gca_pool_geom* g) {
  PoolMagic q;
  q.osp = gca_make_magic((unsigned)(g->OD * g->OH * g->OW)); q.ohw = gca_make_magic((unsigned)(g->OH * g->OW));
  q.ow = gca_make_magic((unsigned)g->OW);
  q.sp = gca_make_magic((unsigned)(g->D * g->H * g->W)); q.hw = gca_make_magic((unsigned)(g->H * g->W));
  q.w = gca_make_magic((unsigned)g->W);
  q.c = gca_make_magic((unsigned)g->C);
  q.sd = gca_make_magic((unsigned)g->sd); q.sh = gca_make_magic((unsigned)g->sh); q.sw = gca_make_magic((unsigned)g->sw);
  return q;
}

int gca_maxpool3d_fwd(const gca_pool_geom* g, const void* x, void* y, int32_t* argmax, const float* scale,
                      const float* shift, int act_f16, void* stream) {
  if (!pool_ok(g) || !x || !y || ((scale == nullptr) != (shift == nullptr))) return GCA_EINVAL;
  const long long total = (long long)g->N * g->C * g->OD * g->OH * g->OW;
  if (total >= (1LL << 31) || (long long)g->D * g->H * g->W >= (1LL << 31)) return GCA_EINVAL;
  const PoolMagic q = pool_magic(g);
  const dim3 grid((unsigned)gca_ceil_div(total, 256));
  hipStream_t st = (hipStream_t)stream;
#define GCA_POOL_FWD(KD, KH, KW)                                                                                              \
  do {                                                                                                                      \
    if (act_f16)                                                                                                            \
      hipLaunchKernelGGL((maxpool3d_fwd_kernel<gca_half, KD, KH, KW>), grid, dim3(256), 0, st, *g, q, (const gca_half*)x,   \
                         (gca_half*)y, argmax, (unsigned)total, scale, shift);                                              \
    else                                                                                                                    \
      hipLaunchKernelGGL((maxpool3d_fwd_kernel<float, KD, KH, KW>), grid, dim3(256), 0, st, *g, q, (const float*)x,         \
                         (float*)y, argmax, (unsigned)total, scale, shift);                                                 \
  } while (0)
  if (g->kd == 3 && g->kh == 3 && g->kw == 3) GCA_POOL_FWD(3, 3, 3);
  else if (g->kd == 1 && g->kh == 3 && g->kw == 3) GCA_POOL_FWD(1, 3, 3);
  else if (g->kd == 2 && g->kh == 2 && g->kw == 2) GCA_POOL_FWD(2, 2, 2);
  else GCA_POOL_FWD(0, 0, 0);
#undef GCA_POOL_FWD
  return gca_launch_status();
}

int gca_maxpool3d_bwd(const gca_pool_geom* g, const void* dy, const int32_t* argmax, void* dx,
                      int accumulate, int act_f16, void* stream) {
  if (!pool_ok(g) || !dy || !argmax || !dx) return GCA_EINVAL;
  const long long total = (long long)g->N * g->C * g->D * g->H * g->W;
  if (total >= (1LL << 31)) return GCA_EINVAL;
  const PoolMagic q = pool_magic(g);
  const dim3 grid((unsigned)gca_ceil_div(total, 256));
  hipStream_t st = (hipStream_t)stream;
  const int cd = (int)gca_ceil_div(g->kd, g->sd), ch = (int)gca_ceil_div(g->kh, g->sh), cw = (int)gca_ceil_div(g->kw, g->sw);
#define GCA_POOL_BWD(CD, CH, CW)                                                                                             \
  do {                                                                                                                     \
    if (act_f16)                                                                                                           \
      hipLaunchKernelGGL((maxpool3d_bwd_kernel<gca_half, CD, CH, CW>), grid, dim3(256), 0, st, *g, q, (const gca_half*)dy, \
                         argmax, (gca_half*)dx, (unsigned)total, accumulate ? 1 : 0);                                      \
    else                                                                                                                   \
      hipLaunchKernelGGL((maxpool3d_bwd_kernel<float, CD, CH, CW>), grid, dim3(256), 0, st, *g, q, (const float*)dy,       \
                         argmax, (float*)dx, (unsigned)total, accumulate ? 1 : 0);                                         \
  } while (0)
  if (g->kd == 3 && g->kh == 3 && g->kw == 3 && g->sd == 2 && g->sh == 2 && g->sw == 2 && g->pd == 1 && g->ph == 1 && g->pw == 1 &&
      g->W % 4 == 0 && ((long long)g->D * g->H * g->W) % 4 == 0 && ((uintptr_t)dx % 16) == 0 && pool_tiled_on()) {
    const int nbd = (int)gca_ceil_div(g->D, PB_D), nbh = (int)gca_ceil_div(g->H, PB_H), nbw = (int)gca_ceil_div(g->W, PB_W);
    const long long nblk = (long long)g->N * g->C * nbd * nbh * nbw;
    if (nblk <= 0x7fffffffLL) {
      if (act_f16)
        hipLaunchKernelGGL(maxpool3d_bwd_tiled333s2_kernel<gca_half>, dim3((unsigned)nblk), dim3(256), 0, st, *g, (const gca_half*)dy,
                           argmax, (gca_half*)dx, accumulate ? 1 : 0, nbd, nbh, nbw);
      else
        hipLaunchKernelGGL(maxpool3d_bwd_tiled333s2_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, st, *g, (const float*)dy,
                           argmax, (float*)dx, accumulate ? 1 : 0, nbd, nbh, nbw);
      return gca_launch_status();
    }
  }
  if (cd == 2 && ch == 2 && cw == 2) GCA_POOL_BWD(2, 2, 2);
  else if (cd == 1 && ch == 2 && cw == 2) GCA_POOL_BWD(1, 2, 2);
  else if (cd == 3 && ch == 3 && cw == 3) GCA_POOL_BWD(3, 3, 3);
  else if (cd == 1 && ch == 1 && cw == 1) GCA_POOL_BWD(1, 1, 1);
  else GCA_POOL_BWD(0, 0, 0);
#undef GCA_POOL_BWD
  return gca_launch_status();
}

int gca_wavgpool_fwd(const void* x, const float* wt, float norm, int64_t NC, int64_t D, int64_t HW,
                     float* y, int x_f16, void* stream) {
  if (!x || !y || NC <= 0 || D <= 0 || HW <= 0) return GCA_EINVAL;
  if (x_f16)
    hipLaunchKernelGGL(wavgpool_fwd_kernel<gca_half>, dim3((unsigned)gca_ceil_div(NC, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const gca_half*)x, wt, norm, (long long)NC, (int)D, (int)HW, y);
  else
    hipLaunchKernelGGL(wavgpool_fwd_kernel<float>, dim3((unsigned)gca_ceil_div(NC, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, wt, norm, (long long)NC, (int)D, (int)HW, y);
  return gca_launch_status();
}

int gca_wavgpool_bwd(const float* dy, const float* wt, float norm, int64_t NC, int64_t D, int64_t HW,
                     void* dx, int x_f16, void* stream) {
  if (!dy || !dx || NC <= 0 || D <= 0 || HW <= 0) return GCA_EINVAL;
  const long long total = (long long)NC * D * HW;
  if (x_f16)
    hipLaunchKernelGGL(wavgpool_bwd_kernel<gca_half>, dim3((unsigned)gca_ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, wt, norm, total, (int)D, (int)HW, (gca_half*)dx);
  else
    hipLaunchKernelGGL(wavgpool_bwd_kernel<float>, dim3((unsigned)gca_ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, wt, norm, total, (int)D, (int)HW, (float*)dx);
  return gca_launch_status();
}

}  // extern "C"
