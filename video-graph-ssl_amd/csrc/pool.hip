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
  const int safe = (max(d0, 0) * g.H + max(h0, 0)) * g.W + max(w0, 0);   // first in-bounds tap of the window
  int bi = safe;
  if (KD > 0) {
    // every tap's address is known up front (out-of-window taps read `safe`, never the running argmax: that would chain
    // each load behind the previous compare), so the KD*KH*KW loads are issued together and the compares follow
    float v[KD * KH * KW > 0 ? KD * KH * KW : 1];
#pragma unroll
    for (int a = 0; a < KD; ++a)
#pragma unroll
      for (int b = 0; b < KH; ++b)
#pragma unroll
        for (int c = 0; c < KW; ++c) {
          const int d = d0 + a, h = h0 + b, w = w0 + c;
          const bool ok = (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
          v[(a * KH + b) * KW + c] = (float)xp[ok ? (d * g.H + h) * g.W + w : safe];
        }
#pragma unroll
    for (int a = 0; a < KD; ++a)
#pragma unroll
      for (int b = 0; b < KH; ++b)
#pragma unroll
        for (int c = 0; c < KW; ++c) {
          const int d = d0 + a, h = h0 + b, w = w0 + c;
          const bool ok = (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
          const float t = GCA_POOL_VAL(v[(a * KH + b) * KW + c]);
          if (ok && (t > best || isnan(t))) { best = t; bi = (d * g.H + h) * g.W + w; }
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

// 3x3x3 / stride 2 / pad 1 with W % 4 == 0 (every stem pool of the model zoo): one thread per PAIR of outputs adjacent in W.
// The pair's windows span columns 4j-1 .. 4j+3, i.e. one aligned 4-element load per (d,h) row plus the left neighbour,
// so a wave reads whole contiguous rows (18 loads per two outputs instead of 54 stride-2 ones), all issued before the
// first compare.  Same tie break / NaN rule and the same fused BatchNorm+ReLU producer as maxpool3d_fwd_kernel.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3d_fwd_pair333s2_kernel(gca_pool_geom g, gca_magic psp, gca_magic phw,
                                                                      gca_magic pw2, gca_magic cm, const T* __restrict__ x,
                                                                      T* __restrict__ y, int* __restrict__ argmax,
                                                                      unsigned npairs, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift) {
  typedef gca_act<T> A_;
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= npairs) return;
  const unsigned plane = gca_fdiv(p, psp), o = p - plane * psp.d;
  float sc = 1.f, sf = 0.f;
  if (scale) { const unsigned c = plane - gca_fdiv(plane, cm) * cm.d; sc = scale[c]; sf = shift[c]; }
#define GCA_POOL_VAL(v) (scale ? fmaxf((v) * sc + sf, 0.f) : (v))
  const int od = (int)gca_fdiv(o, phw), r = (int)(o - (unsigned)od * phw.d);
  const int oh = (int)gca_fdiv((unsigned)r, pw2), j = r - oh * (int)pw2.d;
  const int d0 = 2 * od - 1, h0 = 2 * oh - 1, wq = 4 * j, wl = j > 0 ? wq - 1 : wq;
  const T* xp = x + (long long)plane * ((long long)g.D * g.H * g.W);
  const int safe_row = (max(d0, 0) * g.H + max(h0, 0)) * g.W;
  float4 m[9];
  float l[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int d = d0 + a, h = h0 + b;
      const bool ok = (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H;
      const int row = ok ? (d * g.H + h) * g.W : safe_row;
      m[a * 3 + b] = A_::ld4(xp + row + wq);
      l[a * 3 + b] = A_::ld(xp + row + wl);
    }
  float b0 = -INFINITY, b1 = -INFINITY;
  int i0 = safe_row + wl, i1 = safe_row + wq + 1;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int d = d0 + a, h = h0 + b;
      const bool ok = (unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H;
      const int row = (d * g.H + h) * g.W + wq;
      const float vl = GCA_POOL_VAL(l[a * 3 + b]), v0 = GCA_POOL_VAL(m[a * 3 + b].x), v1 = GCA_POOL_VAL(m[a * 3 + b].y),
                  v2 = GCA_POOL_VAL(m[a * 3 + b].z), v3 = GCA_POOL_VAL(m[a * 3 + b].w);
      if (ok && j > 0 && (vl > b0 || isnan(vl))) { b0 = vl; i0 = row - 1; }
      if (ok && (v0 > b0 || isnan(v0))) { b0 = v0; i0 = row; }
      if (ok && (v1 > b0 || isnan(v1))) { b0 = v1; i0 = row + 1; }
      if (ok && (v1 > b1 || isnan(v1))) { b1 = v1; i1 = row + 1; }
      if (ok && (v2 > b1 || isnan(v2))) { b1 = v2; i1 = row + 2; }
      if (ok && (v3 > b1 || isnan(v3))) { b1 = v3; i1 = row + 3; }
    }
#undef GCA_POOL_VAL
  if (sizeof(T) == 4) {
    *reinterpret_cast<float2*>(y + 2ull * p) = make_float2(b0, b1);
  } else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 ov; ov.x = (_Float16)b0; ov.y = (_Float16)b1;
    *reinterpret_cast<h2*>(y + 2ull * p) = ov;
  }
  if (argmax) *reinterpret_cast<int2*>(argmax + 2ull * p) = make_int2(i0, i1);
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

// Backward of the 3x3x3 / stride 2 / pad 1 pools behind the R(2+1)D and 3D-ResNet stems (resnet2p1d.py:178, resnet.py:127)
// with W % 4 == 0.  Window o covers inputs 2o-1 .. 2o+1, so the 2 x 2 x 4 input brick d in {2n, 2n+1}, h in {2m, 2m+1},
// w in 4j .. 4j+3 can only be the argmax of windows od in {n, n+1} x oh in {m, m+1} x ow in {2j, 2j+1, 2j+2}: one thread
// fetches those twelve (argmax, dy) pairs (an aligned pair + one scalar per (od,oh), all issued before the first compare;
// invalid slots read a clamped, valid address and are masked) and finishes the brick's sixteen elements with four 16-byte
// stores.  Summation order (od, oh, ow) ascending as in maxpool3d_bwd_kernel: deterministic, no atomics; 32-bit index
// arithmetic (the host checks total < 2^31).  Earlier forms on the (32,64,16,56,56) stem map: per element 0.48 ms, an
// LDS-staged 4x16x16 box 0.32 ms, one thread per 4 elements with the same twelve loads 0.30 ms (one exposed round trip per
// wave: 400 K short waves).  The same LDS tiling of the FORWARD pool was measured at 0.35 ms and dropped for the
// paired-output kernel above.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3d_bwd_brick333s2_kernel(gca_pool_geom g, gca_magic msp, gca_magic mhw, gca_magic mw,
                                                                       const T* __restrict__ dy, const int* __restrict__ argmax,
                                                                       T* __restrict__ dx, unsigned nbricks, int accumulate) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nbricks) return;
  const unsigned plane = gca_fdiv(t, msp), sidx = t - plane * msp.d;
  const int n = (int)gca_fdiv(sidx, mhw), r = (int)(sidx - (unsigned)n * mhw.d);
  const int m = (int)gca_fdiv((unsigned)r, mw), j = r - m * (int)mw.d;
  const bool vd1 = n + 1 < g.OD, vh1 = m + 1 < g.OH, vw2 = 2 * j + 2 < g.OW;           // (n < OD = ceil(D/2), m < OH always)
  const unsigned o00 = ((plane * (unsigned)g.OD + (unsigned)n) * (unsigned)g.OH + (unsigned)m) * (unsigned)g.OW + 2u * (unsigned)j;
  const unsigned step_d = vd1 ? (unsigned)(g.OH * g.OW) : 0u, step_h = vh1 ? (unsigned)g.OW : 0u, step_w = vw2 ? 2u : 1u;
  int2 a01[4];
  int a2[4];
  float g0[4], g1[4], g2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {                                   // c = 2*(od - n) + (oh - m)
    const unsigned o = o00 + ((c & 2) ? step_d : 0u) + ((c & 1) ? step_h : 0u);
    a01[c] = *reinterpret_cast<const int2*>(argmax + o);
    a2[c] = argmax[o + step_w];
    if (sizeof(T) == 4) {
      const float2 v = *reinterpret_cast<const float2*>(dy + o);
      g0[c] = v.x; g1[c] = v.y;
    } else {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 v = *reinterpret_cast<const h2*>(dy + o);
      g0[c] = (float)v.x; g1[c] = (float)v.y;
    }
    g2[c] = (float)dy[o + step_w];
  }
  // masked slots can match nothing
  if (!vw2) { a2[0] = a2[1] = a2[2] = a2[3] = -1; }
  if (!vd1) { a01[2] = a01[3] = make_int2(-1, -1); a2[2] = a2[3] = -1; }
  if (!vh1) { a01[1] = a01[3] = make_int2(-1, -1); a2[1] = a2[3] = -1; }
  const unsigned plane_base = plane * (unsigned)(g.D * g.H * g.W);
#pragma unroll
  for (int dd = 0; dd < 2; ++dd)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int d = 2 * n + dd, h = 2 * m + hh;
      if (d >= g.D || h >= g.H) continue;
      const int s0 = (d * g.H + h) * g.W + 4 * j;                 // plane-relative index of the quad's first element
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cd = 0; cd <= dd; ++cd)                            // even d / h lie in one window only, odd ones in two
#pragma unroll
        for (int ch = 0; ch <= hh; ++ch) {
          const int c = 2 * cd + ch;
          // window 2j covers w 4j-1..4j+1, window 2j+1 covers 4j+1..4j+3, window 2j+2 covers 4j+3..4j+5
          acc[0] += a01[c].x == s0 ? g0[c] : 0.f;
          acc[1] += a01[c].x == s0 + 1 ? g0[c] : 0.f;
          acc[1] += a01[c].y == s0 + 1 ? g1[c] : 0.f;
          acc[2] += a01[c].y == s0 + 2 ? g1[c] : 0.f;
          acc[3] += a01[c].y == s0 + 3 ? g1[c] : 0.f;
          acc[3] += a2[c] == s0 + 3 ? g2[c] : 0.f;
        }
      T* out = dx + (plane_base + (unsigned)s0);
      if (accumulate) {
        const float4 o = gca_act<T>::ld4(out);
        acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
      }
      gca_act<T>::st4(out, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
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


// nn.AvgPool3d(kernel_size=k) with its default stride = k and no padding (temporal_graph.py:100, the max_pool=False option
// of TemporalGraphAug): windows do not overlap, trailing rows / columns that do not fill a window are dropped (floor mode).
__global__ __launch_bounds__(256) void avgpool3d_fwd_kernel(gca_pool_geom g, PoolMagic q, const float* __restrict__ x,
                                                            float* __restrict__ y, unsigned total) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned plane = gca_fdiv(i, q.osp), o = i - plane * q.osp.d;
  const int od = (int)gca_fdiv(o, q.ohw), r = (int)(o - (unsigned)od * q.ohw.d);
  const int oh = (int)gca_fdiv((unsigned)r, q.ow), ow = r - oh * g.OW;
  const float* xp = x + (long long)plane * ((long long)g.D * g.H * g.W);
  float s = 0.f;
  for (int a = 0; a < g.kd; ++a)
    for (int b = 0; b < g.kh; ++b)
      for (int c = 0; c < g.kw; ++c) s += xp[((od * g.kd + a) * g.H + oh * g.kh + b) * g.W + ow * g.kw + c];
  y[i] = s / (float)(g.kd * g.kh * g.kw);
}
__global__ __launch_bounds__(256) void avgpool3d_bwd_kernel(gca_pool_geom g, PoolMagic q, const float* __restrict__ dy,
                                                            float* __restrict__ dx, unsigned total, int accumulate) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;          // one thread per INPUT element: at most one window covers it
  if (i >= total) return;
  const unsigned plane = gca_fdiv(i, q.sp), e = i - plane * q.sp.d;
  const int d = (int)gca_fdiv(e, q.hw), r = (int)(e - (unsigned)d * q.hw.d);
  const int h = (int)gca_fdiv((unsigned)r, q.w), w = r - h * g.W;
  const int od = d / g.kd, oh = h / g.kh, ow = w / g.kw;
  float v = 0.f;
  if (od < g.OD && oh < g.OH && ow < g.OW)
    v = dy[(long long)plane * ((long long)g.OD * g.OH * g.OW) + ((long long)od * g.OH + oh) * g.OW + ow] / (float)(g.kd * g.kh * g.kw);
  dx[i] = accumulate ? dx[i] + v : v;
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
  const size_t esz = act_f16 ? 2 : 4;
  if (g->kd == 3 && g->kh == 3 && g->kw == 3 && g->sd == 2 && g->sh == 2 && g->sw == 2 && g->pd == 1 && g->ph == 1 && g->pw == 1 &&
      g->W % 4 == 0 && ((uintptr_t)x % (4 * esz)) == 0 && ((uintptr_t)y % (2 * esz)) == 0 && ((uintptr_t)argmax % 8) == 0 &&
      pool_tiled_on()) {
    // OW = W/2 is even here, so output pairs never straddle a row and pair p covers outputs 2p, 2p+1
    const unsigned npairs = (unsigned)(total / 2);
    const gca_magic psp = gca_make_magic((unsigned)(g->OD * g->OH * (g->OW / 2))), phw = gca_make_magic((unsigned)(g->OH * (g->OW / 2))),
                    pw2 = gca_make_magic((unsigned)(g->OW / 2));
    const dim3 pgrid((unsigned)gca_ceil_div((long long)npairs, 256));
    if (act_f16)
      hipLaunchKernelGGL(maxpool3d_fwd_pair333s2_kernel<gca_half>, pgrid, dim3(256), 0, st, *g, psp, phw, pw2, q.c, (const gca_half*)x,
                         (gca_half*)y, argmax, npairs, scale, shift);
    else
      hipLaunchKernelGGL(maxpool3d_fwd_pair333s2_kernel<float>, pgrid, dim3(256), 0, st, *g, psp, phw, pw2, q.c, (const float*)x,
                         (float*)y, argmax, npairs, scale, shift);
    return gca_launch_status();
  }
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
      g->W % 4 == 0 && ((uintptr_t)dx % (act_f16 ? 8 : 16)) == 0 && ((uintptr_t)dy % (act_f16 ? 4 : 8)) == 0 &&
      ((uintptr_t)argmax % 8) == 0 && pool_tiled_on()) {
    // OW = W/2 is even: the (argmax, dy) pair of windows 2j, 2j+1 is aligned
    const int d2 = (g->D + 1) / 2, h2 = (g->H + 1) / 2, w4 = g->W / 4;
    const unsigned nbricks = (unsigned)((long long)g->N * g->C * d2 * h2 * w4);
    const gca_magic msp = gca_make_magic((unsigned)(d2 * h2 * w4)), mhw = gca_make_magic((unsigned)(h2 * w4)),
                    mw = gca_make_magic((unsigned)w4);
    const dim3 bgrid((unsigned)gca_ceil_div((long long)nbricks, 256));
    if (act_f16)
      hipLaunchKernelGGL(maxpool3d_bwd_brick333s2_kernel<gca_half>, bgrid, dim3(256), 0, st, *g, msp, mhw, mw, (const gca_half*)dy,
                         argmax, (gca_half*)dx, nbricks, accumulate ? 1 : 0);
    else
      hipLaunchKernelGGL(maxpool3d_bwd_brick333s2_kernel<float>, bgrid, dim3(256), 0, st, *g, msp, mhw, mw, (const float*)dy,
                         argmax, (float*)dx, nbricks, accumulate ? 1 : 0);
    return gca_launch_status();
  }
  if (cd == 2 && ch == 2 && cw == 2) GCA_POOL_BWD(2, 2, 2);
  else if (cd == 1 && ch == 2 && cw == 2) GCA_POOL_BWD(1, 2, 2);
  else if (cd == 3 && ch == 3 && cw == 3) GCA_POOL_BWD(3, 3, 3);
  else if (cd == 1 && ch == 1 && cw == 1) GCA_POOL_BWD(1, 1, 1);
  else GCA_POOL_BWD(0, 0, 0);
#undef GCA_POOL_BWD
  return gca_launch_status();
}

static bool avgpool_ok(const gca_pool_geom* g) {
  return pool_ok(g) && g->sd == g->kd && g->sh == g->kh && g->sw == g->kw && g->pd == 0 && g->ph == 0 && g->pw == 0;
}
int gca_avgpool3d_fwd(const gca_pool_geom* g, const float* x, float* y, void* stream) {
  if (!avgpool_ok(g) || !x || !y) return GCA_EINVAL;
  const long long total = (long long)g->N * g->C * g->OD * g->OH * g->OW;
  if (total >= (1LL << 31) || (long long)g->D * g->H * g->W >= (1LL << 31)) return GCA_EINVAL;
  hipLaunchKernelGGL(avgpool3d_fwd_kernel, dim3((unsigned)gca_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, *g,
                     pool_magic(g), x, y, (unsigned)total);
  return gca_launch_status();
}
int gca_avgpool3d_bwd(const gca_pool_geom* g, const float* dy, float* dx, int accumulate, void* stream) {
  if (!avgpool_ok(g) || !dy || !dx) return GCA_EINVAL;
  const long long total = (long long)g->N * g->C * g->D * g->H * g->W;
  if (total >= (1LL << 31)) return GCA_EINVAL;
  hipLaunchKernelGGL(avgpool3d_bwd_kernel, dim3((unsigned)gca_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, *g,
                     pool_magic(g), dy, dx, (unsigned)total, accumulate ? 1 : 0);
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
