// Training-mode BatchNorm (3d / 1d) fused with ReLU and the residual add, gfx950.
// All of it is HBM-bound streaming: float4 accesses, one pass per tensor, per-channel partial
// sums finished in fp64 so the batch statistics do not depend on the tile decomposition.
// Reference call sites: see include/gca_hip.h (BatchNorm section).
#include "gca_common.h"

namespace {

constexpr int STAT_CHUNK = 8192;     // elements of one channel handled per stats block
constexpr int MAX_PARTS = 256;
constexpr long long BN_SMALL_ELEMS = 32768;   // N*SP at or below this: forward / backward of a channel in one workgroup

__host__ __device__ inline long long stats_parts(long long N, long long C, long long SP) {
  long long p = (N * SP + STAT_CHUNK - 1) / STAT_CHUNK;
  if (p < 1) p = 1;
  if (p > MAX_PARTS) p = MAX_PARTS;
  return p;
}

// grid (P, C): block (part, c) reduces a contiguous slice of channel c's N*SP elements.
// `f(i_global)` style is avoided: we walk (n, sp) so reads stay contiguous inside a plane.
// VEC = 4 (SP % 4 == 0, 16-byte aligned operands): float4 loads, two of them in flight per tensor.
template <typename T, int MODE, int VEC>   // MODE 0: sum x, x^2     1: sum dz, dz*xhat (backward)
__global__ __launch_bounds__(256) void bn_reduce_kernel(
    const T* __restrict__ a, const T* __restrict__ z, const T* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
    long long N, long long C, long long SP, long long zs, int P, float* __restrict__ out0,
    float* __restrict__ out1, const float* __restrict__ scale, const float* __restrict__ shift) {
  typedef gca_act<T> A_;
  __shared__ double sh[4];
  const int c = blockIdx.y, part = blockIdx.x;
  // relu == 2: the ReLU mask is recomputed from the conv output exactly as bn_apply computed it (x*scale+shift > 0)
  // instead of being read back from z -- one tensor less to stream (valid when nothing was added before the ReLU)
  const float rsc = relu == 2 ? scale[c] : 0.f, rsf = relu == 2 ? shift[c] : 0.f;
  const long long M = N * SP;
  long long per = (M + P - 1) / P;
  if (VEC == 4) per = (per + 3) & ~3LL;              // slices start on float4 boundaries
  const long long lo = part * per;
  long long hi = lo + per; if (hi > M) hi = M;
  // fp64 accumulators: these sums cancel heavily (a BN output gradient is nearly zero-mean), and ATen's CPU
  // batch norm accumulates in double too.  The kernel stays HBM-bound (3 DP ops per element).
  double s0 = 0.0, s1 = 0.0;
  float mu = 0.f, is = 0.f;
  if (MODE == 1) { mu = mean[c]; is = invstd[c]; }
  if (VEC == 4) {
    auto accum = [&](const float4 av, const float4 zv, const float4 xv) __attribute__((always_inline)) {
      const float ae[4] = {av.x, av.y, av.z, av.w}, ze[4] = {zv.x, zv.y, zv.z, zv.w}, xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (MODE == 0) { const double v = (double)ae[e]; s0 += v; s1 += v * v; }
        else {
          float dz = ae[e];
          if (relu == 1 && !(ze[e] > 0.f)) dz = 0.f;
          if (relu == 2 && !(xe[e] * rsc + rsf > 0.f)) dz = 0.f;
          s0 += (double)dz; s1 += (double)dz * (double)((xe[e] - mu) * is);
        }
      }
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = lo + 4LL * threadIdx.x; i < hi; i += 2048) {
      const long long i2 = i + 1024;
      const bool two = i2 < hi;
      const long long n = i / SP, sp = i - n * SP;
      const long long n2 = two ? i2 / SP : n, sp2 = two ? i2 - n2 * SP : sp;
      const long long idx = (n * C + c) * SP + sp, idx2 = (n2 * C + c) * SP + sp2;
      const long long zi = n * zs + c * SP + sp, zi2 = n2 * zs + c * SP + sp2;
      float4 a0, a1, z0 = zero4, z1 = zero4, x0 = zero4, x1 = zero4;
      if (MODE == 0) {
        a0 = A_::ld4(a + idx);
        a1 = two ? A_::ld4(a + idx2) : zero4;
      } else {
        a0 = A_::ld4(a + zi);
        a1 = two ? A_::ld4(a + zi2) : zero4;
        if (relu == 1) { z0 = A_::ld4(z + zi); if (two) z1 = A_::ld4(z + zi2); }
        x0 = A_::ld4(x + idx);
        x1 = two ? A_::ld4(x + idx2) : make_float4(mu, mu, mu, mu);
      }
      accum(a0, z0, x0);
      if (two) accum(a1, z1, x1);
    }
  } else {
  const long long i0 = lo + threadIdx.x;
  const bool walk = SP >= 256;                      // big planes: one division, then walk incrementally
  long long n = i0 / SP, sp = i0 - n * SP;
  for (long long i = i0; i < hi; i += 256, sp += 256) {
    if (walk) { while (sp >= SP) { sp -= SP; ++n; } }
    else { n = i / SP; sp = i - n * SP; }
    const long long idx = (n * C + c) * SP + sp;
    if (MODE == 0) {
      const double v = (double)A_::ld(a + idx);
      s0 += v; s1 += v * v;
    } else {
      const long long zidx = n * zs + c * SP + sp;       // dz / z may be channel slices of a concat buffer
      float dz = A_::ld(a + zidx);
      const float xv = A_::ld(x + idx);
      if (relu == 1 && !(A_::ld(z + zidx) > 0.f)) dz = 0.f;
      if (relu == 2 && !(xv * rsc + rsf > 0.f)) dz = 0.f;
      s0 += (double)dz; s1 += (double)dz * (double)((xv - mu) * is);
    }
  }
  }
  s0 = gca_block_sum256_d(s0, sh);
  s1 = gca_block_sum256_d(s1, sh);
  if (threadIdx.x == 0) {
    out0[(long long)c * P + part] = (float)s0;
    out1[(long long)c * P + part] = (float)s1;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const float* __restrict__ psum, const float* __restrict__ psq, long long P, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ smean,
    float* __restrict__ sinvstd, float* __restrict__ scale, float* __restrict__ shift, long long* __restrict__ nbt) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  if (nbt && c == 0 && threadIdx.x == 0) *nbt += 1;          // num_batches_tracked (BatchNorm bookkeeping)
  double s = 0.0, q = 0.0;
  {
    // eight partials of each array in flight per thread (P reaches several thousand on the big layers: one dependent load per
    // trip was 20+ exposed round trips); the accumulation order per thread is unchanged
    const float* ps = psum + c * P;
    const float* pq = psq + c * P;
    long long i = threadIdx.x;
    for (; i + 7 * 256 < P; i += 8 * 256) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = ps[i + u * 256]; b[u] = pq[i + u * 256]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += (double)a[u]; q += (double)b[u]; }
    }
    for (; i < P; i += 256) { s += (double)ps[i]; q += (double)pq[i]; }
  }
  s = gca_block_sum256_d(s, sh);
  q = gca_block_sum256_d(q, sh);
  if (threadIdx.x == 0) {
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    if (smean) smean[c] = (float)m;
    if (sinvstd) sinvstd[c] = (float)is;
    if (rmean) rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * m);
    if (rvar) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * (float)is;
    scale[c] = sc;
    shift[c] = b - (float)m * sc;
  }
}

__global__ void bn_fold_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                    long long C, float* scale, float* shift) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rv[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * is;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

// z = [relu](x*scale[c] + shift[c] [+ res]);  VEC = 4 when SP % 4 == 0 (a float4 never straddles a plane)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const T* __restrict__ res, int relu, long long total, long long C, long long SP, long long zs,
    T* __restrict__ z) {
  typedef gca_act<T> A_;
  const long long stride = (long long)gridDim.x * 256 * VEC;
  const bool small = total < (1LL << 31);           // 32-bit index math (the usual case)
  const long long zskip = zs - C * SP;              // extra elements between samples of z (concat slice)
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC; i < total; i += stride) {
    int c; long long n;
    if (small) { const unsigned q = (unsigned)i / (unsigned)SP; n = q / (unsigned)C; c = (int)(q - (unsigned)n * (unsigned)C); }
    else { const long long q = i / SP; n = q / C; c = (int)(q - n * C); }
    const long long zi = i + n * zskip;
    const float sc = scale[c], sf = shift[c];
    if (VEC == 4) {
      float4 v = A_::ld4(x + i);
      v.x = v.x * sc + sf; v.y = v.y * sc + sf; v.z = v.z * sc + sf; v.w = v.w * sc + sf;
      if (res) { const float4 r = A_::ld4(res + i); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      A_::st4(z + zi, v);
    } else {
      float v = A_::ld(x + i) * sc + sf;
      if (res) v += A_::ld(res + i);
      if (relu) v = fmaxf(v, 0.f);
      A_::st(z + zi, v);
    }
  }
}

// per channel: finish the backward sums, write dgamma/dbeta (+=), coefficient triplet for the apply pass
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    const float* __restrict__ p0, const float* __restrict__ p1, int P, double count,
    const float* __restrict__ gamma, const float* __restrict__ invstd,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef /* [3][C] */, long long C) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  {
    const float* q0 = p0 + (long long)c * P;
    const float* q1 = p1 + (long long)c * P;
    int i = threadIdx.x;
    for (; i + 7 * 256 < P; i += 8 * 256) {          // eight partials of each array in flight (see bn_finalize_kernel)
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = q0[i + u * 256]; b[u] = q1[i + u * 256]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += (double)a[u]; s1 += (double)b[u]; }
    }
    for (; i < P; i += 256) { s0 += (double)q0[i]; s1 += (double)q1[i]; }
  }
  s0 = gca_block_sum256_d(s0, sh);
  s1 = gca_block_sum256_d(s1, sh);
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[c] += (float)s0;
    if (dgamma) dgamma[c] += (float)s1;
    const float g = gamma ? gamma[c] : 1.f;
    coef[c] = g * invstd[c];
    coef[C + c] = (float)(s0 / count);
    coef[2 * C + c] = (float)(s1 / count);
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const T* __restrict__ dzin, const T* __restrict__ z, const T* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ coef,
    int relu, long long total, long long C, long long SP, long long zs, T* __restrict__ dx,
    T* __restrict__ dres, int dres_acc, const float* __restrict__ scale, const float* __restrict__ shift) {
  typedef gca_act<T> A_;
  const long long stride = (long long)gridDim.x * 256 * VEC;
  const bool small = total < (1LL << 31);           // 32-bit index math (the usual case)
  const long long zskip = zs - C * SP;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC; i < total; i += stride) {
    int c; long long n;
    if (small) { const unsigned q = (unsigned)i / (unsigned)SP; n = q / (unsigned)C; c = (int)(q - (unsigned)n * (unsigned)C); }
    else { const long long q = i / SP; n = q / C; c = (int)(q - n * C); }
    const long long zi = i + n * zskip;
    const float A = coef[c], B = coef[C + c], Cc = coef[2 * C + c], mu = mean[c], is = invstd[c];
    const float rsc = relu == 2 ? scale[c] : 0.f, rsf = relu == 2 ? shift[c] : 0.f;
    if (VEC == 4) {
      float4 d = A_::ld4(dzin + zi);
      if (relu == 1) {
        const float4 zz = A_::ld4(z + zi);
        if (!(zz.x > 0.f)) d.x = 0.f; if (!(zz.y > 0.f)) d.y = 0.f; if (!(zz.z > 0.f)) d.z = 0.f; if (!(zz.w > 0.f)) d.w = 0.f;
      }
      const float4 xv = A_::ld4(x + i);
      if (relu == 2) {
        if (!(xv.x * rsc + rsf > 0.f)) d.x = 0.f; if (!(xv.y * rsc + rsf > 0.f)) d.y = 0.f;
        if (!(xv.z * rsc + rsf > 0.f)) d.z = 0.f; if (!(xv.w * rsc + rsf > 0.f)) d.w = 0.f;
      }
      float4 o;
      o.x = A * (d.x - B - (xv.x - mu) * is * Cc);
      o.y = A * (d.y - B - (xv.y - mu) * is * Cc);
      o.z = A * (d.z - B - (xv.z - mu) * is * Cc);
      o.w = A * (d.w - B - (xv.w - mu) * is * Cc);
      A_::st4(dx + i, o);
      if (dres) {
        if (dres_acc) { const float4 r = A_::ld4(dres + i); d.x += r.x; d.y += r.y; d.z += r.z; d.w += r.w; }
        A_::st4(dres + i, d);
      }
    } else {
      float d = A_::ld(dzin + zi);
      const float xv = A_::ld(x + i);
      if (relu == 1 && !(A_::ld(z + zi) > 0.f)) d = 0.f;
      if (relu == 2 && !(xv * rsc + rsf > 0.f)) d = 0.f;
      A_::st(dx + i, A * (d - B - (xv - mu) * is * Cc));
      if (dres) A_::st(dres + i, dres_acc ? A_::ld(dres + i) + d : d);
    }
  }
}

// Training-mode BatchNorm forward of one channel in ONE workgroup (small N*SP): fold the conv epilogue's partial sums
// (fp64, as bn_finalize_kernel), update the running statistics, then normalise the channel (+residual, +ReLU, concat
// slice) -- finalize and apply in one launch.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_fwd_small_kernel(
    const float* __restrict__ psum, const float* __restrict__ psq, long long P, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ smean, float* __restrict__ sinvstd,
    float* __restrict__ scale, float* __restrict__ shift, long long* __restrict__ nbt,
    const T* __restrict__ x, const T* __restrict__ res, int relu, int N, int C, int SP, long long zs,
    T* __restrict__ z) {
  typedef gca_act<T> A_;
  __shared__ double sh[4];
  __shared__ float ab[2];
  const int c = blockIdx.x;
  if (nbt && c == 0 && threadIdx.x == 0) *nbt += 1;
  double s = 0.0, q = 0.0;
  {
    // eight partials of each array in flight per thread (P reaches several thousand on the big layers: one dependent load per
    // trip was 20+ exposed round trips); the accumulation order per thread is unchanged
    const float* ps = psum + c * P;
    const float* pq = psq + c * P;
    long long i = threadIdx.x;
    for (; i + 7 * 256 < P; i += 8 * 256) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = ps[i + u * 256]; b[u] = pq[i + u * 256]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += (double)a[u]; q += (double)b[u]; }
    }
    for (; i < P; i += 256) { s += (double)ps[i]; q += (double)pq[i]; }
  }
  s = gca_block_sum256_d(s, sh);
  q = gca_block_sum256_d(q, sh);
  if (threadIdx.x == 0) {
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    if (smean) smean[c] = (float)m;
    if (sinvstd) sinvstd[c] = (float)is;
    if (rmean) rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * m);
    if (rvar) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * (float)is;
    const float sf = b - (float)m * sc;
    scale[c] = sc; shift[c] = sf;
    ab[0] = sc; ab[1] = sf;
  }
  __syncthreads();
  const float sc = ab[0], sf = ab[1];
  const int per = SP / VEC, total = N * per;
  // UF slots per thread per trip, all loads issued before the first use (clamped index: unconditional loads): a channel of
  // these layers is 6-25 trips of 256 threads, and one exposed L2 round trip per trip was most of the kernel's 11 us
  // (now 9.4; eight slots per trip measured no better: 10.4)
  constexpr int UF = 4;
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * UF) {
    long long xi[UF], zi[UF];
    float4 xv[UF], rv[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int i = min(i0 + 256 * u, total - 1);
      const int n = i / per, sp = (i - n * per) * VEC;
      xi[u] = ((long long)n * C + c) * SP + sp; zi[u] = (long long)n * zs + (long long)c * SP + sp;
      if (VEC == 4) {
        xv[u] = A_::ld4(x + xi[u]);
        if (res) rv[u] = A_::ld4(res + xi[u]);
      } else {
        xv[u].x = A_::ld(x + xi[u]);
        if (res) rv[u].x = A_::ld(res + xi[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      if (i0 + 256 * u >= total) break;
      if (VEC == 4) {
        float4 v = xv[u];
        v.x = v.x * sc + sf; v.y = v.y * sc + sf; v.z = v.z * sc + sf; v.w = v.w * sc + sf;
        if (res) { const float4 r = rv[u]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        A_::st4(z + zi[u], v);
      } else {
        float v = xv[u].x * sc + sf;
        if (res) v += rv[u].x;
        if (relu) v = fmaxf(v, 0.f);
        A_::st(z + zi[u], v);
      }
    }
  }
}

// The same for a conv whose split-K slabs are still un-folded (gca_conv_fwd_slabs): the workgroup of channel c first folds
// slab[s][c][n] over s in the fixed order 0, 1, ... (the bits conv_splitk_finish_kernel would write), stores the conv output
// y -- the backward pass needs it -- and takes the batch statistics from those values in fp64; then finalize and apply as
// above.  One launch instead of three (finish, finalize + apply), and the statistics skip the fp32 partial sums.
__global__ __launch_bounds__(256) void bn_fwd_small_slab_kernel(
    const float* __restrict__ slab, int splits, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ smean, float* __restrict__ sinvstd,
    float* __restrict__ scale, float* __restrict__ shift, long long* __restrict__ nbt,
    float* __restrict__ y, const float* __restrict__ res, int relu, int N, int C, int SP, long long zs,
    float* __restrict__ z) {
  __shared__ double sh[4];
  __shared__ float ab[2];
  const int c = blockIdx.x;
  if (nbt && c == 0 && threadIdx.x == 0) *nbt += 1;
  const int total = N * SP;
  const long long sstride = (long long)C * total;
  const float* s0 = slab + (long long)c * total;
  double s = 0.0, q = 0.0;
  constexpr int UF = 4;
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * UF) {
    float v[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) v[u] = 0.f;
    int k = 0;
    for (; k + 2 <= splits; k += 2) {                 // 2 x UF loads in flight, summed in the fixed order 0, 1, 2, ...
      float a[UF], b[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        const int i = min(i0 + 256 * u, total - 1);
        a[u] = s0[(long long)k * sstride + i];
        b[u] = s0[(long long)(k + 1) * sstride + i];
      }
#pragma unroll
      for (int u = 0; u < UF; ++u) { v[u] += a[u]; v[u] += b[u]; }
    }
    if (k < splits) {
#pragma unroll
      for (int u = 0; u < UF; ++u) v[u] += s0[(long long)k * sstride + min(i0 + 256 * u, total - 1)];
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int i = i0 + 256 * u;
      if (i >= total) break;
      const int n = i / SP, sp = i - n * SP;
      y[((long long)n * C + c) * SP + sp] = v[u];
      s += (double)v[u]; q += (double)v[u] * (double)v[u];
    }
  }
  s = gca_block_sum256_d(s, sh);
  q = gca_block_sum256_d(q, sh);
  if (threadIdx.x == 0) {
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    if (smean) smean[c] = (float)m;
    if (sinvstd) sinvstd[c] = (float)is;
    if (rmean) rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * m);
    if (rvar) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * (float)is;
    const float sf = b - (float)m * sc;
    scale[c] = sc; shift[c] = sf;
    ab[0] = sc; ab[1] = sf;
  }
  __threadfence_block();                               // the y values of this channel are re-read by other threads below
  __syncthreads();
  const float sc = ab[0], sf = ab[1];
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * UF) {
    float xv[UF], rv[UF];
    long long zi[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int i = min(i0 + 256 * u, total - 1);
      const int n = i / SP, sp = i - n * SP;
      const long long xi = ((long long)n * C + c) * SP + sp;
      zi[u] = (long long)n * zs + (long long)c * SP + sp;
      xv[u] = y[xi];
      rv[u] = res ? res[xi] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      if (i0 + 256 * u >= total) break;
      float v = xv[u] * sc + sf + rv[u];
      if (relu) v = fmaxf(v, 0.f);
      z[zi[u]] = v;
    }
  }
}

// Whole BatchNorm backward of one channel in ONE workgroup (small N*SP: the deep, narrow layers and every layer of
// a small batch): pass 1 reduces (sum dz, sum dz*xhat) in fp64, the block derives the coefficients and accumulates
// dgamma / dbeta, pass 2 re-reads the (L2-resident) operands and writes dx (+ the residual gradient).  Replaces
// three launches (reduce, finalize, apply) that cost more in launch latency than in work at these sizes.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(
    const T* __restrict__ dzin, const T* __restrict__ z, const T* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
    int N, int C, int SP, long long zs, T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
    T* __restrict__ dres, int dres_acc, const float* __restrict__ scale, const float* __restrict__ shift) {
  typedef gca_act<T> A_;
  __shared__ double sh[4];
  const int c = blockIdx.x;
  const float mu = mean[c], is = invstd[c];
  const float rsc = relu == 2 ? scale[c] : 0.f, rsf = relu == 2 ? shift[c] : 0.f;
  const int per = SP / VEC, total = N * per;                 // VEC-wide slots of this channel
  double s0 = 0.0, s1 = 0.0;
  // UF slots per thread per trip with every load issued before the first use (clamped index: unconditional loads): one
  // exposed round trip per 256-thread trip was most of this kernel's 17-20 us
  constexpr int UF = 4;
  auto load_slot = [&](int i, long long& xi, long long& zi, float (&dv)[VEC], float (&xv)[VEC], float (&zv)[VEC]) __attribute__((always_inline)) {
    const int n = i / per, sp = (i - n * per) * VEC;
    xi = ((long long)n * C + c) * SP + sp; zi = (long long)n * zs + (long long)c * SP + sp;
    if (VEC == 4) {
      const float4 d4 = A_::ld4(dzin + zi), x4 = A_::ld4(x + xi);
      dv[0] = d4.x; dv[VEC > 1 ? 1 : 0] = d4.y; dv[VEC > 2 ? 2 : 0] = d4.z; dv[VEC > 3 ? 3 : 0] = d4.w;
      xv[0] = x4.x; xv[VEC > 1 ? 1 : 0] = x4.y; xv[VEC > 2 ? 2 : 0] = x4.z; xv[VEC > 3 ? 3 : 0] = x4.w;
      if (relu == 1) { const float4 z4 = A_::ld4(z + zi); zv[0] = z4.x; zv[VEC > 1 ? 1 : 0] = z4.y; zv[VEC > 2 ? 2 : 0] = z4.z; zv[VEC > 3 ? 3 : 0] = z4.w; }
    } else { dv[0] = A_::ld(dzin + zi); xv[0] = A_::ld(x + xi); if (relu == 1) zv[0] = A_::ld(z + zi); }
  };
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * UF) {
    long long xi[UF], zi[UF];
    float dv[UF][VEC], xv[UF][VEC], zv[UF][VEC];
#pragma unroll
    for (int u = 0; u < UF; ++u) load_slot(min(i0 + 256 * u, total - 1), xi[u], zi[u], dv[u], xv[u], zv[u]);
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      if (i0 + 256 * u >= total) break;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float d = dv[u][e];
        if (relu == 1 && !(zv[u][e] > 0.f)) d = 0.f;
        if (relu == 2 && !(xv[u][e] * rsc + rsf > 0.f)) d = 0.f;
        s0 += (double)d; s1 += (double)d * (double)((xv[u][e] - mu) * is);
      }
    }
  }
  s0 = gca_block_sum256_d(s0, sh);
  s1 = gca_block_sum256_d(s1, sh);
  const double count = (double)N * (double)SP;
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[c] += (float)s0;
    if (dgamma) dgamma[c] += (float)s1;
  }
  const float A = (gamma ? gamma[c] : 1.f) * is, B = (float)(s0 / count), Cc = (float)(s1 / count);
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * UF) {
    long long xi[UF], zi[UF];
    float dv[UF][VEC], xv[UF][VEC], zv[UF][VEC];
    float4 ro[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      load_slot(min(i0 + 256 * u, total - 1), xi[u], zi[u], dv[u], xv[u], zv[u]);
      if (dres && dres_acc) { if (VEC == 4) ro[u] = A_::ld4(dres + xi[u]); else ro[u].x = A_::ld(dres + xi[u]); }
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      if (i0 + 256 * u >= total) break;
      float ov[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if (relu == 1 && !(zv[u][e] > 0.f)) dv[u][e] = 0.f;
        if (relu == 2 && !(xv[u][e] * rsc + rsf > 0.f)) dv[u][e] = 0.f;
        ov[e] = A * (dv[u][e] - B - (xv[u][e] - mu) * is * Cc);
      }
      if (VEC == 4) {
        A_::st4(dx + xi[u], make_float4(ov[0], ov[VEC > 1 ? 1 : 0], ov[VEC > 2 ? 2 : 0], ov[VEC > 3 ? 3 : 0]));
        if (dres) {
          float4 r = make_float4(dv[u][0], dv[u][VEC > 1 ? 1 : 0], dv[u][VEC > 2 ? 2 : 0], dv[u][VEC > 3 ? 3 : 0]);
          if (dres_acc) { r.x += ro[u].x; r.y += ro[u].y; r.z += ro[u].z; r.w += ro[u].w; }
          A_::st4(dres + xi[u], r);
        }
      } else {
        A_::st(dx + xi[u], ov[0]);
        if (dres) A_::st(dres + xi[u], dres_acc ? ro[u].x + dv[u][0] : dv[u][0]);
      }
    }
  }
}

inline unsigned ew_grid(long long total, int vec) {
  long long b = gca_ceil_div(total, 256LL * vec);
  if (b > 8192) b = 8192;          // grid-stride the rest
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int gca_bn_finalize(const float* stat_sum, const float* stat_sq, int64_t P, int64_t C, double count,
                               const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float* save_mean, float* save_invstd, float* scale, float* shift, void* stream);

template <typename T>
static int bn_stats_t(const T* x, int64_t N, int64_t C, int64_t SP, float* stat_sum, float* stat_sq,
                 int64_t* parts_out, void* stream) {
  if (!x || !stat_sum || !stat_sq || N <= 0 || C <= 0 || SP <= 0) return GCA_EINVAL;
  const int P = (int)stats_parts(N, C, SP);
  if (parts_out) *parts_out = P;
  if (SP % 4 == 0 && ((uintptr_t)x % 16) == 0)
    hipLaunchKernelGGL((bn_reduce_kernel<T, 0, 4>), dim3(P, (unsigned)C), dim3(256), 0, (hipStream_t)stream, x, (const T*)nullptr,
                       (const T*)nullptr, nullptr, nullptr, 0, (long long)N, (long long)C, (long long)SP, (long long)(C * SP), P,
                       stat_sum, stat_sq, nullptr, nullptr);
  else
    hipLaunchKernelGGL((bn_reduce_kernel<T, 0, 1>), dim3(P, (unsigned)C), dim3(256), 0, (hipStream_t)stream, x, (const T*)nullptr,
                       (const T*)nullptr, nullptr, nullptr, 0, (long long)N, (long long)C, (long long)SP, (long long)(C * SP), P,
                       stat_sum, stat_sq, nullptr, nullptr);
  return gca_launch_status();
}

template <typename T>
static int bn_apply_t(const T* x, const float* scale, const float* shift, const T* residual,
                 int relu, int64_t N, int64_t C, int64_t SP, T* z, int64_t z_batch_stride, void* stream) {
  if (!x || !scale || !shift || !z || N <= 0 || C <= 0 || SP <= 0) return GCA_EINVAL;
  if (z_batch_stride != 0 && z_batch_stride < C * SP) return GCA_EINVAL;
  const long long zs = z_batch_stride ? z_batch_stride : C * SP;
  const long long total = (long long)N * C * SP;
  const bool v4 = (SP % 4 == 0) && (zs % 4 == 0) && (((uintptr_t)x | (uintptr_t)z | (uintptr_t)residual) % 16 == 0);
  if (v4)
    hipLaunchKernelGGL((bn_apply_kernel<T, 4>), dim3(ew_grid(total, 4)), dim3(256), 0, (hipStream_t)stream, x, scale,
                       shift, residual, relu, total, (long long)C, (long long)SP, zs, z);
  else
    hipLaunchKernelGGL((bn_apply_kernel<T, 1>), dim3(ew_grid(total, 1)), dim3(256), 0, (hipStream_t)stream, x, scale,
                       shift, residual, relu, total, (long long)C, (long long)SP, zs, z);
  return gca_launch_status();
}

template <typename T>
static int bn_train_fwd_t(const float* stat_sum, const float* stat_sq, int64_t P, int64_t C, double count,
                     const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float* save_mean, float* save_invstd, float* scale, float* shift,
                     const T* x, const T* residual, int relu, int64_t N, int64_t SP, T* z,
                     int64_t z_batch_stride, void* stream) {
  if (!x || !z || N <= 0 || SP <= 0 || (double)N * (double)SP != count) return GCA_EINVAL;
  if (!stat_sum || !stat_sq || P <= 0 || C <= 0 || !scale || !shift) return GCA_EINVAL;
  if (z_batch_stride != 0 && z_batch_stride < C * SP) return GCA_EINVAL;
  if (N * SP <= BN_SMALL_ELEMS && N * C * SP < (1LL << 31)) {
    const long long zs = z_batch_stride ? z_batch_stride : C * SP;
    const bool v4 = (SP % 4 == 0) && (zs % 4 == 0) && (((uintptr_t)x | (uintptr_t)z | (uintptr_t)residual) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (v4)
      hipLaunchKernelGGL((bn_fwd_small_kernel<T, 4>), dim3((unsigned)C), dim3(256), 0, st, stat_sum, stat_sq, (long long)P, count,
                         gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift,
                         (long long*)num_batches_tracked, x, residual, relu, (int)N, (int)C, (int)SP, zs, z);
    else
      hipLaunchKernelGGL((bn_fwd_small_kernel<T, 1>), dim3((unsigned)C), dim3(256), 0, st, stat_sum, stat_sq, (long long)P, count,
                         gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift,
                         (long long*)num_batches_tracked, x, residual, relu, (int)N, (int)C, (int)SP, zs, z);
    return gca_launch_status();
  }
  int rc = gca_bn_finalize(stat_sum, stat_sq, P, C, count, gamma, beta, eps, momentum, running_mean, running_var,
                           num_batches_tracked, save_mean, save_invstd, scale, shift, stream);
  if (rc) return rc;
  return bn_apply_t<T>(x, scale, shift, residual, relu, N, C, SP, z, z_batch_stride, stream);
}

template <typename T>
static int bn_bwd_t(const T* dz_in, const T* z, const T* x, const float* gamma,
               const float* save_mean, const float* save_invstd, int relu,
               int64_t N, int64_t C, int64_t SP, T* dx, float* dgamma, float* dbeta,
               T* dres, int dres_accumulate, int64_t z_batch_stride, const float* scale, const float* shift,
               void* ws, void* stream) {
  if (!dz_in || !x || !save_mean || !save_invstd || !dx || !ws || N <= 0 || C <= 0 || SP <= 0) return GCA_EINVAL;
  if (relu < 0 || relu > 2 || (relu == 1 && !z) || (relu == 2 && (!scale || !shift))) return GCA_EINVAL;
  if (relu != 1) z = dz_in;                 // never dereferenced; keeps the alignment test below meaningful
  if (z_batch_stride != 0 && z_batch_stride < C * SP) return GCA_EINVAL;
  const long long zs = z_batch_stride ? z_batch_stride : C * SP;
  hipStream_t st = (hipStream_t)stream;
  if (N * SP <= BN_SMALL_ELEMS && N * C * SP < (1LL << 31)) {
    const bool s4 = (SP % 4 == 0) && (zs % 4 == 0) &&
                    (((uintptr_t)dz_in | (uintptr_t)z | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres) % 16 == 0);
    if (s4)
      hipLaunchKernelGGL((bn_bwd_small_kernel<T, 4>), dim3((unsigned)C), dim3(256), 0, st, dz_in, z, x, gamma, save_mean, save_invstd,
                         relu, (int)N, (int)C, (int)SP, zs, dx, dgamma, dbeta, dres, dres_accumulate, scale, shift);
    else
      hipLaunchKernelGGL((bn_bwd_small_kernel<T, 1>), dim3((unsigned)C), dim3(256), 0, st, dz_in, z, x, gamma, save_mean, save_invstd,
                         relu, (int)N, (int)C, (int)SP, zs, dx, dgamma, dbeta, dres, dres_accumulate, scale, shift);
    return gca_launch_status();
  }
  const int P = (int)stats_parts(N, C, SP);
  float* p0 = reinterpret_cast<float*>(ws);
  float* p1 = p0 + (long long)C * P;
  float* coef = p1 + (long long)C * P;
  if ((SP % 4 == 0) && (zs % 4 == 0) && (((uintptr_t)dz_in | (uintptr_t)z | (uintptr_t)x) % 16 == 0))
    hipLaunchKernelGGL((bn_reduce_kernel<T, 1, 4>), dim3(P, (unsigned)C), dim3(256), 0, st, dz_in, z, x, save_mean,
                       save_invstd, relu, (long long)N, (long long)C, (long long)SP, zs, P, p0, p1, scale, shift);
  else
    hipLaunchKernelGGL((bn_reduce_kernel<T, 1, 1>), dim3(P, (unsigned)C), dim3(256), 0, st, dz_in, z, x, save_mean,
                       save_invstd, relu, (long long)N, (long long)C, (long long)SP, zs, P, p0, p1, scale, shift);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)C), dim3(256), 0, st, p0, p1, P, (double)N * (double)SP,
                     gamma, save_invstd, dgamma, dbeta, coef, (long long)C);
  const long long total = (long long)N * C * SP;
  const bool v4 = (SP % 4 == 0) && (zs % 4 == 0) &&
                  (((uintptr_t)dz_in | (uintptr_t)z | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres) % 16 == 0);
  if (v4)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 4>), dim3(ew_grid(total, 4)), dim3(256), 0, st, dz_in, z, x, save_mean,
                       save_invstd, coef, relu, total, (long long)C, (long long)SP, zs, dx, dres, dres_accumulate, scale, shift);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1>), dim3(ew_grid(total, 1)), dim3(256), 0, st, dz_in, z, x, save_mean,
                       save_invstd, coef, relu, total, (long long)C, (long long)SP, zs, dx, dres, dres_accumulate, scale, shift);
  return gca_launch_status();
}

extern "C" {

int64_t gca_bn_stats_parts(int64_t N, int64_t C, int64_t SP) { return stats_parts(N, C, SP); }


int gca_bn_finalize(const float* stat_sum, const float* stat_sq, int64_t P, int64_t C, double count,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* save_mean, float* save_invstd, float* scale, float* shift, void* stream) {
  if (!stat_sum || !stat_sq || P <= 0 || C <= 0 || count <= 0 || !scale || !shift) return GCA_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, stat_sum, stat_sq,
                     (long long)P, count, gamma, beta, eps, momentum, running_mean, running_var, save_mean,
                     save_invstd, scale, shift, (long long*)num_batches_tracked);
  return gca_launch_status();
}


int gca_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, int64_t C, float* scale, float* shift, void* stream) {
  if (!running_mean || !running_var || !scale || !shift || C <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((unsigned)gca_ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     gamma, beta, running_mean, running_var, eps, (long long)C, scale, shift);
  return gca_launch_status();
}


int64_t gca_bn_bwd_ws_bytes(int64_t N, int64_t C, int64_t SP) {
  if (N <= 0 || C <= 0 || SP <= 0) return GCA_EINVAL;
  return (int64_t)sizeof(float) * (2 * C * stats_parts(N, C, SP) + 3 * C);
}


int gca_bn_stats(const void* x, int64_t N, int64_t C, int64_t SP, float* stat_sum, float* stat_sq,
                 int64_t* parts_out, int act_f16, void* stream) {
  return act_f16 ? bn_stats_t<gca_half>((const gca_half*)x, N, C, SP, stat_sum, stat_sq, parts_out, stream)
                 : bn_stats_t<float>((const float*)x, N, C, SP, stat_sum, stat_sq, parts_out, stream);
}

int gca_bn_train_fwd(const float* stat_sum, const float* stat_sq, int64_t P, int64_t C, double count,
                     const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float* save_mean, float* save_invstd, float* scale, float* shift,
                     const void* x, const void* residual, int relu, int64_t N, int64_t SP, void* z,
                     int64_t z_batch_stride, int act_f16, void* stream) {
  if (act_f16)
    return bn_train_fwd_t<gca_half>(stat_sum, stat_sq, P, C, count, gamma, beta, eps, momentum, running_mean, running_var,
                                    num_batches_tracked, save_mean, save_invstd, scale, shift, (const gca_half*)x,
                                    (const gca_half*)residual, relu, N, SP, (gca_half*)z, z_batch_stride, stream);
  return bn_train_fwd_t<float>(stat_sum, stat_sq, P, C, count, gamma, beta, eps, momentum, running_mean, running_var,
                               num_batches_tracked, save_mean, save_invstd, scale, shift, (const float*)x,
                               (const float*)residual, relu, N, SP, (float*)z, z_batch_stride, stream);
}

int gca_bn_train_fwd_slabs(const float* slabs, int64_t splits, double count, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float* save_mean, float* save_invstd, float* scale, float* shift, float* y, const float* residual,
                           int relu, int64_t N, int64_t C, int64_t SP, float* z, int64_t z_batch_stride, void* stream) {
  if (!slabs || splits < 2 || splits > 1024 || count <= 0 || !scale || !shift || !y || !z || N <= 0 || C <= 0 || SP <= 0)
    return GCA_EINVAL;
  if (N * SP > BN_SMALL_ELEMS || N * C * SP >= (1LL << 31)) return GCA_EINVAL;       // the one-workgroup-per-channel regime only
  if (z_batch_stride != 0 && z_batch_stride < C * SP) return GCA_EINVAL;
  const long long zs = z_batch_stride ? z_batch_stride : C * SP;
  hipLaunchKernelGGL(bn_fwd_small_slab_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, slabs, (int)splits, count,
                     gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift,
                     (long long*)num_batches_tracked, y, residual, relu, (int)N, (int)C, (int)SP, zs, z);
  return gca_launch_status();
}

int gca_bn_apply(const void* x, const float* scale, const float* shift, const void* residual,
                 int relu, int64_t N, int64_t C, int64_t SP, void* z, int64_t z_batch_stride, int act_f16, void* stream) {
  return act_f16 ? bn_apply_t<gca_half>((const gca_half*)x, scale, shift, (const gca_half*)residual, relu, N, C, SP,
                                        (gca_half*)z, z_batch_stride, stream)
                 : bn_apply_t<float>((const float*)x, scale, shift, (const float*)residual, relu, N, C, SP, (float*)z,
                                     z_batch_stride, stream);
}

int gca_bn_bwd(const void* dz_in, const void* z, const void* x, const float* gamma,
               const float* save_mean, const float* save_invstd, int relu,
               int64_t N, int64_t C, int64_t SP, void* dx, float* dgamma, float* dbeta,
               void* dres, int dres_accumulate, int64_t z_batch_stride, const float* scale, const float* shift,
               void* ws, int act_f16, void* stream) {
  if (act_f16)
    return bn_bwd_t<gca_half>((const gca_half*)dz_in, (const gca_half*)z, (const gca_half*)x, gamma, save_mean, save_invstd,
                              relu, N, C, SP, (gca_half*)dx, dgamma, dbeta, (gca_half*)dres, dres_accumulate, z_batch_stride,
                              scale, shift, ws, stream);
  return bn_bwd_t<float>((const float*)dz_in, (const float*)z, (const float*)x, gamma, save_mean, save_invstd, relu, N, C, SP,
                         (float*)dx, dgamma, dbeta, (float*)dres, dres_accumulate, z_batch_stride, scale, shift, ws, stream);
}

}  // extern "C"
