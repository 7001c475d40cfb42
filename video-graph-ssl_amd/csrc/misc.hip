// Head pieces (ReLU, row L2-normalise, negative cosine) and the multi-tensor parameter updates
// (EMA key-encoder update, SGD) over flat parameter arenas.  All HBM-bound float4 streaming.
// Reference call sites: see include/gca_hip.h.
#include <cstdint>
#include "gca_common.h"

namespace {

inline unsigned ew_blocks(long long n4) {
  long long b = gca_ceil_div(n4, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

__global__ void relu_fwd_kernel(const float* __restrict__ x, long long n, float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long n,
                                float* __restrict__ dx) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// one wave per row: y = x / max(||x||, eps)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, long long rows, int dim, float eps,
                                                         float* __restrict__ y, float* __restrict__ inv) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + r * dim;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) s += xr[i] * xr[i];
  s = gca_wave_sum(s);
  const float iv = 1.f / fmaxf(sqrtf(s), eps);
  for (int i = lane; i < dim; i += 64) y[r * dim + i] = xr[i] * iv;
  if (lane == 0 && inv) inv[r] = iv;
}
// dx = inv * (dy - y * <dy, y>)    (exact where ||x|| > eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv, long long rows, int dim,
                                                         float* __restrict__ dx) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) s += dy[r * dim + i] * y[r * dim + i];
  s = gca_wave_sum(s);
  const float iv = inv[r];
  for (int i = lane; i < dim; i += 64) dx[r * dim + i] = iv * (dy[r * dim + i] - y[r * dim + i] * s);
}

// loss (+)= -scale/rows * sum_i cos(p_i, z_i);  dp_i = -scale/rows * ( z_i/(|p||z|) - cos * p_i/|p|^2 )
// F.cosine_similarity clamps each norm with eps = 1e-8.
__global__ __launch_bounds__(256) void negcos_kernel(const float* __restrict__ p, const float* __restrict__ z,
                                                     long long rows, int dim, float scale, float* __restrict__ rowcos,
                                                     float* __restrict__ dp) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float pp = 0.f, zz = 0.f, pz = 0.f;
  for (int i = lane; i < dim; i += 64) {
    const float a = p[r * dim + i], b = z[r * dim + i];
    pp += a * a; zz += b * b; pz += a * b;
  }
  pp = gca_wave_sum(pp); zz = gca_wave_sum(zz); pz = gca_wave_sum(pz);
  const float np = fmaxf(sqrtf(pp), 1e-8f), nz = fmaxf(sqrtf(zz), 1e-8f);
  const float c = pz / (np * nz);
  if (lane == 0) rowcos[r] = c;
  if (dp) {
    const float k = -scale / (float)rows;
    for (int i = lane; i < dim; i += 64)
      dp[r * dim + i] = k * (z[r * dim + i] / (np * nz) - c * p[r * dim + i] / (np * np));
  }
}
__global__ __launch_bounds__(256) void negcos_finish_kernel(const float* __restrict__ rowcos, long long rows, float scale,
                                                            float* __restrict__ loss, int accumulate) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long long i = threadIdx.x; i < rows; i += 256) s += rowcos[i];
  s = gca_block_sum256(s, sh);
  if (threadIdx.x == 0) {
    const float v = -scale * s / (float)rows;
    *loss = accumulate ? *loss + v : v;
  }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ pe, const float* __restrict__ p, long long n4, float m) {
  const float om = 1.f - m;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 a = reinterpret_cast<float4*>(pe)[i];
    const float4 b = reinterpret_cast<const float4*>(p)[i];
    // p_k.mul_(m).add_(p_q, alpha=1-m)  (tools/train_video_contrast_dis.py:180)
    a.x = a.x * m + om * b.x; a.y = a.y * m + om * b.y; a.z = a.z * m + om * b.z; a.w = a.w * m + om * b.w;
    reinterpret_cast<float4*>(pe)[i] = a;
  }
}

// torch.optim.SGD: d = g + wd*p; buf = (first ? d : mom*buf + d); d = nesterov ? d + mom*buf : buf; p -= lr*d
// One 256-element chunk per (lr, wd) pair: chunk c = i4 / 64.
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  long long n4, const float* __restrict__ clr, const float* __restrict__ cwd,
                                                  float lr_scale, float mom, int nesterov, int first,
                                                  const float* __restrict__ gscale) {
  const float gs = gscale ? gscale[1] : 1.f;       // clip coefficient of gca_grad_clip_coef (grads.mul_(coef) folded in)
  if (gscale && gscale[2] != 0.f) return;          // a non-finite gradient was found (gca_grad_unscale_clip): optimizer.step() is skipped
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long c = i >> 6;
    const float lr = clr[c] * lr_scale, wd = cwd[c];
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 gv = reinterpret_cast<const float4*>(g)[i];
    if (gscale) { gv.x *= gs; gv.y *= gs; gv.z *= gs; gv.w *= gs; }
    float4 bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(buf)[i];
    float d;
#define GCA_SGD1(f)                                   \
    d = gv.f + wd * pv.f;                             \
    bv.f = first ? d : mom * bv.f + d;                \
    d = nesterov ? d + mom * bv.f : bv.f;             \
    pv.f = pv.f - lr * d;
    GCA_SGD1(x) GCA_SGD1(y) GCA_SGD1(z) GCA_SGD1(w)
#undef GCA_SGD1
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(buf)[i] = bv;
  }
}

// clip_grad_norm_ (tools/train_video_contrast_dis.py:420-423; torch.nn.utils.clip_grad_norm_, 2-norm): per-block partial
// sums of squares in fp64 over the whole gradient arena (padding elements are zero), then one block folds them in a fixed
// order: out[0] = total_norm, out[1] = min(1, max_norm / (total_norm + 1e-6)).  Deterministic.
constexpr int CLIP_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sqsum_partial_kernel(const float* __restrict__ g, long long n4, double* __restrict__ part) {
  __shared__ double sh[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  s = gca_block_sum256_d(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ part, int nparts, float max_norm,
                                                        float* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = gca_block_sum256_d(s, sh);
  if (threadIdx.x == 0) {
    const float tn = (float)sqrt(s);
    out[0] = tn;
    out[1] = fminf(max_norm / (tn + 1e-6f), 1.0f);
    out[2] = 0.f;                                   // (torch's clip_grad_norm_ does not skip on a non-finite norm: neither does this)
    out[3] = 0.f;
  }
}

// Dynamic loss scaling (the reference's apex amp, tools/train_video_contrast_dis.py:134-141,413-418: scaled loss, un-scaled
// gradients, optimizer.step() skipped and the scale halved when a gradient is inf / nan, scale doubled after
// `interval` clean steps), decided on the device from the same fp64 sum of squares the clip uses: a non-finite element
// makes the sum non-finite.  state = {scale S, clean steps in a row, skipped steps so far, steps seen}.
__global__ __launch_bounds__(256) void unscale_clip_kernel(const double* __restrict__ part, int nparts, float max_norm,
                                                           float* __restrict__ state, float growth, float backoff,
                                                           int interval, float max_scale, float* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = gca_block_sum256_d(s, sh);
  if (threadIdx.x == 0) {
    float S = state[0];
    const bool finite = s == s && s < 1.7e308 && s >= 0.0;
    if (finite) {
      const float tn = (float)(sqrt(s) / (double)S);            // norm of the UN-scaled gradient
      const float coef = max_norm > 0.f ? fminf(max_norm / (tn + 1e-6f), 1.0f) : 1.0f;
      out[0] = tn; out[1] = coef / S; out[2] = 0.f; out[3] = S;
      const float clean = state[1] + 1.f;
      if (interval > 0 && clean >= (float)interval) { S = fminf(S * growth, max_scale); state[1] = 0.f; }
      else state[1] = clean;
    } else {
      out[0] = __builtin_inff(); out[1] = 0.f; out[2] = 1.f; out[3] = S;
      S = fmaxf(S * backoff, 1.f);
      state[1] = 0.f;
      state[2] += 1.f;
    }
    state[0] = S;
    state[3] += 1.f;
  }
}
__global__ void scale_dev_kernel(float* __restrict__ y, long long n, const float* __restrict__ a_dev, float a_host) {
  const float a = (a_dev ? *a_dev : 1.f) * a_host;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] *= a;
}

__global__ void fill_kernel(float* __restrict__ p, long long n, float v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long long n, float a) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] += a * x[i];
}
// fp16-storage activations: gradient accumulation y += a*x in fp32, one rounding; and the fp32 -> fp16 cast of the input clips
__global__ void axpy_f16_kernel(gca_half* __restrict__ y, const gca_half* __restrict__ x, long long n, float a) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = (gca_half)((float)y[i] + a * (float)x[i]);
}
// rows of re elements (re % 4 == 0), source rows src_stride apart (a channel slice of a wider clip batch), dense output
__global__ void cast_f16_kernel(const float* __restrict__ x, long long re4, long long src_stride, gca_half* __restrict__ y) {
  const float* xr = x + (long long)blockIdx.y * src_stride;
  gca_half* yr = y + (long long)blockIdx.y * re4 * 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < re4; i += (long long)gridDim.x * blockDim.x)
    gca_act<gca_half>::st4(yr + 4 * i, *reinterpret_cast<const float4*>(xr + 4 * i));
}
__global__ void scale_kernel(float* __restrict__ y, long long n, float a) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] *= a;
}
// dst[r, :] = src[idx[r] * src_stride : +re]  -- one grid row per gathered row, float4 when alignment allows.
template <int V>
__global__ void gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx, long long re,
                                   long long src_stride, float* __restrict__ dst) {
  const long long r = blockIdx.y;
  const float* s = src + idx[r] * src_stride;
  float* d = dst + r * re;
  const long long step = (long long)gridDim.x * blockDim.x;
  if (V == 4) {
    const long long n4 = re >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step)
      reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
  } else {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < re; i += step) d[i] = s[i];
  }
}

}  // namespace

extern "C" {

int gca_relu_fwd(const float* x, int64_t n, float* y, void* stream) {
  if (!x || !y || n <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(relu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, y);
  return gca_launch_status();
}
int gca_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, void* stream) {
  if (!dy || !y || !dx || n <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, (long long)n, dx);
  return gca_launch_status();
}
int gca_l2norm_fwd(const float* x, int64_t rows, int64_t dim, float eps, float* y, float* inv_norm, void* stream) {
  if (!x || !y || rows <= 0 || dim <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)gca_ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)rows, (int)dim, eps, y, inv_norm);
  return gca_launch_status();
}
int gca_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, int64_t rows, int64_t dim,
                   float* dx, void* stream) {
  if (!dy || !y || !inv_norm || !dx || rows <= 0 || dim <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)gca_ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy,
                     y, inv_norm, (long long)rows, (int)dim, dx);
  return gca_launch_status();
}
int gca_negcos_fwd_bwd(const float* p, const float* z, int64_t rows, int64_t dim, float scale,
                       float* loss, int accumulate, float* dp, void* stream) {
  // `loss` must have room for 1 + rows floats: loss[0] = value, loss[1..rows] = per-row cosines
  // (no hidden allocation across the ABI).
  if (!p || !z || !loss || !dp || rows <= 0 || dim <= 0) return GCA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(negcos_kernel, dim3((unsigned)gca_ceil_div(rows, 4)), dim3(256), 0, st, p, z, (long long)rows,
                     (int)dim, scale, loss + 1, dp);
  hipLaunchKernelGGL(negcos_finish_kernel, dim3(1), dim3(256), 0, st, loss + 1, (long long)rows, scale, loss,
                     accumulate ? 1 : 0);
  return gca_launch_status();
}

int gca_ema_update(float* p_ema, const float* p, int64_t n, float m, void* stream) {
  if (!p_ema || !p || n <= 0 || (n & 3)) return GCA_EINVAL;
  hipLaunchKernelGGL(ema_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, p_ema, p, (long long)(n / 4), m);
  return gca_launch_status();
}
int gca_sgd_step(float* p, const float* grad, float* mom_buf, int64_t n, const float* chunk_lr,
                 const float* chunk_wd, float lr_scale, float momentum, int nesterov, int first_step,
                 const float* grad_clip, void* stream) {
  if (!p || !grad || !mom_buf || !chunk_lr || !chunk_wd || n <= 0 || (n & 255)) return GCA_EINVAL;
  hipLaunchKernelGGL(sgd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, p, grad, mom_buf,
                     (long long)(n / 4), chunk_lr, chunk_wd, lr_scale, momentum, nesterov, first_step, grad_clip);
  return gca_launch_status();
}
int64_t gca_grad_clip_ws_bytes(void) { return (int64_t)CLIP_BLOCKS * (int64_t)sizeof(double); }
int gca_grad_clip_coef(const float* grad, int64_t n, float max_norm, float* out4, void* ws, void* stream) {
  float* out2 = out4;
  if (!grad || !out2 || !ws || n <= 0 || (n & 3) || !(max_norm > 0.f)) return GCA_EINVAL;
  long long blocks = gca_ceil_div(n / 4, 256 * 8);
  if (blocks > CLIP_BLOCKS) blocks = CLIP_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, (long long)(n / 4),
                     reinterpret_cast<double*>(ws));
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const double*>(ws),
                     (int)blocks, max_norm, out2);
  return gca_launch_status();
}
int gca_grad_unscale_clip(const float* grad, int64_t n, float max_norm, float* scale_state4, float growth, float backoff,
                          int growth_interval, float max_scale, float* out4, void* ws, void* stream) {
  if (!grad || !out4 || !scale_state4 || !ws || n <= 0 || (n & 3) || !(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f) ||
      growth_interval < 0 || !(max_scale >= 1.f))
    return GCA_EINVAL;
  long long blocks = gca_ceil_div(n / 4, 256 * 8);
  if (blocks > CLIP_BLOCKS) blocks = CLIP_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, (long long)(n / 4),
                     reinterpret_cast<double*>(ws));
  hipLaunchKernelGGL(unscale_clip_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const double*>(ws),
                     (int)blocks, max_norm, scale_state4, growth, backoff, growth_interval, max_scale, out4);
  return gca_launch_status();
}
int gca_scale_dev(float* y, int64_t n, const float* a_dev, float a_host, void* stream) {
  if (!y || n <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, y, (long long)n, a_dev, a_host);
  return gca_launch_status();
}
int gca_fill(float* p, int64_t n, float v, void* stream) {
  if (!p || n <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, (long long)n, v);
  return gca_launch_status();
}
int gca_axpy(float* y, const float* x, int64_t n, float a, void* stream) {
  if (!y || !x || n <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, y, x, (long long)n, a);
  return gca_launch_status();
}
int gca_axpy_f16(void* y, const void* x, int64_t n, float a, void* stream) {
  if (!y || !x || n <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(axpy_f16_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (gca_half*)y, (const gca_half*)x,
                     (long long)n, a);
  return gca_launch_status();
}
int gca_cast_f16(const float* x, int64_t rows, int64_t row_elems, int64_t src_row_stride, void* y, void* stream) {
  if (!y || !x || rows <= 0 || rows > 65535 || row_elems <= 0 || row_elems % 4 || src_row_stride < row_elems ||
      src_row_stride % 4 || ((uintptr_t)x % 16) || ((uintptr_t)y % 8))
    return GCA_EINVAL;
  const long long re4 = row_elems / 4;
  const unsigned bx = (unsigned)(re4 / 256 + 1 < 4096 ? re4 / 256 + 1 : 4096);
  hipLaunchKernelGGL(cast_f16_kernel, dim3(bx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, re4,
                     (long long)src_row_stride, (gca_half*)y);
  return gca_launch_status();
}
int gca_scale(float* y, int64_t n, float a, void* stream) {
  if (!y || n <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, y, (long long)n, a);
  return gca_launch_status();
}
int gca_gather_rows(const float* src, const int64_t* idx, int64_t rows, int64_t row_elems, int64_t src_row_stride,
                    float* dst, void* stream) {
  if (!src || !idx || !dst || rows <= 0 || rows > 65535 || row_elems <= 0 || src_row_stride < row_elems) return GCA_EINVAL;
  const bool v4 = row_elems % 4 == 0 && src_row_stride % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0;
  const long long per = v4 ? row_elems / 4 : row_elems;
  long long bx = gca_ceil_div(per, 256 * 4);
  if (bx < 1) bx = 1;
  if (bx > 1024) bx = 1024;
  const dim3 grid((unsigned)bx, (unsigned)rows);
  if (v4) hipLaunchKernelGGL(gather_rows_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, src, (const long long*)idx,
                             (long long)row_elems, (long long)src_row_stride, dst);
  else hipLaunchKernelGGL(gather_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, src, (const long long*)idx,
                          (long long)row_elems, (long long)src_row_stride, dst);
  return gca_launch_status();
}

}  // extern "C"
