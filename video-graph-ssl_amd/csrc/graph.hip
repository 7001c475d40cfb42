// Temporal-graph block kernels (lib/ops/module_wrappers/temporal_graph.py) for gfx950.
// Clip graphs are tiny and dense (T <= 32 frame nodes), so there is no CSR anywhere: the
// adjacency is a dense (B,T,T) tile kept in LDS and the message passing
//     out[b,c,i,:] = sum_j adj[b,i,j] * s[b,c,j,:] + s[b,c,i,:]
// is a dense neighbourhood GEMV per (b,c,hw) column: every lane owns 4 consecutive hw positions,
// loads the T frame values once (float4, coalesced along W) and produces all T outputs -> one
// HBM pass over s, one over out (algorithmic bytes 2*B*C*T*HW*4).
#include "gca_common.h"
#include <math.h>

namespace {

// G[b,i,j] = sum_{c,hw} X[b,c,i,hw] * Y[b,c,j,hw]   (the (T x F)(F x T) similarity / adjacency-gradient product).
// Two passes, both deterministic: gram_tile_kernel<T> -- grid (channel chunks, B); a thread keeps the T frames of
// X and of Y at its hw position in registers and accumulates the whole T x T outer product, so every input
// element is read exactly once (HBM-bound: 2*B*C*T*HW*4 bytes); gram_finish_kernel sums the chunk partials.
template <int T>
__global__ __launch_bounds__(256) void gram_tile_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                        int C, int HW, int cper, float* __restrict__ part) {
  __shared__ float sh[4][T * T];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int c0 = chunk * cper, c1 = min(C, c0 + cper);
  float acc[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) acc[i][j] = 0.f;
  for (int c = c0; c < c1; ++c) {
    const float* xp = X + ((long long)(b * C + c) * T) * HW;
    const float* yp = Y + ((long long)(b * C + c) * T) * HW;
    for (int p = threadIdx.x; p < HW; p += 256) {
      float xv[T], yv[T];
#pragma unroll
      for (int t = 0; t < T; ++t) { xv[t] = xp[t * HW + p]; yv[t] = yp[t * HW + p]; }
#pragma unroll
      for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] += xv[i] * yv[j];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const float v = gca_wave_sum(acc[i][j]);
      if (lane == 0) sh[wave][i * T + j] = v;
    }
  __syncthreads();
  if (threadIdx.x < T * T)
    part[((long long)b * gridDim.x + chunk) * (T * T) + threadIdx.x] =
        sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}
// One block per clip: its 256 threads are TT outputs x (256 / TT) phases; a phase sums every (256/TT)-th chunk partial with
// eight loads in flight, the phases are combined through LDS in a fixed order (deterministic).  (One thread per output
// walking all chunks -- up to 256 dependent loads in a single block -- took 31 us per launch on the 4-clip SimSiam step.)
__global__ __launch_bounds__(256) void gram_finish_kernel(const float* __restrict__ part, int nchunk, int TT,
                                                          float* __restrict__ G) {
  __shared__ float sh[256];
  const int b = blockIdx.x, ij = threadIdx.x % TT, ph = threadIdx.x / TT, nph = 256 / TT;
  const float* p = part + (long long)b * nchunk * TT + ij;
  float s = 0.f;
  int k = ph;
  for (; k + 7 * nph < nchunk; k += 8 * nph) {
    float l[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) l[u] = p[(long long)(k + u * nph) * TT];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += l[u];
  }
  for (; k < nchunk; k += nph) s += p[(long long)k * TT];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (ph == 0) {
    float t = 0.f;
    for (int q = 0; q < nph; ++q) t += sh[q * TT + ij];
    G[b * TT + ij] = t;
  }
}

// Fallback for node counts other than 2 / 4 / 8: one workgroup per (b,i,j).
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                   int C, int T, int HW, float* __restrict__ G) {
  __shared__ float sh[4];
  const int j = blockIdx.x % T, i = (blockIdx.x / T) % T, b = blockIdx.x / (T * T);
  const float* xp = X + ((long long)b * C * T + i) * HW;
  const float* yp = Y + ((long long)b * C * T + j) * HW;
  const long long cs = (long long)T * HW;
  float s = 0.f;
  for (int c = 0; c < C; ++c)
    for (int p = threadIdx.x; p < HW; p += 256) s += xp[c * cs + p] * yp[c * cs + p];
  s = gca_block_sum256(s, sh);
  if (threadIdx.x == 0) G[blockIdx.x] = s;
}

inline int gram_chunks(long long B, int C) {           // channel chunks so that the grid is ~1024 workgroups
  long long n = gca_ceil_div(1024, B);
  if (n > C) n = C;
  if (n < 1) n = 1;
  const int cper = (int)gca_ceil_div(C, n);
  return (int)gca_ceil_div(C, cper);
}
inline bool gram_tiled(int T) { return T == 2 || T == 4 || T == 8; }
inline long long gram_ws_floats(long long B, int C, int T) { return gram_tiled(T) ? B * gram_chunks(B, C) * T * T : 0; }

int launch_gram(const float* X, const float* Y, long long B, int C, int T, int HW, float* G, float* ws, hipStream_t st) {
  if (!gram_tiled(T) || !ws) {
    hipLaunchKernelGGL(gram_kernel, dim3((unsigned)(B * T * T)), dim3(256), 0, st, X, Y, C, T, HW, G);
    return gca_launch_status();
  }
  const int nchunk = gram_chunks(B, C), cper = (int)gca_ceil_div(C, nchunk);
  const dim3 grid((unsigned)nchunk, (unsigned)B);
  if (T == 2) hipLaunchKernelGGL((gram_tile_kernel<2>), grid, dim3(256), 0, st, X, Y, C, HW, cper, ws);
  else if (T == 4) hipLaunchKernelGGL((gram_tile_kernel<4>), grid, dim3(256), 0, st, X, Y, C, HW, cper, ws);
  else hipLaunchKernelGGL((gram_tile_kernel<8>), grid, dim3(256), 0, st, X, Y, C, HW, cper, ws);
  const int TT = T * T;
  hipLaunchKernelGGL(gram_finish_kernel, dim3((unsigned)B), dim3(256), 0, st, ws, nchunk, TT, G);      // TT in {4, 16, 64} divides 256
  return gca_launch_status();
}

__device__ __forceinline__ float theta_hop(int h, float alpha) {
  const float e = expf(-(float)h);
  return e / (1.f + e * e) + alpha;
}
__device__ __forceinline__ float clampp(float p) {
  const float eps = 1.1920928955078125e-07f;        // torch.finfo(float32).eps
  return fminf(fmaxf(p, eps), 1.f - eps);
}

// one thread per (b,i) row: softmax over j, hop weighting, relaxed-Bernoulli reparameterised sample
__global__ void adj_fwd_kernel(const float* S, int BT, int T, int max_hop, float alpha, float temp,
                               const float* __restrict__ u, float* sim, float* __restrict__ pre,
                               float* __restrict__ adj) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= BT) return;
  const int i = r % T;
  const float* s = S + (long long)r * T;
  float m = -INFINITY;
  for (int j = 0; j < T; ++j) m = fmaxf(m, s[j]);
  float den = 0.f;
  for (int j = 0; j < T; ++j) den += expf(s[j] - m);
  for (int j = 0; j < T; ++j) {
    const float sm = expf(s[j] - m) / den;
    const int h = abs(i - j);
    const float p = (h <= max_hop) ? sm * theta_hop(h, alpha) : 0.f;
    if (sim) sim[(long long)r * T + j] = sm;
    if (pre) pre[(long long)r * T + j] = p;
    if (adj) {
      const float pc = clampp(p), uc = clampp(u[(long long)r * T + j]);
      const float lg = (logf(uc) - log1pf(-uc) + logf(pc) - log1pf(-pc)) / temp;
      adj[(long long)r * T + j] = 1.f / (1.f + expf(-lg));
    }
  }
}

// dadj -> dS (gradient wrt the pre-softmax similarity), one thread per (b,i) row
__global__ void adj_bwd_kernel(const float* dadj, const float* __restrict__ sim,
                               const float* __restrict__ pre, const float* __restrict__ adj, int BT, int T,
                               int max_hop, float alpha, float temp, float* dS) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= BT) return;
  const int i = r % T;
  const float eps = 1.1920928955078125e-07f;
  float dot = 0.f;
  // dsim_j = dadj_j * a(1-a)/temp / (p(1-p)) * theta(h)   inside the clamp range and the hop band
  for (int pass = 0; pass < 2; ++pass) {
    for (int j = 0; j < T; ++j) {
      const long long e = (long long)r * T + j;
      const int h = abs(i - j);
      float ds = 0.f;
      const float p = pre[e];
      if (h <= max_hop && p > eps && p < 1.f - eps) {
        const float a = adj[e];
        ds = dadj[e] * a * (1.f - a) / temp / (p * (1.f - p)) * theta_hop(h, alpha);
      }
      if (pass == 0) dot += ds * sim[e];
      else dS[e] = sim[e] * (ds - dot);
    }
  }
}

// out[b,c,i,:] = sum_j M[b,i,j] * x[b,c,j,:] (+ x[b,c,i,:]);  transpose -> uses M[b,j,i]
template <int TT>
__global__ __launch_bounds__(256) void tmix_kernel(const float* __restrict__ M, const float* __restrict__ x,
                                                   int C, int HW4, int transpose, int skip, float* __restrict__ out) {
  __shared__ float Ms[TT * TT];
  const int b = blockIdx.y;
  for (int e = threadIdx.x; e < TT * TT; e += 256) {
    const int i = e / TT, j = e % TT;
    Ms[e] = transpose ? M[((long long)b * TT + j) * TT + i] : M[((long long)b * TT + i) * TT + j];
  }
  __syncthreads();
  const long long cols = (long long)C * HW4;
  for (long long col = (long long)blockIdx.x * 256 + threadIdx.x; col < cols; col += (long long)gridDim.x * 256) {
    const int c = (int)(col / HW4), p4 = (int)(col - (long long)c * HW4);
    const float4* xp = reinterpret_cast<const float4*>(x) + ((long long)(b * C + c) * TT) * HW4 + p4;
    float4* op = reinterpret_cast<float4*>(out) + ((long long)(b * C + c) * TT) * HW4 + p4;
    float4 v[TT];
#pragma unroll
    for (int j = 0; j < TT; ++j) v[j] = xp[(long long)j * HW4];
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      float4 o = skip ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < TT; ++j) {
        const float m = Ms[i * TT + j];
        o.x += m * v[j].x; o.y += m * v[j].y; o.z += m * v[j].z; o.w += m * v[j].w;
      }
      op[(long long)i * HW4] = o;
    }
  }
}

// generic (any T, any HW): one thread per output element
__global__ __launch_bounds__(256) void tmix_generic_kernel(const float* __restrict__ M, const float* __restrict__ x,
                                                           long long total, int C, int T, int HW, int transpose,
                                                           int skip, float* __restrict__ out) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int p = (int)(e % HW);
    const int i = (int)((e / HW) % T);
    const long long bc = e / ((long long)HW * T);
    const int b = (int)(bc / C);
    const float* xp = x + bc * T * HW + p;
    float o = skip ? xp[(long long)i * HW] : 0.f;
    for (int j = 0; j < T; ++j) {
      const float m = transpose ? M[((long long)b * T + j) * T + i] : M[((long long)b * T + i) * T + j];
      o += m * xp[(long long)j * HW];
    }
    out[e] = o;
  }
}

int launch_tmix(const float* M, const float* x, long long B, int C, int T, int HW, int transpose, int skip, float* out,
                hipStream_t st) {
  const bool vec = (HW % 4 == 0) && (((uintptr_t)x | (uintptr_t)out) % 16 == 0);
  if (vec && (T == 2 || T == 4 || T == 8 || T == 16)) {
    const int HW4 = HW / 4;
    long long bx = gca_ceil_div((long long)C * HW4, 256);
    if (bx > 1024) bx = 1024;
    dim3 grid((unsigned)bx, (unsigned)B);
    switch (T) {
      case 2: hipLaunchKernelGGL((tmix_kernel<2>), grid, dim3(256), 0, st, M, x, C, HW4, transpose, skip, out); break;
      case 4: hipLaunchKernelGGL((tmix_kernel<4>), grid, dim3(256), 0, st, M, x, C, HW4, transpose, skip, out); break;
      case 8: hipLaunchKernelGGL((tmix_kernel<8>), grid, dim3(256), 0, st, M, x, C, HW4, transpose, skip, out); break;
      default: hipLaunchKernelGGL((tmix_kernel<16>), grid, dim3(256), 0, st, M, x, C, HW4, transpose, skip, out); break;
    }
  } else {
    const long long total = B * C * T * HW;
    long long bx = gca_ceil_div(total, 256);
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(tmix_generic_kernel, dim3((unsigned)bx), dim3(256), 0, st, M, x, total, C, T, HW, transpose, skip, out);
  }
  return gca_launch_status();
}

}  // namespace

extern "C" {

int gca_graph_adj_fwd(const float* gq, const float* gk, int64_t B, int64_t Ci, int64_t T, int64_t HW,
                      int max_hop, float alpha, float temperature, const float* u,
                      float* sim, float* adj_pre, float* adj, void* ws, void* stream) {
  if (!gq || !gk || B <= 0 || Ci <= 0 || T <= 0 || HW <= 0 || temperature <= 0.f) return GCA_EINVAL;
  if (adj && !u) return GCA_EINVAL;
  if (!sim) return GCA_EINVAL;      // sim doubles as the scratch for the raw similarities
  hipStream_t st = (hipStream_t)stream;
  int rc0 = launch_gram(gq, gk, B, (int)Ci, (int)T, (int)HW, sim, reinterpret_cast<float*>(ws), st);
  if (rc0) return rc0;
  const int BT = (int)(B * T);
  hipLaunchKernelGGL(adj_fwd_kernel, dim3((unsigned)gca_ceil_div(BT, 64)), dim3(64), 0, st, sim, BT, (int)T, max_hop,
                     alpha, temperature, u, sim, adj_pre, adj);
  return gca_launch_status();
}

int gca_graph_adj_bwd(const float* dadj, const float* gq, const float* gk, const float* sim,
                      const float* adj_pre, const float* adj, int64_t B, int64_t Ci, int64_t T, int64_t HW,
                      int max_hop, float alpha, float temperature, float* dgq, float* dgk, void* stream) {
  // dadj is an in/out buffer: it is overwritten with dS (gradient wrt the pre-softmax similarity).
  if (!dadj || !gq || !gk || !sim || !adj_pre || !adj || !dgq || !dgk || B <= 0 || Ci <= 0 || T <= 0 || HW <= 0)
    return GCA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* dS = const_cast<float*>(dadj);
  const int BT = (int)(B * T);
  hipLaunchKernelGGL(adj_bwd_kernel, dim3((unsigned)gca_ceil_div(BT, 64)), dim3(64), 0, st, dadj, sim, adj_pre, adj, BT,
                     (int)T, max_hop, alpha, temperature, dS);
  int rc = gca_launch_status();
  if (rc) return rc;
  rc = launch_tmix(dS, gk, B, (int)Ci, (int)T, (int)HW, 0, 0, dgq, st);     // dgq_i = sum_j dS_ij gk_j
  if (rc) return rc;
  return launch_tmix(dS, gq, B, (int)Ci, (int)T, (int)HW, 1, 0, dgk, st);   // dgk_j = sum_i dS_ij gq_i
}

int gca_graph_gcn_fwd(const float* adj, const float* s, int64_t B, int64_t C, int64_t T, int64_t HW,
                      float* out, void* stream) {
  if (!adj || !s || !out || B <= 0 || C <= 0 || T <= 0 || HW <= 0) return GCA_EINVAL;
  return launch_tmix(adj, s, B, (int)C, (int)T, (int)HW, 0, 1, out, (hipStream_t)stream);
}

int64_t gca_graph_gram_ws_bytes(int64_t B, int64_t C, int64_t T, int64_t HW) {
  (void)HW;
  if (B <= 0 || C <= 0 || T <= 0) return GCA_EINVAL;
  return 256 + (int64_t)sizeof(float) * gram_ws_floats(B, (int)C, (int)T);
}
int64_t gca_graph_gcn_bwd_ws_bytes(int64_t B, int64_t C, int64_t T, int64_t HW) { return gca_graph_gram_ws_bytes(B, C, T, HW); }

int gca_graph_gcn_bwd(const float* adj, const float* s, const float* dout, int64_t B, int64_t C, int64_t T,
                      int64_t HW, float* ds, float* dadj, void* ws, void* stream) {
  if (!adj || !s || !dout || !ds || B <= 0 || C <= 0 || T <= 0 || HW <= 0) return GCA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_tmix(adj, dout, B, (int)C, (int)T, (int)HW, 1, 1, ds, st);
  if (rc || !dadj) return rc;
  return launch_gram(dout, s, B, (int)C, (int)T, (int)HW, dadj, reinterpret_cast<float*>(ws), st);
}

}  // extern "C"
