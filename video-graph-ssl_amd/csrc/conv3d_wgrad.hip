// Weight gradient of the 3D convolution on the fp32 matrix cores, plus the bias gradient.
//
//   dW[ko, c, tap] = sum_{n, o} dY[n, ko, o] * X[n, c, o*s - p + tap]        M = Ko, N = C*taps, K = n*o
//
// The reduction runs over the (huge) spatial axis, which is also the contiguous one in NCDHW, so BOTH
// operands are loaded with lanes along k (coalesced) and land in LDS as [row][k] with a 144-byte pitch:
// the scalar stores are conflict-free (lanes = consecutive k) and every lane then fetches four k-steps of
// its operand row with one conflict-free ds_read_b128 (same k-permutation trick as conv3d.hip).  dY rows
// are read as float4 when the output plane size allows it.  X is gathered through a buffer resource (an
// out-of-window tap is an out-of-range offset and reads 0 in hardware).  K is split across workgroups into
// fp32 partial slabs that are summed in a fixed order (deterministic).
#include "conv_common.h"

using namespace gca_conv;

namespace {

constexpr int LDW = WBK + 4;       // LDS row pitch in floats (144 B)

typedef gca_magic Magic;
inline Magic make_magic(unsigned d) { return gca_make_magic(d); }
__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned /*d*/, Magic g) { return gca_fdiv(n, g); }

struct WgradParams {
  int C, D, H, W, K, OD, OH, OW;
  int sd, sh, sw, pd, ph, pw;
  int Kred;                 // C*taps  (GEMM N)
  int tilesM, tilesN, splits;
  int kt_per_split, kt_total;
  int chk;
  unsigned Ktot;            // NB*OD*OH*OW (GEMM K) -- < 2^30
  unsigned x_nstride;       // elements between clips of x
  unsigned x_bytes, dy_bytes;
  Magic m_osp, m_ohw, m_ow;
};

template <int BM, int BNW, bool AVEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const int2* __restrict__ table,
    float* __restrict__ slab, WgradParams p) {
  constexpr int WM = 2, WN = 2;
  constexpr int TM = BM / WM / 32, TN = BNW / WN / 32;
  constexpr int A_PER = AVEC ? BM / 32 : BM / 8;     // float4 (4 k) or scalar loads per thread for dY
  constexpr int B_PER = BNW / 8;                     // scalar gathers per thread for X

  __shared__ __attribute__((aligned(16))) float As[2][BM][LDW];
  __shared__ __attribute__((aligned(16))) float Bs[2][BNW][LDW];
  __shared__ int2 Ts[BNW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = blockIdx.x;
  const int split = bid % p.splits; bid /= p.splits;
  const int tileM = bid % p.tilesM, tileN = bid / p.tilesM;

  if (tid < BNW) Ts[tid] = table[tileN * BNW + tid];
  __syncthreads();

  const int kl = tid & 31, g = tid >> 5;             // B (and scalar A): k lane, row group
  const int kq = tid & 7, ga = tid >> 3;             // vector A: k quad, row group
  const unsigned OSP = (unsigned)(p.OD * p.OH * p.OW), OHW = (unsigned)(p.OH * p.OW);
  const int HW = p.H * p.W;
  const bool chkD = p.chk & 1, chkH = p.chk & 2, chkW = p.chk & 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, p.dy_bytes, 0x00020000);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kt0 = split * p.kt_per_split;
  int kt1 = kt0 + p.kt_per_split; if (kt1 > p.kt_total) kt1 = p.kt_total;
  const int lh = lane >> 5, ll = lane & 31;

  float areg[AVEC ? 1 : A_PER];
  float4 avec[AVEC ? A_PER : 1];
  float breg[B_PER];

  auto load_tiles = [&](int kt) __attribute__((always_inline)) {
    // ---- A: dY[img, m, o]
    if (AVEC) {
      const unsigned kp = (unsigned)kt * WBK + kq * 4;          // 4 consecutive positions, same image (OSP % 4 == 0)
      const unsigned kc = kp < p.Ktot ? kp : 0u;
      const unsigned img = fdiv(kc, OSP, p.m_osp);
      const unsigned o = kc - img * OSP;
      const unsigned base = (img * (unsigned)p.K * OSP + o) * 4u;
      const unsigned kinv = kp < p.Ktot ? 0u : 0xffffffffu;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int m = tileM * BM + ga + 32 * i;
        const unsigned voff = (base + (unsigned)m * OSP * 4u) | kinv | (m < p.K ? 0u : 0xffffffffu);
        // whole-vector bit_cast: element-wise __builtin_bit_cast miscompiles to a replicated dword load (ROCm 7.2)
        const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, (int)voff, 0, 0));
        avec[i] = make_float4(f.x, f.y, f.z, f.w);
      }
    }
    // ---- per-lane spatial position of the B column (and of scalar A)
    const unsigned kp = (unsigned)kt * WBK + kl;
    const bool kv = kp < p.Ktot;
    const unsigned kc = kv ? kp : 0u;
    const unsigned img = fdiv(kc, OSP, p.m_osp);
    const unsigned o = kc - img * OSP;
    if (!AVEC) {
      const unsigned base = (img * (unsigned)p.K * OSP + o) * 4u;
      const unsigned kinv = kv ? 0u : 0xffffffffu;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int m = tileM * BM + g + 8 * i;
        const unsigned voff = (base + (unsigned)m * OSP * 4u) | kinv | (m < p.K ? 0u : 0xffffffffu);
        areg[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, (int)voff, 0, 0));
      }
    }
    // ---- B: X window element for (c, tap) = table row n'
    const unsigned od = fdiv(o, OHW, p.m_ohw), r = o - od * OHW;
    const unsigned oh = fdiv(r, (unsigned)p.OW, p.m_ow), ow = r - oh * (unsigned)p.OW;
    const int id0 = (int)od * p.sd - p.pd, ih0 = (int)oh * p.sh - p.ph, iw0 = (int)ow * p.sw - p.pw;
    const unsigned bbase = (img * p.x_nstride + (unsigned)(id0 * HW + ih0 * p.W + iw0)) * 4u;
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int2 e = Ts[g + 8 * j];
      int off, dd, dh, dw, rvalid;
      decode_row(e, off, dd, dh, dw, rvalid);
      bool ok = kv & (rvalid != 0);
      if (chkD) ok = ok & ((unsigned)(id0 + dd) < (unsigned)p.D);
      if (chkH) ok = ok & ((unsigned)(ih0 + dh) < (unsigned)p.H);
      if (chkW) ok = ok & ((unsigned)(iw0 + dw) < (unsigned)p.W);
      const unsigned voff = (bbase + (unsigned)off * 4u) | (ok ? 0u : 0xffffffffu);
      breg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (int)voff, 0, 0));
    }
  };
  auto store_tiles = [&](int buf) __attribute__((always_inline)) {
    if (AVEC) {
#pragma unroll
      for (int i = 0; i < A_PER; ++i) *reinterpret_cast<float4*>(&As[buf][ga + 32 * i][kq * 4]) = avec[i];
    } else {
#pragma unroll
      for (int i = 0; i < A_PER; ++i) As[buf][g + 8 * i][kl] = areg[i];
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) Bs[buf][g + 8 * j][kl] = breg[j];
  };

  if (kt0 < kt1) {
    load_tiles(kt0);
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) load_tiles(kt + 1);            // in flight during the MFMA phase below
#pragma unroll
    for (int t = 0; t < WBK / 8; ++t) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][wm * (TM * 32) + i * 32 + ll][8 * t + 4 * lh]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[buf][wn * (TN * 32) + j * 32 + ll][8 * t + 4 * lh]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < kt1) store_tiles(buf ^ 1);
    __syncthreads();
  }

  float* out = slab + (long long)split * p.K * p.Kred;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = tileN * BNW + wn * (TN * 32) + j * 32 + ll;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = tileM * BM + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.K && n < p.Kred) out[(long long)m * p.Kred + n] = acc[i][j][r];
      }
  }
}

// dw[i] (+)= sum_s slab[s][i]   (fixed order: deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n,
                                     int splits, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += slab[(long long)k * n + i];
  if (accumulate) s += dw[i];
  dw[i] = s;
}

__global__ void bias_grad_kernel(const float* __restrict__ dy, long long N, long long K, long long SP,
                                 float* __restrict__ db, int accumulate) {
  __shared__ float sh[4];
  const long long k = blockIdx.x;
  float s = 0.f;
  const long long tot = N * SP;
  for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
    const long long n = i / SP, sp = i - n * SP;
    s += dy[(n * K + k) * SP + sp];
  }
  s = gca_block_sum256(s, sh);
  if (threadIdx.x == 0) db[k] = accumulate ? db[k] + s : s;
}

void wgrad_plan(const gca_conv_geom* g, WgradParams& p, bool& bm64, bool& bn64, bool& avec) {
  p.C = g->C; p.D = g->D; p.H = g->H; p.W = g->W; p.K = g->K;
  p.OD = g->OD; p.OH = g->OH; p.OW = g->OW;
  p.sd = g->sd; p.sh = g->sh; p.sw = g->sw; p.pd = g->pd; p.ph = g->ph; p.pw = g->pw;
  p.Kred = g->C * taps(g);
  const long long osp = (long long)g->OD * g->OH * g->OW;
  p.Ktot = (unsigned)((long long)g->N * osp);
  const long long cdhw = (long long)g->C * g->D * g->H * g->W;
  p.x_nstride = (unsigned)(g->x_batch_stride ? g->x_batch_stride : cdhw);
  const long long xb = (long long)g->N * p.x_nstride * 4, yb = (long long)g->N * g->K * osp * 4;
  p.x_bytes = xb > 0xfffff000LL ? 0xfffff000u : (unsigned)xb;
  p.dy_bytes = yb > 0xfffff000LL ? 0xfffff000u : (unsigned)yb;
  p.m_osp = make_magic((unsigned)osp);
  p.m_ohw = make_magic((unsigned)(g->OH * g->OW));
  p.m_ow = make_magic((unsigned)g->OW);
  avec = osp % 4 == 0;
  auto small = [](int dk) {
    if (dk <= 64) return true;
    const int t128 = (int)gca_ceil_div(dk, 128) * 128, t64 = (int)gca_ceil_div(dk, 64) * 64;
    return (t128 - dk) * 4 > t128 && t64 < t128;
  };
  bm64 = small(g->K);
  bn64 = small(p.Kred);
  p.tilesM = (int)gca_ceil_div(g->K, bm64 ? 64 : 128);
  p.tilesN = (int)gca_ceil_div(p.Kred, bn64 ? 64 : 128);
  p.kt_total = (int)gca_ceil_div((long long)p.Ktot, WBK);
  const long long tiles = (long long)p.tilesM * p.tilesN;
  long long want = gca_ceil_div(1024, tiles);                 // aim for ~4 workgroups per CU
  long long maxs = p.kt_total / 4 > 0 ? p.kt_total / 4 : 1;   // >= 4 k-tiles per split
  if (want > maxs) want = maxs;
  if (want > 512) want = 512;
  if (want < 1) want = 1;
  if (g->tune_wgrad_splits > 0) want = g->tune_wgrad_splits < p.kt_total ? g->tune_wgrad_splits : p.kt_total;
  p.kt_per_split = (int)gca_ceil_div(p.kt_total, want);
  p.splits = (int)gca_ceil_div(p.kt_total, p.kt_per_split);
  p.chk = ((g->pd > 0 || (g->OD - 1) * g->sd + g->kd > g->D) ? 1 : 0) |
          ((g->ph > 0 || (g->OH - 1) * g->sh + g->kh > g->H) ? 2 : 0) |
          ((g->pw > 0 || (g->OW - 1) * g->sw + g->kw > g->W) ? 4 : 0);
}

template <int BM, int BNW>
void launch_w(bool avec, dim3 grid, hipStream_t st, const float* x, const float* dy, const int2* t, float* slab,
              const WgradParams& p) {
  if (avec) hipLaunchKernelGGL((conv_wgrad_kernel<BM, BNW, true>), grid, dim3(256), 0, st, x, dy, t, slab, p);
  else hipLaunchKernelGGL((conv_wgrad_kernel<BM, BNW, false>), grid, dim3(256), 0, st, x, dy, t, slab, p);
}

}  // namespace

extern "C" {

int64_t gca_conv_wgrad_ws_bytes(const gca_conv_geom* g) {
  if (!geom_ok(g)) return GCA_EINVAL;
  WgradParams p{}; bool a, b, v;
  wgrad_plan(g, p, a, b, v);
  return (int64_t)p.splits * g->K * p.Kred * (int64_t)sizeof(float);
}

int gca_conv_wgrad(const gca_conv_geom* g, const float* x, const float* dy, const int32_t* table,
                   float* dw, int accumulate, void* ws, void* stream) {
  if (!geom_ok(g) || !x || !dy || !table || !dw || !ws) return GCA_EINVAL;
  WgradParams p{}; bool bm64, bn64, avec;
  wgrad_plan(g, p, bm64, bn64, avec);
  hipStream_t st = (hipStream_t)stream;
  const long long nblk = (long long)p.tilesM * p.tilesN * p.splits;
  if (nblk > 0x7fffffffLL) return GCA_EINVAL;
  const int2* t = reinterpret_cast<const int2*>(table);
  float* slab = reinterpret_cast<float*>(ws);
  dim3 grid((unsigned)nblk);
  if (bm64 && bn64) launch_w<64, 64>(avec, grid, st, x, dy, t, slab, p);
  else if (bm64) launch_w<64, 128>(avec, grid, st, x, dy, t, slab, p);
  else if (bn64) launch_w<128, 64>(avec, grid, st, x, dy, t, slab, p);
  else launch_w<128, 128>(avec, grid, st, x, dy, t, slab, p);
  int rc = gca_launch_status();
  if (rc) return rc;
  const long long n = (long long)g->K * p.Kred;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)gca_ceil_div(n, 256)), dim3(256), 0, st, slab, dw, n,
                     p.splits, accumulate ? 1 : 0);
  return gca_launch_status();
}

int gca_bias_grad(const float* dy, int64_t N, int64_t K, int64_t SP, float* db, int accumulate, void* stream) {
  if (!dy || !db || N <= 0 || K <= 0 || SP <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, dy, (long long)N,
                     (long long)K, (long long)SP, db, accumulate ? 1 : 0);
  return gca_launch_status();
}

}  // extern "C"
