// 3D convolution for gfx950 as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 64 FLOP/clk/SIMD).
//
//   forward : Y[n,ko,o]   = sum_{c,tap} W[ko,c,tap] * X[n,c,o*s-p+tap]        M=Ko   N=n*o      K=c*tap
//   dgrad   : dX[n,c,i]   = sum_{ko,tap} W[ko,c,tap] * dY[n,ko,(i+p-tap)/s]   M=C    N=n*i      K=ko*tap
//   wgrad   : dW[ko,c,tap]= sum_{n,o}  dY[n,ko,o] * X[n,c,o*s-p+tap]          M=Ko   N=c*tap    K=n*o (split)
//
// Activations stay NCDHW: for a fixed GEMM-K row (channel, tap) the GEMM-N direction is the
// W axis of the image, so a wave's 64 lanes read 64 consecutive floats (coalesced along W).
// Nothing is materialised: the im2col window is gathered straight into LDS through a small
// per-row table (element offset + tap id).  Window bounds are resolved ONCE per thread into a
// 64-bit tap-validity mask (its GEMM column is fixed for the whole K loop), so the gather in the
// hot loop is: bit test, index select, one dword load -- no branches, no per-element compares.
// The weight operand is pre-packed k-major and zero padded (gca_conv_pack) so its tile loads are
// unpredicated float4.  Layers whose output grid cannot fill 256 CUs split the K loop over
// workgroups (fp32 partial slabs + a deterministic finishing pass that also emits the BN sums).
//
// Reference call sites replaced: every nn.Conv3d / nn.Linear on the path (see include/gca_hip.h).
#include "gca_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;          // GEMM-K tile of the forward/dgrad kernels
constexpr int BN = 128;         // GEMM-N (spatial) tile of the forward/dgrad kernels
constexpr int WBK = 32;         // GEMM-K (spatial) tile of the wgrad kernel
constexpr int MPAD = 64;        // packed weights: M padded to this
constexpr int TABLE_PAD_W = 128;
constexpr int FAST_MAX_TAPS = 62;
constexpr int NUM_CU = 256;

// row table entry: x = element offset; y = dd | dh<<8 | dw<<16 (signed bytes) | valid<<24 | tap6<<25
__device__ __forceinline__ void decode_row(int2 e, int& off, int& dd, int& dh, int& dw, int& valid) {
  off = e.x;
  dd = (e.y << 24) >> 24;
  dh = (e.y << 16) >> 24;
  dw = (e.y << 8) >> 24;
  valid = (e.y >> 24) & 1;
}

struct IgemmParams {
  int NB;                  // batch
  int SC, SD, SH, SW;      // gathered (source) tensor: channels + spatial dims
  int DK;                  // destination channels (GEMM M, un-padded)
  int OD, OH, OW;          // destination spatial dims (GEMM N = NB*OD*OH*OW)
  int m_d, m_h, m_w;       // MODE 0: src = dst*m + o + delta ; MODE 1: src = (dst + o + delta)/m
  int o_d, o_h, o_w;
  int kd, kh, kw, tap_sign;// tap enumeration for the per-thread validity mask (delta = sign * (a,b,c))
  int Kpad, Mpad;
  int tilesM, tilesN;
  int splits, kt_per_split;
  int P;                   // stat partials per channel
  int force_bm, force_splits;   // host-side tuning overrides (0 = heuristic)
  int chk;                 // bit0: test D, bit1: test H, bit2: test W
  int accumulate;
  long long Ntot;
  long long src_nstride;   // elements between consecutive images of the gathered tensor
};

// ---------------------------------------------------------------------------------------------
// forward / dgrad implicit GEMM.  256 threads = 4 waves laid out WM x WN; each wave owns
// (BM/WM) x (BN/WN) of the block tile as TM x TN MFMA 32x32 tiles.
// MODE 0: linear gather (forward, unit-stride dgrad).  MODE 1: divide gather (strided dgrad).
// FAST: taps <= 62 and MODE 0 -> tap-mask path.
// ---------------------------------------------------------------------------------------------
template <int BM, int WM, int WN, int MODE, bool FAST>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const float* __restrict__ src, const float* __restrict__ apack, const int2* __restrict__ table,
    const float* __restrict__ bias, float* __restrict__ dst, float* __restrict__ psum,
    float* __restrict__ psq, float* __restrict__ slab, IgemmParams p) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_F4 = BK * BM / 4 / 256;     // float4 loads per thread for the A tile
  constexpr int B_PER = BK * BN / 256;        // gathers per thread for the B tile (8)
  static_assert(A_F4 >= 1, "tile too small");
  static_assert(!(FAST && MODE == 1), "fast path is linear-gather only");

  __shared__ float As[2][BK][BM];
  __shared__ float Bs[2][BK][BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid % p.splits; bid /= p.splits;
  const int tileM = bid % p.tilesM, tileN = bid / p.tilesM;

  // ---- per-thread column (GEMM-N) state: fixed for the whole K loop
  const int col = tid % BN;
  const int r0 = (tid / BN) * B_PER;          // this thread's first B row inside a k-tile (wave-uniform)
  const long long ng = (long long)tileN * BN + col;
  const bool cvalid = ng < p.Ntot;
  const int OSP = p.OD * p.OH * p.OW, OHW = p.OH * p.OW;
  const int SHW = p.SH * p.SW;
  int id0 = 0, ih0 = 0, iw0 = 0;
  int colbase = 0;                            // 32-bit element index (tensors are < 2^30 elements)
  {
    const long long ngc = cvalid ? ng : 0;
    const int img = (int)(ngc / OSP);
    const int sp = (int)(ngc - (long long)img * OSP);
    const int od = sp / OHW, r = sp - od * OHW;
    const int oh = r / p.OW, ow = r - oh * p.OW;
    if (MODE == 0) {
      id0 = od * p.m_d + p.o_d; ih0 = oh * p.m_h + p.o_h; iw0 = ow * p.m_w + p.o_w;
      colbase = (int)((long long)img * p.src_nstride) + id0 * SHW + ih0 * p.SW + iw0;
    } else {
      id0 = od + p.o_d; ih0 = oh + p.o_h; iw0 = ow + p.o_w;
      colbase = (int)((long long)img * p.src_nstride);
    }
  }
  const bool chkD = p.chk & 1, chkH = p.chk & 2, chkW = p.chk & 4;

  // tap-validity mask: bit t = tap t of this column's window lies inside the source tensor
  unsigned mlo = 0, mhi = 0;
  if (FAST) {
    int a = 0, b = 0, c = 0;
    const int ntaps = p.kd * p.kh * p.kw;
    for (int t = 0; t < ntaps; ++t) {
      bool ok = cvalid;
      if (chkD) ok = ok & ((unsigned)(id0 + p.tap_sign * a) < (unsigned)p.SD);
      if (chkH) ok = ok & ((unsigned)(ih0 + p.tap_sign * b) < (unsigned)p.SH);
      if (chkW) ok = ok & ((unsigned)(iw0 + p.tap_sign * c) < (unsigned)p.SW);
      if (t < 32) mlo |= (unsigned)ok << t; else mhi |= (unsigned)ok << (t - 32);
      if (++c == p.kw) { c = 0; if (++b == p.kh) { b = 0; ++a; } }
    }
  }

  float breg[B_PER];
  float4 areg[A_F4];
  unsigned bmask = 0;

  auto load_tiles = [&](int kt) {
    // A: packed weights [Kpad][Mpad], rows kt*BK.., cols tileM*BM..
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (BM / 4), c4 = idx % (BM / 4);
      areg[i] = *reinterpret_cast<const float4*>(apack + (long long)(kt * BK + row) * p.Mpad + tileM * BM + c4 * 4);
    }
    // B: gathered window.  Rows are uniform across the wave: fetch their table entries first (scalar,
    // one batch), then issue all gathers back to back.
    const int2* trow = table + kt * BK + __builtin_amdgcn_readfirstlane(r0);
    int2 e[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) e[i] = trow[i];
    bmask = 0;      // validity bits; applied when the tile is written to LDS, so the loads stay in flight
    if (FAST) {
      int idx[B_PER];
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int tap6 = (e[i].y >> 25) & 63;            // 63 = padded row, never valid
        const unsigned m = tap6 < 32 ? mlo : mhi;
        const unsigned ok = (m >> (tap6 & 31)) & 1u;
        bmask |= ok << i;
        idx[i] = ok ? colbase + e[i].x : 0;
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) breg[i] = src[(unsigned)idx[i]];
    } else {
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        int off, dd, dh, dw, rvalid;
        decode_row(e[i], off, dd, dh, dw, rvalid);
        bool ok = cvalid & (rvalid != 0);
        int idx;
        if (MODE == 0) {
          if (chkD) ok = ok & ((unsigned)(id0 + dd) < (unsigned)p.SD);
          if (chkH) ok = ok & ((unsigned)(ih0 + dh) < (unsigned)p.SH);
          if (chkW) ok = ok & ((unsigned)(iw0 + dw) < (unsigned)p.SW);
          idx = colbase + off;
        } else {
          const unsigned td = (unsigned)(id0 + dd), th = (unsigned)(ih0 + dh), tw = (unsigned)(iw0 + dw);
          unsigned qd, qh, qw;
          if (p.m_d == 1) qd = td; else if (p.m_d == 2) { qd = td >> 1; ok = ok & !(td & 1); } else { qd = td / (unsigned)p.m_d; ok = ok & (qd * p.m_d == td); }
          if (p.m_h == 1) qh = th; else if (p.m_h == 2) { qh = th >> 1; ok = ok & !(th & 1); } else { qh = th / (unsigned)p.m_h; ok = ok & (qh * p.m_h == th); }
          if (p.m_w == 1) qw = tw; else if (p.m_w == 2) { qw = tw >> 1; ok = ok & !(tw & 1); } else { qw = tw / (unsigned)p.m_w; ok = ok & (qw * p.m_w == tw); }
          ok = ok & (qd < (unsigned)p.SD) & (qh < (unsigned)p.SH) & (qw < (unsigned)p.SW);
          idx = colbase + off + (int)qd * SHW + (int)qh * p.SW + (int)qw;
        }
        bmask |= (unsigned)ok << i;
        breg[i] = src[(unsigned)(ok ? idx : 0)];
      }
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (BM / 4), c4 = idx % (BM / 4);
      *reinterpret_cast<float4*>(&As[buf][row][c4 * 4]) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) Bs[buf][r0 + i][col] = ((bmask >> i) & 1u) ? breg[i] : 0.f;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.Kpad / BK;
  const int kt0 = split * p.kt_per_split;
  int kt1 = kt0 + p.kt_per_split; if (kt1 > nk) kt1 = nk;
  const int lh = lane >> 5, ll = lane & 31;
  if (kt0 < kt1) {
    load_tiles(kt0);
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) load_tiles(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][kk + lh][wm * (TM * 32) + i * 32 + ll];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kk + lh][wn * (TN * 32) + j * 32 + ll];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < kt1) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int mbase = tileM * BM + wm * (TM * 32);
  if (p.splits > 1) {
    // partial tile -> slab[split][m][n]; bias / accumulate / BN sums happen in conv_splitk_finish_kernel
    float* sl = slab + (long long)split * p.DK * p.Ntot;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long long n = (long long)tileN * BN + wn * (TN * 32) + j * 32 + ll;
      if (n < p.Ntot) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < p.DK) sl[(long long)m * p.Ntot + n] = acc[i][j][r];
          }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long long n = (long long)tileN * BN + wn * (TN * 32) + j * 32 + ll;
    const bool nv = n < p.Ntot;
    const long long nc = nv ? n : 0;
    const int img = (int)(nc / OSP);
    const int sp = (int)(nc - (long long)img * OSP);
    float* d0 = dst + ((long long)img * p.DK) * OSP + sp;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float old[16];
      if (p.accumulate) {                       // all 16 read-modify-write loads in flight together
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          old[r] = (nv && m < p.DK) ? d0[(long long)m * OSP] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (nv && m < p.DK) {
          float v = acc[i][j][r];
          if (bias) v += bias[m];
          if (p.accumulate) v += old[r];
          d0[(long long)m * OSP] = v;
        }
      }
    }
  }
  if (psum) {
    // per-channel partial sum / sum of squares over this wave's columns (invalid columns hold 0)
    const int part = tileN * WN + wn;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) { const float v = acc[i][j][r]; s += v; q += v * v; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ll == 0 && m < p.DK) {
          psum[(long long)m * p.P + part] = s;
          psq[(long long)m * p.P + part] = q;
        }
      }
    }
  }
}

// Split-K finishing pass, grid (channels, parts): sum the slabs in fixed order, add bias, (+=) store in
// NCDHW, and emit the BN partial sums [K][parts] of the conv output.
constexpr int FINISH_CHUNK = 4096;
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(
    const float* __restrict__ slab, int splits, const float* __restrict__ bias, float* __restrict__ dst,
    float* __restrict__ psum, float* __restrict__ psq, int DK, int OSP, long long Ntot, int accumulate) {
  __shared__ double sh[4];
  const int m = blockIdx.x, part = blockIdx.y, P = gridDim.y;
  const float b = bias ? bias[m] : 0.f;
  const long long lo = (long long)part * FINISH_CHUNK;
  long long hi = lo + FINISH_CHUNK; if (hi > Ntot) hi = Ntot;
  double s = 0.0, q = 0.0;
  for (long long n = lo + threadIdx.x; n < hi; n += 256) {
    float v = 0.f;
    for (int k = 0; k < splits; ++k) v += slab[((long long)k * DK + m) * Ntot + n];
    s += (double)v; q += (double)v * (double)v;       // statistics of the conv output proper (no bias on that path)
    v += b;
    const long long img = n / OSP, sp = n - img * OSP;
    float* d = dst + (img * DK + m) * OSP + sp;
    if (accumulate) v += *d;
    *d = v;
  }
  if (psum) {
    s = gca_block_sum256_d(s, sh);
    q = gca_block_sum256_d(q, sh);
    if (threadIdx.x == 0) { psum[(long long)m * P + part] = (float)s; psq[(long long)m * P + part] = (float)q; }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad: both operands are gathered with lanes along the (contiguous) spatial axis.
//   A[k'][m] = dY[img, m, o]       B[k'][n'] = X[img, c(n'), o*s - p + tap(n')]
// One workgroup = one (tileM, tileN, split) and writes its partial tile to a slab.
// ---------------------------------------------------------------------------------------------
struct WgradParams {
  int NB, C, D, H, W, K, OD, OH, OW;
  int sd, sh, sw, pd, ph, pw;
  int Kred;                 // C*taps  (GEMM N)
  int tilesM, tilesN, splits;
  int kt_per_split, kt_total;
  int chk;
  long long Ktot;           // NB*OD*OH*OW (GEMM K)
  long long x_nstride;
};

template <int BM, int BNW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const int2* __restrict__ table,
    float* __restrict__ slab, WgradParams p) {
  constexpr int WM = 2, WN = 2;
  constexpr int TM = BM / WM / 32, TN = BNW / WN / 32;
  constexpr int LDA = BM + 1, LDB = BNW + 1;     // odd strides: lanes run along k' on the LDS write
  constexpr int A_PER = BM / 8, B_PER = BNW / 8;

  __shared__ float As[WBK * LDA];
  __shared__ float Bs[WBK * LDB];
  __shared__ int2 Ts[BNW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = blockIdx.x;
  const int split = bid % p.splits; bid /= p.splits;
  const int tileM = bid % p.tilesM, tileN = bid / p.tilesM;

  if (tid < BNW) Ts[tid] = table[tileN * BNW + tid];
  __syncthreads();

  const int kl = tid & 31, g = tid >> 5;
  const int OSP = p.OD * p.OH * p.OW, OHW = p.OH * p.OW;
  const int HW = p.H * p.W;
  const bool chkD = p.chk & 1, chkH = p.chk & 2, chkW = p.chk & 4;

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

  float areg[A_PER], breg[B_PER];
  unsigned amask = 0, bmask = 0;     // validity bits, applied at the LDS store so the loads stay in flight
  auto load_tiles = [&](int kt) {
    const long long kp = (long long)kt * WBK + kl;
    const bool kv = kp < p.Ktot;
    const long long kc = kv ? kp : 0;
    const int img = (int)(kc / OSP);
    const int o = (int)(kc - (long long)img * OSP);
    const int od = o / OHW, r = o - od * OHW;
    const int oh = r / p.OW, ow = r - oh * p.OW;
    // A: dY[img, m, o]  (32-bit element indices; tensors are < 2^30 elements)
    const int abase = img * p.K * OSP + o;
    int aidx[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int m = tileM * BM + g + 8 * i;
      aidx[i] = (kv & (m < p.K)) ? abase + m * OSP : -1;
    }
    amask = 0;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) { amask |= (unsigned)(aidx[i] >= 0) << i; areg[i] = dy[(unsigned)(aidx[i] < 0 ? 0 : aidx[i])]; }
    // B: X window element for (c, tap) = table row n'
    const int id0 = od * p.sd - p.pd, ih0 = oh * p.sh - p.ph, iw0 = ow * p.sw - p.pw;
    const int bbase = (int)((long long)img * p.x_nstride) + id0 * HW + ih0 * p.W + iw0;
    int bidx[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int2 e = Ts[g + 8 * j];
      int off, dd, dh, dw, rvalid;
      decode_row(e, off, dd, dh, dw, rvalid);
      bool ok = kv & (rvalid != 0);
      if (chkD) ok = ok & ((unsigned)(id0 + dd) < (unsigned)p.D);
      if (chkH) ok = ok & ((unsigned)(ih0 + dh) < (unsigned)p.H);
      if (chkW) ok = ok & ((unsigned)(iw0 + dw) < (unsigned)p.W);
      bidx[j] = ok ? bbase + off : -1;
    }
    bmask = 0;
#pragma unroll
    for (int j = 0; j < B_PER; ++j) { bmask |= (unsigned)(bidx[j] >= 0) << j; breg[j] = x[(unsigned)(bidx[j] < 0 ? 0 : bidx[j])]; }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) As[kl * LDA + g + 8 * i] = ((amask >> i) & 1u) ? areg[i] : 0.f;
#pragma unroll
    for (int j = 0; j < B_PER; ++j) Bs[kl * LDB + g + 8 * j] = ((bmask >> j) & 1u) ? breg[j] : 0.f;
  };

  if (kt0 < kt1) load_tiles(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();                 // previous tile fully consumed
    store_tiles();
    __syncthreads();
    if (kt + 1 < kt1) load_tiles(kt + 1);   // overlaps the MFMA phase below
#pragma unroll
    for (int kk = 0; kk < WBK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(kk + lh) * LDA + wm * (TM * 32) + i * 32 + ll];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + lh) * LDB + wn * (TN * 32) + j * 32 + ll];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
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

// packed[k][m] = W[(k / T) * s_kq + (k % T) + m * s_m]  (zero outside k<Kred, m<M), 32x32 LDS transpose
__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int Kred, int M,
                                 int Kpad, int Mpad, int T, long long s_kq, long long s_m) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: ty 0..7
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, k = k0 + tx;
    float v = 0.f;
    if (m < M && k < Kred) v = w[(long long)(k / T) * s_kq + (k % T) + (long long)m * s_m];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, m = m0 + tx;
    if (k < Kpad && m < Mpad) packed[(long long)k * Mpad + m] = tile[tx][r];
  }
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

inline bool geom_ok(const gca_conv_geom* g) {
  if (!g) return false;
  if (g->N <= 0 || g->C <= 0 || g->D <= 0 || g->H <= 0 || g->W <= 0 || g->K <= 0) return false;
  if (g->kd <= 0 || g->kh <= 0 || g->kw <= 0 || g->sd <= 0 || g->sh <= 0 || g->sw <= 0) return false;
  if (g->pd < 0 || g->ph < 0 || g->pw < 0) return false;
  if (g->kd > 127 || g->kh > 127 || g->kw > 127) return false;
  const int od = (g->D + 2 * g->pd - g->kd) / g->sd + 1;
  const int oh = (g->H + 2 * g->ph - g->kh) / g->sh + 1;
  const int ow = (g->W + 2 * g->pw - g->kw) / g->sw + 1;
  if (od != g->OD || oh != g->OH || ow != g->OW || od <= 0 || oh <= 0 || ow <= 0) return false;
  // the kernels address both tensors with 32-bit byte offsets from their base: < 2^30 elements (4 GiB) each
  const long long cdhw = (long long)g->C * g->D * g->H * g->W;
  if (g->x_batch_stride != 0 && g->x_batch_stride < cdhw) return false;
  const long long in_elems = (long long)g->N * (g->x_batch_stride ? g->x_batch_stride : cdhw);
  const long long out_elems = (long long)g->N * g->K * od * oh * ow;
  if (in_elems >= (1LL << 30) || out_elems >= (1LL << 30)) return false;
  for (int v : {g->tune_fwd_bm, g->tune_dgrad_bm}) if (v != 0 && v != 64 && v != 128) return false;
  for (int v : {g->tune_fwd_splits, g->tune_dgrad_splits, g->tune_wgrad_splits}) if (v < 0 || v > 1024) return false;
  return true;
}

inline int taps(const gca_conv_geom* g) { return g->kd * g->kh * g->kw; }
inline bool unit_stride(const gca_conv_geom* g) { return g->sd == 1 && g->sh == 1 && g->sw == 1; }

// ---- launch configuration ------------------------------------------------------------------
struct IgemmCfg { int bm; int splits; int kt_per_split; };

// Heuristic default (the host side may override it per geometry after measuring: gca_conv_geom.tune_*).
// BM = 128 halves the gather work per FLOP (a gathered B element feeds 128 output channels instead of 64)
// but only two such workgroups fit a CU; the K loop is split when the tile grid cannot occupy the CUs and
// the partial slabs stay small.
inline IgemmCfg choose_cfg(int M, long long Ntot, int nk, int force_bm, int force_splits) {
  const long long tn = gca_ceil_div(Ntot, BN);
  IgemmCfg best{64, 1, nk};
  double best_cost = 1e300;
  for (int bm = 64; bm <= 128; bm += 64) {
    if (force_bm ? bm != force_bm : (bm == 128 && M <= 64)) continue;
    const long long tiles = gca_ceil_div(M, bm) * tn;
    int s = 1;
    if (force_splits > 0) s = force_splits;
    else if (tiles < 2 * NUM_CU && nk >= 8 && (long long)M * Ntot <= (1LL << 20)) {
      long long want = gca_ceil_div(2 * NUM_CU, tiles);
      if (want > nk / 4) want = nk / 4;
      if (want > 16) want = 16;
      if (want > 1) s = (int)want;
    }
    if (s > nk) s = nk;
    if (s < 1) s = 1;
    const int per = (int)gca_ceil_div(nk, s);
    s = (int)gca_ceil_div(nk, per);
    const double wg_per_cu = (double)(tiles * s) / NUM_CU;
    const double occ = wg_per_cu < 1.0 ? 0.55 : (wg_per_cu < 2.0 ? 0.75 : 1.0);   // latency hiding needs >= 2 WGs / CU
    const double rounds = (double)gca_ceil_div(tiles * s, NUM_CU);
    const double eff = (bm == 128 ? 1.0 : 0.85) * occ;
    const double cost = rounds * bm * (double)per / eff + (s > 1 ? 0.05 * rounds * bm * per + 8.0 * bm : 0.0);
    if (cost < best_cost) { best_cost = cost; best = IgemmCfg{bm, s, per}; }
  }
  return best;
}

template <int BM, int MODE, bool FAST>
void launch_one(dim3 grid, hipStream_t st, const float* src, const float* apack, const int2* table, const float* bias,
                float* dst, float* psum, float* psq, float* slab, const IgemmParams& p) {
  hipLaunchKernelGGL((conv_igemm_kernel<BM, 2, 2, MODE, FAST>), grid, dim3(256), 0, st, src, apack, table, bias, dst,
                     psum, psq, slab, p);
}

int run_igemm(int mode, bool fast, const float* src, const float* apack, const int2* table, const float* bias,
              float* dst, float* psum, float* psq, float* slab, IgemmParams p, hipStream_t st) {
  const IgemmCfg c = choose_cfg(p.DK, p.Ntot, p.Kpad / BK, p.force_bm, p.force_splits);
  p.tilesM = (int)gca_ceil_div(p.DK, c.bm);
  p.tilesN = (int)gca_ceil_div(p.Ntot, BN);
  p.splits = c.splits; p.kt_per_split = c.kt_per_split;
  p.P = c.splits > 1 ? (int)gca_ceil_div(p.Ntot, FINISH_CHUNK) : p.tilesN * 2;
  if (c.splits > 1 && !slab) return GCA_EINVAL;
  const long long nblk = (long long)p.tilesM * p.tilesN * c.splits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return GCA_EINVAL;
  dim3 grid((unsigned)nblk);
  float* ps = c.splits > 1 ? nullptr : psum;
  float* pq = c.splits > 1 ? nullptr : psq;
  if (c.bm == 64) {
    if (mode == 1) launch_one<64, 1, false>(grid, st, src, apack, table, bias, dst, ps, pq, slab, p);
    else if (fast) launch_one<64, 0, true>(grid, st, src, apack, table, bias, dst, ps, pq, slab, p);
    else launch_one<64, 0, false>(grid, st, src, apack, table, bias, dst, ps, pq, slab, p);
  } else {
    if (mode == 1) launch_one<128, 1, false>(grid, st, src, apack, table, bias, dst, ps, pq, slab, p);
    else if (fast) launch_one<128, 0, true>(grid, st, src, apack, table, bias, dst, ps, pq, slab, p);
    else launch_one<128, 0, false>(grid, st, src, apack, table, bias, dst, ps, pq, slab, p);
  }
  int rc = gca_launch_status();
  if (rc || c.splits == 1) return rc;
  hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)p.DK, (unsigned)p.P), dim3(256), 0, st, slab, c.splits, bias, dst, psum,
                     psq, p.DK, p.OD * p.OH * p.OW, p.Ntot, p.accumulate);
  return gca_launch_status();
}

void fwd_params(const gca_conv_geom* g, IgemmParams& p) {
  p.NB = g->N; p.SC = g->C; p.SD = g->D; p.SH = g->H; p.SW = g->W; p.DK = g->K;
  p.OD = g->OD; p.OH = g->OH; p.OW = g->OW;
  p.m_d = g->sd; p.m_h = g->sh; p.m_w = g->sw; p.o_d = -g->pd; p.o_h = -g->ph; p.o_w = -g->pw;
  p.kd = g->kd; p.kh = g->kh; p.kw = g->kw; p.tap_sign = 1;
  p.Kpad = (int)gca_round_up((int64_t)g->C * taps(g), BK);
  p.Mpad = (int)gca_round_up(g->K, MPAD);
  p.Ntot = (long long)g->N * g->OD * g->OH * g->OW;
  // a dimension needs the bounds test unless every window stays inside by construction
  p.chk = ((g->pd > 0 || (g->OD - 1) * g->sd + g->kd > g->D) ? 1 : 0) |
          ((g->ph > 0 || (g->OH - 1) * g->sh + g->kh > g->H) ? 2 : 0) |
          ((g->pw > 0 || (g->OW - 1) * g->sw + g->kw > g->W) ? 4 : 0);
  p.accumulate = 0;
  p.src_nstride = g->x_batch_stride ? g->x_batch_stride : (long long)g->C * g->D * g->H * g->W;
  p.force_bm = g->tune_fwd_bm; p.force_splits = g->tune_fwd_splits;
}

void dgrad_params(const gca_conv_geom* g, IgemmParams& p, int& mode) {
  p.NB = g->N; p.SC = g->K; p.SD = g->OD; p.SH = g->OH; p.SW = g->OW; p.DK = g->C;
  p.OD = g->D; p.OH = g->H; p.OW = g->W;
  p.m_d = g->sd; p.m_h = g->sh; p.m_w = g->sw; p.o_d = g->pd; p.o_h = g->ph; p.o_w = g->pw;
  p.kd = g->kd; p.kh = g->kh; p.kw = g->kw; p.tap_sign = -1;
  p.Kpad = (int)gca_round_up((int64_t)g->K * taps(g), BK);
  p.Mpad = (int)gca_round_up(g->C, MPAD);
  p.Ntot = (long long)g->N * g->D * g->H * g->W;
  p.src_nstride = (long long)g->K * g->OD * g->OH * g->OW;
  p.force_bm = g->tune_dgrad_bm; p.force_splits = g->tune_dgrad_splits;
  if (unit_stride(g)) {
    mode = 0; p.m_d = p.m_h = p.m_w = 1;
    p.chk = ((g->kd > 1 || g->pd > 0) ? 1 : 0) | ((g->kh > 1 || g->ph > 0) ? 2 : 0) | ((g->kw > 1 || g->pw > 0) ? 4 : 0);
  } else {
    mode = 1; p.chk = 7;
  }
}

}  // namespace

extern "C" {

int gca_version(void) { return 2; }

int64_t gca_conv_pack_elems(const gca_conv_geom* g, int which) {
  if (!geom_ok(g) || (which != 0 && which != 1)) return GCA_EINVAL;
  const int64_t kred = (which == 0 ? (int64_t)g->C : (int64_t)g->K) * taps(g);
  const int64_t m = which == 0 ? g->K : g->C;
  return gca_round_up(kred, BK) * gca_round_up(m, MPAD);
}

int gca_conv_pack(const gca_conv_geom* g, int which, const float* w, float* packed, void* stream) {
  if (!geom_ok(g) || (which != 0 && which != 1) || !w || !packed) return GCA_EINVAL;
  const int T = taps(g);
  int Kred, M, Tdiv; long long s_kq, s_m;
  if (which == 0) { Kred = g->C * T; M = g->K; Tdiv = Kred; s_kq = 0; s_m = Kred; }
  else { Kred = g->K * T; M = g->C; Tdiv = T; s_kq = (long long)g->C * T; s_m = T; }
  const int Kpad = (int)gca_round_up(Kred, BK), Mpad = (int)gca_round_up(M, MPAD);
  dim3 grid((unsigned)gca_ceil_div(Kpad, 32), (unsigned)gca_ceil_div(Mpad, 32));
  hipLaunchKernelGGL(conv_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, packed, Kred, M, Kpad, Mpad,
                     Tdiv, s_kq, s_m);
  return gca_launch_status();
}

int64_t gca_conv_table_rows(const gca_conv_geom* g, int which) {
  if (!geom_ok(g) || which < 0 || which > 2) return GCA_EINVAL;
  const int64_t kred = (which == 1 ? (int64_t)g->K : (int64_t)g->C) * taps(g);
  return gca_round_up(kred, which == 2 ? TABLE_PAD_W : BK);
}

int gca_conv_table_build_host(const gca_conv_geom* g, int which, int32_t* t) {
  if (!geom_ok(g) || which < 0 || which > 2 || !t) return GCA_EINVAL;
  const int T = taps(g);
  const int64_t rows = gca_conv_table_rows(g, which);
  const int64_t kred = (which == 1 ? (int64_t)g->K : (int64_t)g->C) * T;
  const int64_t HW = (int64_t)g->H * g->W, DHW = HW * g->D;
  const int64_t OHW = (int64_t)g->OH * g->OW, OSP = OHW * g->OD;
  const bool linear_dgrad = unit_stride(g);
  for (int64_t k = 0; k < rows; ++k) {
    int32_t off = 0, pk = 63 << 25;          // padded row: invalid, tap id 63
    if (k < kred) {
      const int ch = (int)(k / T), tap = (int)(k % T);
      const int a = tap / (g->kh * g->kw), r = tap % (g->kh * g->kw);
      const int b = r / g->kw, c = r % g->kw;
      int dd, dh, dw; int64_t o;
      if (which == 1) {
        dd = -a; dh = -b; dw = -c;
        o = (int64_t)ch * OSP;
        if (linear_dgrad) o += -(int64_t)a * OHW - (int64_t)b * g->OW - c;
      } else {
        dd = a; dh = b; dw = c;
        o = (int64_t)ch * DHW + (int64_t)a * HW + (int64_t)b * g->W + c;
      }
      off = (int32_t)o;
      const int tap6 = tap < 63 ? tap : 63;
      pk = (dd & 0xff) | ((dh & 0xff) << 8) | ((dw & 0xff) << 16) | (1 << 24) | (tap6 << 25);
    }
    t[2 * k] = off;
    t[2 * k + 1] = pk;
  }
  return GCA_OK;
}

int gca_conv_kernel_cfg(const gca_conv_geom* g, int which, int32_t* out4) {
  if (!geom_ok(g) || which < 0 || which > 1 || !out4) return GCA_EINVAL;
  IgemmParams p{}; int mode = 0;
  if (which == 0) fwd_params(g, p); else dgrad_params(g, p, mode);
  const IgemmCfg c = choose_cfg(p.DK, p.Ntot, p.Kpad / BK, p.force_bm, p.force_splits);
  out4[0] = c.bm; out4[1] = c.splits; out4[2] = mode; out4[3] = (mode == 0 && taps(g) <= FAST_MAX_TAPS) ? 1 : 0;
  return GCA_OK;
}

int64_t gca_conv_fwd_stat_parts(const gca_conv_geom* g) {
  if (!geom_ok(g)) return GCA_EINVAL;
  IgemmParams p{};
  fwd_params(g, p);
  const IgemmCfg c = choose_cfg(p.DK, p.Ntot, p.Kpad / BK, p.force_bm, p.force_splits);
  return c.splits > 1 ? gca_ceil_div(p.Ntot, FINISH_CHUNK) : gca_ceil_div(p.Ntot, BN) * 2;
}

int64_t gca_conv_fwd_ws_bytes(const gca_conv_geom* g) {
  if (!geom_ok(g)) return GCA_EINVAL;
  IgemmParams p{};
  fwd_params(g, p);
  const IgemmCfg c = choose_cfg(p.DK, p.Ntot, p.Kpad / BK, p.force_bm, p.force_splits);
  return c.splits > 1 ? (int64_t)c.splits * p.DK * p.Ntot * (int64_t)sizeof(float) : 0;
}

int64_t gca_conv_dgrad_ws_bytes(const gca_conv_geom* g) {
  if (!geom_ok(g)) return GCA_EINVAL;
  IgemmParams p{}; int mode;
  dgrad_params(g, p, mode);
  const IgemmCfg c = choose_cfg(p.DK, p.Ntot, p.Kpad / BK, p.force_bm, p.force_splits);
  return c.splits > 1 ? (int64_t)c.splits * p.DK * p.Ntot * (int64_t)sizeof(float) : 0;
}

int gca_conv_fwd(const gca_conv_geom* g, const float* x, const float* wpack, const int32_t* table,
                 const float* bias, float* y, float* stat_sum, float* stat_sq, void* ws, void* stream) {
  if (!geom_ok(g) || !x || !wpack || !table || !y) return GCA_EINVAL;
  if ((stat_sum == nullptr) != (stat_sq == nullptr)) return GCA_EINVAL;
  IgemmParams p{};
  fwd_params(g, p);
  return run_igemm(0, taps(g) <= FAST_MAX_TAPS, x, wpack, reinterpret_cast<const int2*>(table), bias, y, stat_sum,
                   stat_sq, reinterpret_cast<float*>(ws), p, (hipStream_t)stream);
}

int gca_conv_dgrad(const gca_conv_geom* g, const float* dy, const float* wpack, const int32_t* table,
                   float* dx, int accumulate, void* ws, void* stream) {
  if (!geom_ok(g) || !dy || !wpack || !table || !dx) return GCA_EINVAL;
  if (g->x_batch_stride != 0 && g->x_batch_stride != (long long)g->C * g->D * g->H * g->W) return GCA_EINVAL;
  IgemmParams p{}; int mode;
  dgrad_params(g, p, mode);
  p.accumulate = accumulate ? 1 : 0;
  return run_igemm(mode, mode == 0 && taps(g) <= FAST_MAX_TAPS, dy, wpack, reinterpret_cast<const int2*>(table),
                   nullptr, dx, nullptr, nullptr, reinterpret_cast<float*>(ws), p, (hipStream_t)stream);
}

static void wgrad_plan(const gca_conv_geom* g, WgradParams& p, bool& bm64, bool& bn64) {
  p.NB = g->N; p.C = g->C; p.D = g->D; p.H = g->H; p.W = g->W; p.K = g->K;
  p.OD = g->OD; p.OH = g->OH; p.OW = g->OW;
  p.sd = g->sd; p.sh = g->sh; p.sw = g->sw; p.pd = g->pd; p.ph = g->ph; p.pw = g->pw;
  p.Kred = g->C * taps(g);
  p.Ktot = (long long)g->N * g->OD * g->OH * g->OW;
  p.x_nstride = g->x_batch_stride ? g->x_batch_stride : (long long)g->C * g->D * g->H * g->W;
  auto small = [](int dk) {
    if (dk <= 64) return true;
    const int t128 = (int)gca_ceil_div(dk, 128) * 128, t64 = (int)gca_ceil_div(dk, 64) * 64;
    return (t128 - dk) * 4 > t128 && t64 < t128;
  };
  bm64 = small(g->K);
  bn64 = small(p.Kred);
  p.tilesM = (int)gca_ceil_div(g->K, bm64 ? 64 : 128);
  p.tilesN = (int)gca_ceil_div(p.Kred, bn64 ? 64 : 128);
  p.kt_total = (int)gca_ceil_div(p.Ktot, WBK);
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

int64_t gca_conv_wgrad_ws_bytes(const gca_conv_geom* g) {
  if (!geom_ok(g)) return GCA_EINVAL;
  WgradParams p{}; bool a, b;
  wgrad_plan(g, p, a, b);
  return (int64_t)p.splits * g->K * p.Kred * (int64_t)sizeof(float);
}

int gca_conv_wgrad(const gca_conv_geom* g, const float* x, const float* dy, const int32_t* table,
                   float* dw, int accumulate, void* ws, void* stream) {
  if (!geom_ok(g) || !x || !dy || !table || !dw || !ws) return GCA_EINVAL;
  WgradParams p{}; bool bm64, bn64;
  wgrad_plan(g, p, bm64, bn64);
  hipStream_t st = (hipStream_t)stream;
  const long long nblk = (long long)p.tilesM * p.tilesN * p.splits;
  if (nblk > 0x7fffffffLL) return GCA_EINVAL;
  const int2* t = reinterpret_cast<const int2*>(table);
  float* slab = reinterpret_cast<float*>(ws);
  dim3 grid((unsigned)nblk), blk(256);
  if (bm64 && bn64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, blk, 0, st, x, dy, t, slab, p);
  else if (bm64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128>), grid, blk, 0, st, x, dy, t, slab, p);
  else if (bn64) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64>), grid, blk, 0, st, x, dy, t, slab, p);
  else hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), grid, blk, 0, st, x, dy, t, slab, p);
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
