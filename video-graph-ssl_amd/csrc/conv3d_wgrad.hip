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
#include <type_traits>
#include "conv_common.h"

using namespace gca_conv;

namespace {

constexpr int LDW = WBK + 4;       // LDS row pitch in floats (144 B)

typedef gca_magic Magic;
inline Magic make_magic(unsigned d) { return gca_make_magic(d); }

struct WgradParams {
  int C, D, H, W, K, OD, OH, OW;
  int kd, kh, kw;
  int sd, sh, sw, pd, ph, pw;
  int Kred;                 // C*taps  (GEMM N)
  int tilesM, tilesN, splits;
  int kt_per_split, kt_total;
  int chk;
  int math;                 // arithmetic of this launch (host side only)
  int half;                 // x and dy are fp16 in HBM (host side only)
  unsigned Ktot;            // NB*OD*OH*OW (GEMM K) -- < 2^30
  unsigned x_nstride;       // elements between clips of x
  unsigned x_bytes, dy_bytes, slab_bytes;
  Magic m_osp, m_ohw, m_ow;
};

// Window validity of one spatial position as a bit per tap (bit = 1: tap INSIDE the input), built from per-axis
// bit rows: kd + kh + kw range tests instead of one test per gathered element.
template <typename MaskT>
__device__ __forceinline__ MaskT tap_valid_mask(const WgradParams& p, int id0, int ih0, int iw0, bool chkD, bool chkH, bool chkW) {
  MaskT vw = (MaskT(1) << p.kw) - 1, vh = (MaskT(1) << p.kh) - 1, vd = (MaskT(1) << p.kd) - 1;
  if (chkW) { vw = 0; for (int c = 0; c < p.kw; ++c) vw |= (MaskT)((unsigned)(iw0 + c) < (unsigned)p.W) << c; }
  if (chkH) { vh = 0; for (int b = 0; b < p.kh; ++b) vh |= (MaskT)((unsigned)(ih0 + b) < (unsigned)p.H) << b; }
  if (chkD) { vd = 0; for (int a = 0; a < p.kd; ++a) vd |= (MaskT)((unsigned)(id0 + a) < (unsigned)p.D) << a; }
  MaskT hw = vw;
  if (p.kh > 1) { hw = 0; for (int b = 0; b < p.kh; ++b) hw |= ((vh >> b) & 1) ? vw << (b * p.kw) : MaskT(0); }
  else if (!(vh & 1)) hw = 0;
  MaskT all = hw;
  if (p.kd > 1) { all = 0; const int khw = p.kh * p.kw; for (int a = 0; a < p.kd; ++a) all |= ((vd >> a) & 1) ? hw << (a * khw) : MaskT(0); }
  else if (!(vd & 1)) all = 0;
  return all;
}

// 4 waves arranged WM x WN; every wave owns TM x TN MFMA tiles: block tile = (32*WM*TM) x (32*WN*TN).
// FAST 1 (<= 31 taps) / 2 (<= 62 taps): the window test is done once per spatial position (a bit per tap) and
// a gathered element then costs v_bfe + add + or; FAST 0: every element tests its own (dd, dh, dw).
// BVEC (pointwise-in-space convs: kh = kw = 1, unit spatial stride, no spatial padding): a table row's 4
// consecutive positions are 4 consecutive floats of X, so the gather uses float4 loads too (and shares the
// position arithmetic with the dY loads).
// H: fp16 storage of x and dy (gca_conv_geom.act_f16), always with MATH 3: the stored halves are the operands of
// v_mfma_f32_32x32x16_f16 as they are (no widening, no split); dW is accumulated and written in fp32.
template <int WM, int WN, int TM, int TN, bool AVEC, int FAST, bool BVEC, int MATH, bool H = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const void* __restrict__ x, const void* __restrict__ dy, const int2* __restrict__ table,
    float* __restrict__ slab, WgradParams p) {
  constexpr unsigned ES = H ? 2u : 4u;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(32 * WN * TN <= TABLE_PAD_W, "the row table is padded for tiles up to TABLE_PAD_W columns");
  constexpr int BM = 32 * WM * TM, BNW = 32 * WN * TN;
  constexpr int A_PER = AVEC ? BM / 32 : BM / 8;     // float4 (4 k) or scalar loads per thread for dY
  constexpr int B_PER = BVEC ? BNW / 32 : BNW / 8;   // float4 or scalar gathers per thread for X
  static_assert(!BVEC || (AVEC && FAST == 1), "BVEC needs the float4 position mapping and the 32-bit tap mask");

  // row pitch: 32 k = 128 B as fp32 or bf16 hi+lo, 192 B as bf16 hi+mid+lo; +16 B pad (36 r and 52 r mod 64 both walk all
  // 16 four-bank groups over 16 rows: conflict-free b128 fragment reads)
  constexpr int LDP = MATH == 2 ? 52 : (MATH == 3 ? 20 : LDW), NP = math_parts(MATH);   // (3: 64-byte rows of halves + 16 B)
  // ONE LDS buffer for everything but the smallest fp32 tiles: the next tile waits in registers anyway, so the price is a
  // second barrier per k-tile, and the 75-120 KB double-buffered tiles become 38-60 KB -- two (or more) workgroups per CU,
  // so that one's load/split/store phase overlaps the other's MFMAs
  constexpr int NBUF = (MATH == 0 && TM * TN <= 2) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float As[NBUF][BM][LDP];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][BNW][LDP];
  __shared__ int2 Ts[BNW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // Split-major order on XCD-contiguous ids: the tiles of one K-slice run next to each other on one XCD, so
  // its dY / X slabs are fetched from HBM once and shared through that XCD's L2 by all the N-tiles.
  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tilesM * p.tilesN;
  const int split = bid / ntile; bid -= split * ntile;
  const int tileM = bid % p.tilesM, tileN = bid / p.tilesM;

  if (tid < BNW) Ts[tid] = table[tileN * BNW + tid];
  __syncthreads();

  const int kl = tid & 31, g = tid >> 5;             // B (and scalar A): k lane, row group
  const int kq = tid & 7, ga = tid >> 3;             // vector A: k quad, row group
  const unsigned OSP = (unsigned)(p.OD * p.OH * p.OW);
  const int HW = p.H * p.W;
  const bool chkD = p.chk & 1, chkH = p.chk & 2, chkW = p.chk & 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dy), 0, p.dy_bytes, 0x00020000);
  auto ld1 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff) __attribute__((always_inline)) {
    // (MATH 3: the half is not widened; its 16 bits travel in the low half of the "float" until they are packed into LDS)
    if constexpr (H && MATH == 3) return __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)voff, 0, 0));
    else if constexpr (H) return (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)voff, 0, 0));
    else return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0));
  };
  auto ld4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff) __attribute__((always_inline)) {
    if constexpr (H && MATH == 3) {
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t u = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, 0, 0));
      const unsigned ux = u.x, uy = u.y;
      return make_float4(__uint_as_float(ux & 0xffffu), __uint_as_float(ux >> 16), __uint_as_float(uy & 0xffffu), __uint_as_float(uy >> 16));
    } else if constexpr (H) {
      typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      const f16x4_t hv = __builtin_bit_cast(f16x4_t, __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, 0, 0)));
      return make_float4((float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w);
    } else {
      // whole-vector bit_cast: element-wise __builtin_bit_cast miscompiles to a replicated dword load (ROCm 7.2)
      const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0));
      return make_float4(f.x, f.y, f.z, f.w);
    }
  };

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
  float breg[BVEC ? 1 : B_PER];
  float4 bvec[BVEC ? B_PER : 1];
  const int a_rows_left = p.K - (tileM * BM + (AVEC ? ga : g));     // dY row (32*i | 8*i) of this thread is real iff < this
  const unsigned a_row0 = (unsigned)(tileM * BM + (AVEC ? ga : g)) * OSP * ES;
  const unsigned a_step = (AVEC ? 32u : 8u) * OSP * ES;

  // The next tile's fetch is split into pieces that are issued BETWEEN the MFMA groups of the current tile (see
  // conv3d.hip): prep = position arithmetic + window mask, then the dY loads, then the X gathers in chunks.
  unsigned nx_abase = 0, nx_bbase = 0, nx_ilo = 0xffffffffu, nx_ihi = 0xffffffffu;
  bool nx_av = false, nx_kv = false;
  int nx_id0 = 0, nx_ih0 = 0, nx_iw0 = 0;
  auto prep = [&](int kt) __attribute__((always_inline)) {
    unsigned img4 = 0, o4 = 0;
    if (AVEC) {
      const unsigned kp = (unsigned)kt * WBK + kq * 4;          // 4 consecutive positions, same image (OSP % 4 == 0)
      nx_av = kp < p.Ktot;
      const unsigned kc = nx_av ? kp : 0u;
      img4 = gca_fdiv(kc, p.m_osp);
      o4 = kc - img4 * OSP;
      nx_abase = (img4 * (unsigned)p.K * OSP + o4) * ES + a_row0;
    }
    if (BVEC) {                                                 // X rows for the same 4 positions
      const unsigned od = gca_fdiv(o4, p.m_ohw), r = o4 - od * p.m_ohw.d;
      const int id0 = (int)od * p.sd - p.pd;
      nx_bbase = (img4 * p.x_nstride + (unsigned)(id0 * HW) + r) * ES;
      nx_ilo = nx_av ? ~tap_valid_mask<unsigned>(p, id0, 0, 0, chkD, false, false) : 0xffffffffu;
      return;
    }
    // per-lane spatial position of the B column (and of scalar A)
    const unsigned kp = (unsigned)kt * WBK + kl;
    nx_kv = kp < p.Ktot;
    const unsigned kc = nx_kv ? kp : 0u;
    const unsigned img = gca_fdiv(kc, p.m_osp);
    const unsigned o = kc - img * OSP;
    if (!AVEC) { nx_av = nx_kv; nx_abase = (img * (unsigned)p.K * OSP + o) * ES + a_row0; }
    const unsigned od = gca_fdiv(o, p.m_ohw), r = o - od * p.m_ohw.d;
    const unsigned oh = gca_fdiv(r, p.m_ow), ow = r - oh * (unsigned)p.OW;
    nx_id0 = (int)od * p.sd - p.pd; nx_ih0 = (int)oh * p.sh - p.ph; nx_iw0 = (int)ow * p.sw - p.pw;
    nx_bbase = (img * p.x_nstride + (unsigned)(nx_id0 * HW + nx_ih0 * p.W + nx_iw0)) * ES;
    if (FAST == 1) nx_ilo = nx_kv ? ~tap_valid_mask<unsigned>(p, nx_id0, nx_ih0, nx_iw0, chkD, chkH, chkW) : 0xffffffffu;
    if (FAST == 2) {
      const unsigned long long inval = nx_kv ? ~tap_valid_mask<unsigned long long>(p, nx_id0, nx_ih0, nx_iw0, chkD, chkH, chkW) : ~0ull;
      nx_ilo = (unsigned)inval; nx_ihi = (unsigned)(inval >> 32);
    }
  };
  auto issue_a = [&]() __attribute__((always_inline)) {          // dY[img, m, o]
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const unsigned voff = (nx_av && (AVEC ? 32 : 8) * i < a_rows_left) ? nx_abase + (unsigned)i * a_step : 0xffffffffu;
      if (AVEC) avec[AVEC ? i : 0] = ld4(ry, voff);
      else areg[AVEC ? 0 : i] = ld1(ry, voff);
    }
  };
  auto issue_b = [&](int j) __attribute__((always_inline)) {     // X window element(s) of table row n' (j: constant)
    if (BVEC) {
      const int2 e = Ts[ga + 32 * j];                            // x: byte offset, y: tap id in the low bits
      const unsigned voff = (nx_bbase + (unsigned)(e.x >> (H ? 1 : 0))) | (unsigned)__builtin_amdgcn_sbfe((int)nx_ilo, e.y, 1);
      bvec[BVEC ? j : 0] = ld4(rx, voff);
      return;
    }
    const int2 e = Ts[g + 8 * j];
    unsigned inv;
    if (FAST == 1) inv = (unsigned)__builtin_amdgcn_sbfe((int)nx_ilo, e.y, 1);
    else if (FAST == 2) {
      const unsigned hsel = (unsigned)-((e.y >> 5) & 1);
      inv = (unsigned)__builtin_amdgcn_sbfe((int)(nx_ilo ^ ((nx_ilo ^ nx_ihi) & hsel)), e.y, 1);
    } else {
      int off, dd, dh, dw, rvalid;
      decode_row(e, off, dd, dh, dw, rvalid);
      bool ok = nx_kv & (rvalid != 0);
      if (chkD) ok = ok & ((unsigned)(nx_id0 + dd) < (unsigned)p.D);
      if (chkH) ok = ok & ((unsigned)(nx_ih0 + dh) < (unsigned)p.H);
      if (chkW) ok = ok & ((unsigned)(nx_iw0 + dw) < (unsigned)p.W);
      inv = ok ? 0u : 0xffffffffu;
    }
    breg[BVEC ? 0 : j] = ld1(rx, (nx_bbase + (unsigned)(e.x >> (H ? 1 : 0))) | inv);
  };
  constexpr int G = (MATH ? WBK / 16 : WBK / 8) * TM * TN;       // MFMA groups (4 exact / 3 bf16x3 MFMAs each) per tile
  constexpr int GSPAN = G > 2 ? G - 2 : 1;
  constexpr int GCHUNK = (B_PER + GSPAN - 1) / GSPAN;
  auto issue_piece = [&](int gi, int kt) __attribute__((always_inline)) {
    if (gi == 0) { prep(kt); issue_a(); }
    else if (gi <= GSPAN) {
#pragma unroll
      for (int j = 0; j < B_PER; ++j)
        if (j >= (gi - 1) * GCHUNK && j < gi * GCHUNK) issue_b(j);
    }
  };
  auto load_tiles = [&](int kt) __attribute__((always_inline)) {   // everything at once (prologue)
    prep(kt);
    issue_a();
#pragma unroll
    for (int j = 0; j < B_PER; ++j) issue_b(j);
  };
  // MATH >= 1 (bf16x3 / bf16x6, see conv3d.hip): a row of 32 k is four 8-k groups [hi x8 | (mid x8 |) lo x8]; the 16-byte
  // operands of MFMA step t for a lane of half h sit at float index 4*NP*(2t + h) + {0, 4, (8)}.
  auto store_split4 = [&](float* row, int chunk, float4 v) __attribute__((always_inline)) {     // k = 4*chunk .. +3
    float* d = row + (chunk >> 1) * (4 * NP) + (chunk & 1) * 2;
    if constexpr (MATH == 3) {           // both operands are stored halves: pack the raw bits
      *reinterpret_cast<uint2*>(d) = make_uint2(__float_as_uint(v.x) | (__float_as_uint(v.y) << 16),
                                                __float_as_uint(v.z) | (__float_as_uint(v.w) << 16));
    } else if (MATH == 2) {
      uint2 h, m, l;
      split_bf16x3(v.x, v.y, h.x, m.x, l.x);
      split_bf16x3(v.z, v.w, h.y, m.y, l.y);
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + 4) = m;
      *reinterpret_cast<uint2*>(d + 8) = l;
    } else {
      uint2 h, l;
      split_bf16x2(v.x, v.y, h.x, l.x);
      split_bf16x2(v.z, v.w, h.y, l.y);
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + 4) = l;
    }
  };
  auto store_split1 = [&](float* row, int k, float v) __attribute__((always_inline)) {
    unsigned short* d = reinterpret_cast<unsigned short*>(row) + (k >> 3) * (8 * NP) + (k & 7);
    if constexpr (MATH == 3) {
      d[0] = (unsigned short)__float_as_uint(v);
    } else if (MATH == 2) {
      unsigned h, m, l;
      split_bf16x3(v, 0.f, h, m, l);
      d[0] = (unsigned short)h; d[8] = (unsigned short)m; d[16] = (unsigned short)l;
    } else {
      unsigned h, l;
      split_bf16x2(v, 0.f, h, l);
      d[0] = (unsigned short)h; d[8] = (unsigned short)l;
    }
  };
  auto store_tiles = [&](int buf) __attribute__((always_inline)) {
    if constexpr (MATH >= 1) {
      if (AVEC) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) store_split4(&As[buf][ga + 32 * i][0], kq, avec[AVEC ? i : 0]);
      } else {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) store_split1(&As[buf][g + 8 * i][0], kl, areg[AVEC ? 0 : i]);
      }
      if (BVEC) {
#pragma unroll
        for (int j = 0; j < B_PER; ++j) store_split4(&Bs[buf][ga + 32 * j][0], kq, bvec[BVEC ? j : 0]);
      } else {
#pragma unroll
        for (int j = 0; j < B_PER; ++j) store_split1(&Bs[buf][g + 8 * j][0], kl, breg[BVEC ? 0 : j]);
      }
      return;
    }
    if (AVEC) {
#pragma unroll
      for (int i = 0; i < A_PER; ++i) *reinterpret_cast<float4*>(&As[buf][ga + 32 * i][kq * 4]) = avec[i];
    } else {
#pragma unroll
      for (int i = 0; i < A_PER; ++i) As[buf][g + 8 * i][kl] = areg[i];
    }
    if (BVEC) {
#pragma unroll
      for (int j = 0; j < B_PER; ++j) *reinterpret_cast<float4*>(&Bs[buf][ga + 32 * j][kq * 4]) = bvec[j];
    } else {
#pragma unroll
      for (int j = 0; j < B_PER; ++j) Bs[buf][g + 8 * j][kl] = breg[j];
    }
  };

  if (kt0 < kt1) {
    load_tiles(kt0);
    store_tiles(0);
  }
  __syncthreads();
  auto tile = [&](auto BUF, auto MORE, int kt_next) __attribute__((always_inline)) {
    constexpr int buf = decltype(BUF)::value;
    constexpr bool more = decltype(MORE)::value;
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
          if (more) {
            __builtin_amdgcn_sched_barrier(0);
            issue_piece((t * TM + i) * TN + j, kt_next);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  if constexpr (MATH >= 1) {
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = NBUF == 2 ? (kt - kt0) & 1 : 0;
      const bool more = kt + 1 < kt1;
#pragma unroll
      for (int t = 0; t < WBK / 16; ++t) {
        float4 af[TM][NP], bf[TN][NP];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < NP; ++q)
            af[i][q] = *reinterpret_cast<const float4*>(&As[buf][wm * (TM * 32) + i * 32 + ll][4 * NP * (2 * t + lh) + 4 * q]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < NP; ++q)
            bf[j][q] = *reinterpret_cast<const float4*>(&Bs[buf][wn * (TN * 32) + j * 32 + ll][4 * NP * (2 * t + lh) + 4 * q]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (MATH == 3) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, bf[j][0]), acc[i][j], 0, 0, 0);
            } else {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, af[i][0]), xl = __builtin_bit_cast(bf16x8, af[i][NP - 1]);
            const bf16x8 yh = __builtin_bit_cast(bf16x8, bf[j][0]), yl = __builtin_bit_cast(bf16x8, bf[j][NP - 1]);
            if (MATH == 2) {
              const bf16x8 xm = __builtin_bit_cast(bf16x8, af[i][1]), ym = __builtin_bit_cast(bf16x8, bf[j][1]);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, ym, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, yh, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ym, acc[i][j], 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) issue_piece((t * TM + i) * TN + j, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      if (NBUF == 1) __syncthreads();                   // every wave is done reading the only buffer
      if (more) store_tiles(NBUF == 2 ? buf ^ 1 : 0);
      __syncthreads();
    }
  } else if (TM * TN <= 2) {       // small tiles: unrolled by two with compile-time buffers, tail peeled (as in conv3d.hip)
    int kt = kt0;
    for (; kt + 2 < kt1; kt += 2) {
      tile(B0{}, std::true_type{}, kt + 1);
      tile(B1{}, std::true_type{}, kt + 2);
    }
    if (kt + 1 < kt1) {
      tile(B0{}, std::true_type{}, kt + 1);
      tile(B1{}, std::false_type{}, 0);
    } else if (kt < kt1) {
      tile(B0{}, std::false_type{}, 0);
    }
  } else {
  for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = NBUF == 2 ? (kt - kt0) & 1 : 0;
      const bool more = kt + 1 < kt1;
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
            __builtin_amdgcn_sched_barrier(0);
            if (more) issue_piece((t * TM + i) * TN + j, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      if (NBUF == 1) __syncthreads();                   // every wave is done reading the only buffer
      if (more) store_tiles(NBUF == 2 ? buf ^ 1 : 0);
      __syncthreads();
    }

  }

  // ---- epilogue: partial tile -> slab[split][m][n] through a buffer resource (one 32-bit offset per column
  // tile, wave-uniform row part in soffset, all-ones offset = dropped store)
  float* out = slab + (long long)split * p.K * p.Kred;
  const __amdgpu_buffer_rsrc_t ro_ = __builtin_amdgcn_make_buffer_rsrc(out, 0, p.slab_bytes, 0x00020000);
  const int m0 = tileM * BM + wm * (TM * 32) + 4 * lh;
  const int rows_left = p.K - m0;
  const unsigned rowb = (unsigned)p.Kred * 4u;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = tileN * BNW + wn * (TN * 32) + j * 32 + ll;
    const unsigned vb = n < p.Kred ? ((unsigned)m0 * (unsigned)p.Kred + (unsigned)n) * 4u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
        const unsigned vo = ro < rows_left ? vb : 0xffffffffu;
        const float v = acc[i][j][r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro_, (int)vo, (int)((unsigned)ro * rowb), 0);
      }
  }
}

// dw[i] (+)= sum_s slab[s][i]   (fixed order: deterministic).  A block owns 64 consecutive outputs; its four waves
// each sum every fourth slab (short dependent chains, 4x the workgroups of a thread-per-output layout) and the four
// partial sums are combined in a fixed order through LDS.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                            long long n, int splits, int accumulate) {
  __shared__ float sh[4][64];
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + x;
  float s = 0.f;
  if (i < n) {
    int k = y;
    for (; k + 60 < splits; k += 64) {              // sixteen loads in flight per chain (the stem's wgrad has 200-500 slabs:
      float l[16];                                  // four at a time was 30+ exposed round trips), summed in the fixed order
#pragma unroll
      for (int u = 0; u < 16; ++u) l[u] = slab[(long long)(k + 4 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += l[u];
    }
    for (; k + 12 < splits; k += 16) {              // four loads in flight per chain
      const float l0 = slab[(long long)k * n + i], l1 = slab[(long long)(k + 4) * n + i];
      const float l2 = slab[(long long)(k + 8) * n + i], l3 = slab[(long long)(k + 12) * n + i];
      s += l0; s += l1; s += l2; s += l3;
    }
    for (; k < splits; k += 4) s += slab[(long long)k * n + i];
  }
  sh[y][x] = s;
  __syncthreads();
  if (y == 0 && i < n) {
    float t = (sh[0][x] + sh[1][x]) + (sh[2][x] + sh[3][x]);
    if (accumulate) t += dw[i];
    dw[i] = t;
  }
}

// The same reduction for n % 4 == 0 (every layer with an output-channel count that is a multiple of four): a thread owns
// FOUR consecutive outputs and reads the slabs with 16-byte loads; per output the summation order is exactly that of
// splitk_reduce_kernel (slabs y, y+4, ... per wave, then the four partial sums), so both give the same bits.
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float4* __restrict__ slab, float4* __restrict__ dw,
                                                             long long n4, int splits, int accumulate) {
  __shared__ float4 sh[4][64];
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + x;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    int k = y;
    for (; k + 60 < splits; k += 64) {
      float4 l[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) l[u] = slab[(long long)(k + 4 * u) * n4 + i];
#pragma unroll
      for (int u = 0; u < 16; ++u) { s.x += l[u].x; s.y += l[u].y; s.z += l[u].z; s.w += l[u].w; }
    }
    for (; k + 12 < splits; k += 16) {
      const float4 l0 = slab[(long long)k * n4 + i], l1 = slab[(long long)(k + 4) * n4 + i];
      const float4 l2 = slab[(long long)(k + 8) * n4 + i], l3 = slab[(long long)(k + 12) * n4 + i];
      s.x += l0.x; s.y += l0.y; s.z += l0.z; s.w += l0.w;
      s.x += l1.x; s.y += l1.y; s.z += l1.z; s.w += l1.w;
      s.x += l2.x; s.y += l2.y; s.z += l2.z; s.w += l2.w;
      s.x += l3.x; s.y += l3.y; s.z += l3.z; s.w += l3.w;
    }
    for (; k < splits; k += 4) {
      const float4 l = slab[(long long)k * n4 + i];
      s.x += l.x; s.y += l.y; s.z += l.z; s.w += l.w;
    }
  }
  sh[y][x] = s;
  __syncthreads();
  if (y == 0 && i < n4) {
    const float4 a = sh[0][x], b = sh[1][x], c = sh[2][x], d = sh[3][x];
    float4 t = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    if (accumulate) { const float4 o = dw[i]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    dw[i] = t;
  }
}

// The same two reductions for MANY weight gradients in one launch (gca_splitk_reduce_batched): a block looks its job up by
// block range and then does exactly what a block of splitk_reduce4_kernel / splitk_reduce_kernel does for that job -- same
// slabs per wave, same fold order, hence the same bits as the per-layer launches.  39 launches of 8-12 us each (all latency)
// become one.
struct ReduceJob { const float* slabs; float* dw; long long n; int splits; int accumulate; int first_block; int nblocks; };
static_assert(sizeof(ReduceJob) == sizeof(gca_reduce_job), "gca_hip.h: gca_reduce_job layout");
__global__ __launch_bounds__(256) void splitk_reduce_batched_kernel(const ReduceJob* __restrict__ jobs, int njobs) {
  __shared__ float4 sh[4][64];
  const int bid = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= bid) lo = mid; else hi = mid - 1;
  }
  const ReduceJob j = jobs[lo];
  const int blk = bid - j.first_block;
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int splits = j.splits;
  if (j.n % 4 == 0) {                                   // (host: vector form only when slabs / dw are 16-byte aligned)
    const long long n4 = j.n >> 2, i = (long long)blk * 64 + x;
    const float4* slab = reinterpret_cast<const float4*>(j.slabs);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      int k = y;
      for (; k + 60 < splits; k += 64) {
        float4 l[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) l[u] = slab[(long long)(k + 4 * u) * n4 + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) { s.x += l[u].x; s.y += l[u].y; s.z += l[u].z; s.w += l[u].w; }
      }
      for (; k + 12 < splits; k += 16) {
        const float4 l0 = slab[(long long)k * n4 + i], l1 = slab[(long long)(k + 4) * n4 + i];
        const float4 l2 = slab[(long long)(k + 8) * n4 + i], l3 = slab[(long long)(k + 12) * n4 + i];
        s.x += l0.x; s.y += l0.y; s.z += l0.z; s.w += l0.w;
        s.x += l1.x; s.y += l1.y; s.z += l1.z; s.w += l1.w;
        s.x += l2.x; s.y += l2.y; s.z += l2.z; s.w += l2.w;
        s.x += l3.x; s.y += l3.y; s.z += l3.z; s.w += l3.w;
      }
      for (; k < splits; k += 4) {
        const float4 l = slab[(long long)k * n4 + i];
        s.x += l.x; s.y += l.y; s.z += l.z; s.w += l.w;
      }
    }
    sh[y][x] = s;
    __syncthreads();
    if (y == 0 && i < n4) {
      const float4 a = sh[0][x], b = sh[1][x], c = sh[2][x], d = sh[3][x];
      float4 t = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
      float4* dw = reinterpret_cast<float4*>(j.dw);
      if (j.accumulate) { const float4 o = dw[i]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
      dw[i] = t;
    }
  } else {
    float* shs = reinterpret_cast<float*>(&sh[0][0]);   // [4][64] floats
    const long long i = (long long)blk * 64 + x;
    float s = 0.f;
    if (i < j.n) {
      int k = y;
      for (; k + 60 < splits; k += 64) {
        float l[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) l[u] = j.slabs[(long long)(k + 4 * u) * j.n + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += l[u];
      }
      for (; k + 12 < splits; k += 16) {
        const float l0 = j.slabs[(long long)k * j.n + i], l1 = j.slabs[(long long)(k + 4) * j.n + i];
        const float l2 = j.slabs[(long long)(k + 8) * j.n + i], l3 = j.slabs[(long long)(k + 12) * j.n + i];
        s += l0; s += l1; s += l2; s += l3;
      }
      for (; k < splits; k += 4) s += j.slabs[(long long)k * j.n + i];
    }
    shs[y * 64 + x] = s;
    __syncthreads();
    if (y == 0 && i < j.n) {
      float t = (shs[x] + shs[64 + x]) + (shs[128 + x] + shs[192 + x]);
      if (j.accumulate) t += j.dw[i];
      j.dw[i] = t;
    }
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

// Tile shapes (rows = output channels, cols = C*taps).  Index = gca_conv_geom.tune_wgrad_tile (0 = heuristic).
struct WgradShape { int bm, bn, wm, wn, tm, tn; };
constexpr int N_WGRAD_SHAPES = 10;
const WgradShape WGRAD_SHAPES[N_WGRAD_SHAPES + 1] = {
    {0, 0, 0, 0, 0, 0},
    {64, 64, 2, 2, 1, 1},  {64, 128, 2, 2, 1, 2},  {128, 64, 2, 2, 2, 1},  {128, 128, 2, 2, 2, 2},   // 1..4
    {96, 128, 1, 4, 3, 1}, {160, 128, 1, 4, 5, 1}, {128, 96, 4, 1, 1, 3},  {128, 160, 4, 1, 1, 5},   // 5..8
    {64, 192, 2, 2, 1, 3}, {192, 64, 2, 2, 3, 1}};                                                    // 9..10

inline int wgrad_fast(const gca_conv_geom* g) { const int t = taps(g); return t <= 31 ? 1 : (t <= 62 ? 2 : 0); }
inline bool wgrad_shape_ok(int idx, int fast) {
  if (idx < 1 || idx > N_WGRAD_SHAPES) return false;
  if (fast == 1) return true;
  if (fast == 2) return idx == 3 || idx == 4 || idx == 7 || idx == 8;       // instantiated for the 7x7 stem family
  return idx <= 4;
}

struct WgradPlan { WgradParams p; int shape; bool avec; int fast; bool bvec; int ts; };   // ts: 11 / 12 = streaming temporal kernel

void wgrad_plan(const gca_conv_geom* g, WgradPlan& pl) {
  WgradParams& p = pl.p;
  p.C = g->C; p.D = g->D; p.H = g->H; p.W = g->W; p.K = g->K;
  p.OD = g->OD; p.OH = g->OH; p.OW = g->OW;
  p.kd = g->kd; p.kh = g->kh; p.kw = g->kw;
  p.sd = g->sd; p.sh = g->sh; p.sw = g->sw; p.pd = g->pd; p.ph = g->ph; p.pw = g->pw;
  p.Kred = g->C * taps(g);
  const long long osp = (long long)g->OD * g->OH * g->OW;
  p.Ktot = (unsigned)((long long)g->N * osp);
  const long long cdhw = (long long)g->C * g->D * g->H * g->W;
  p.x_nstride = (unsigned)(g->x_batch_stride ? g->x_batch_stride : cdhw);
  const long long es = g->act_f16 ? 2 : 4;                       // fp16 storage of x and dy
  const long long xb = (long long)g->N * p.x_nstride * es, yb = (long long)g->N * g->K * osp * es;
  const long long sb = (long long)g->K * p.Kred * 4;
  p.x_bytes = xb > 0xfffff000LL ? 0xfffff000u : (unsigned)xb;
  p.dy_bytes = yb > 0xfffff000LL ? 0xfffff000u : (unsigned)yb;
  p.slab_bytes = sb > 0xfffff000LL ? 0xfffff000u : (unsigned)sb;
  p.m_osp = make_magic((unsigned)osp);
  p.m_ohw = make_magic((unsigned)(g->OH * g->OW));
  p.m_ow = make_magic((unsigned)g->OW);
  pl.avec = osp % 4 == 0;
  pl.fast = wgrad_fast(g);
  pl.bvec = pl.avec && pl.fast == 1 && g->kh == 1 && g->kw == 1 && g->sh == 1 && g->sw == 1 && g->ph == 0 && g->pw == 0 &&
            (g->OH * g->OW) % 4 == 0 && cdhw % 4 == 0 && p.x_nstride % 4 == 0;
  p.kt_total = (int)gca_ceil_div((long long)p.Ktot, WBK);
  p.math = resolve_math(g->tune_wgrad_math, g->act_f16);
  p.half = p.math == 3;
  pl.ts = 0;
  if (g->tune_wgrad_tile == 14 && wgrad_stem_ok(g, p.math)) {
    // conv3d_wgrad_stem.hip: whole dW per workgroup group, split over (clip, od) units
    pl.ts = 14;
    pl.shape = 0;
    p.tilesM = 1;
    p.tilesN = 1;
    p.splits = wgrad_stem_splits(g, g->tune_wgrad_splits);
    p.kt_per_split = 0;
    p.chk = 0;
    return;
  }
  if (g->tune_wgrad_tile >= 11 && g->tune_wgrad_tile <= 13 && wgrad_ts_ok(g, g->tune_wgrad_tile, p.math)) {
    // conv3d_wgrad_ts.hip: its own tiling (32*TM x 32 x all taps per wave) and split (over clip x position-chunk units)
    pl.ts = g->tune_wgrad_tile;
    pl.shape = 0;
    const int tm = pl.ts == 12 ? 2 : 1;
    p.tilesM = (int)gca_ceil_div(g->K, 32 * tm);
    p.tilesN = (int)gca_ceil_div(g->C, 32);
    p.splits = wgrad_ts_splits(g, pl.ts, g->tune_wgrad_splits);
    p.kt_per_split = 0;
    p.chk = 0;
    return;
  }
  // tile shape: least padded MFMA work, discounted for how well a shape amortises its gathers
  int best = 0;
  if (wgrad_shape_ok(g->tune_wgrad_tile, pl.fast)) best = g->tune_wgrad_tile;
  else {
    double best_cost = 1e300;
    for (int i = 1; i <= N_WGRAD_SHAPES; ++i) {
      if (!wgrad_shape_ok(i, pl.fast)) continue;
      const WgradShape& sh = WGRAD_SHAPES[i];
      const double padded = (double)gca_round_up(g->K, sh.bm) * (double)gca_round_up(p.Kred, sh.bn);
      const double eff = (sh.bm * sh.bn >= 128 * 96 ? 1.0 : (sh.bm * sh.bn >= 64 * 128 ? 0.92 : 0.8)) *
                         ((sh.bm + sh.bn) > 256 ? 0.9 : 1.0);            // > 64 KiB of LDS: one workgroup per CU
      const double cost = padded / eff;
      if (cost < best_cost) { best_cost = cost; best = i; }
    }
  }
  pl.shape = best;
  const WgradShape& sh = WGRAD_SHAPES[best];
  p.tilesM = (int)gca_ceil_div(g->K, sh.bm);
  p.tilesN = (int)gca_ceil_div(p.Kred, sh.bn);
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

template <int WM, int WN, int TM, int TN, int FAST, int MATH>
void launch_wm(int avec, dim3 grid, hipStream_t st, const float* x, const float* dy, const int2* t, float* slab,
               const WgradParams& p) {      // avec: 0 scalar, 1 float4 dY, 2 float4 dY and X
  if (avec == 2 && FAST == 1)
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, true, FAST == 1 ? 1 : 1, true, MATH>), grid, dim3(256), 0, st, x, dy, t, slab, p);
  else if (avec) hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, true, FAST, false, MATH>), grid, dim3(256), 0, st, x, dy, t, slab, p);
  else hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, false, FAST, false, MATH>), grid, dim3(256), 0, st, x, dy, t, slab, p);
}

template <int WM, int WN, int TM, int TN, int FAST>
void launch_w(int avec, dim3 grid, hipStream_t st, const float* x, const float* dy, const int2* t, float* slab,
              const WgradParams& p) {
  if (p.half) {
    if (avec == 2 && FAST == 1)
      hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, true, 1, true, 3, true>), grid, dim3(256), 0, st, x, dy, t, slab, p);
    else if (avec) hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, true, FAST, false, 3, true>), grid, dim3(256), 0, st, x, dy, t, slab, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, false, FAST, false, 3, true>), grid, dim3(256), 0, st, x, dy, t, slab, p);
  } else if (p.math == 1) launch_wm<WM, WN, TM, TN, FAST, 1>(avec, grid, st, x, dy, t, slab, p);
  else if (p.math == 2) launch_wm<WM, WN, TM, TN, FAST, 2>(avec, grid, st, x, dy, t, slab, p);
  else launch_wm<WM, WN, TM, TN, FAST, 0>(avec, grid, st, x, dy, t, slab, p);
}

template <int FAST>
int launch_shape(int shape, int avec, dim3 grid, hipStream_t st, const float* x, const float* dy, const int2* t,
                 float* slab, const WgradParams& p) {
  switch (shape) {
    case 1: if (FAST != 2) { launch_w<2, 2, 1, 1, FAST == 2 ? 1 : FAST>(avec, grid, st, x, dy, t, slab, p); return 0; } break;
    case 2: if (FAST != 2) { launch_w<2, 2, 1, 2, FAST == 2 ? 1 : FAST>(avec, grid, st, x, dy, t, slab, p); return 0; } break;
    case 3: launch_w<2, 2, 2, 1, FAST>(avec, grid, st, x, dy, t, slab, p); return 0;
    case 4: launch_w<2, 2, 2, 2, FAST>(avec, grid, st, x, dy, t, slab, p); return 0;
    case 5: if (FAST == 1) { launch_w<1, 4, 3, 1, 1>(avec, grid, st, x, dy, t, slab, p); return 0; } break;
    case 6: if (FAST == 1) { launch_w<1, 4, 5, 1, 1>(avec, grid, st, x, dy, t, slab, p); return 0; } break;
    case 7: if (FAST != 0) { launch_w<4, 1, 1, 3, FAST == 0 ? 1 : FAST>(avec, grid, st, x, dy, t, slab, p); return 0; } break;
    case 8: if (FAST != 0) { launch_w<4, 1, 1, 5, FAST == 0 ? 1 : FAST>(avec, grid, st, x, dy, t, slab, p); return 0; } break;
    case 9: if (FAST == 1) { launch_w<2, 2, 1, 3, 1>(avec, grid, st, x, dy, t, slab, p); return 0; } break;
    case 10: if (FAST == 1) { launch_w<2, 2, 3, 1, 1>(avec, grid, st, x, dy, t, slab, p); return 0; } break;
  }
  return GCA_EINVAL;
}

}  // namespace

extern "C" {

int64_t gca_conv_wgrad_ws_bytes(const gca_conv_geom* g) {
  if (!geom_ok(g)) return GCA_EINVAL;
  WgradPlan pl{};
  wgrad_plan(g, pl);
  return (int64_t)pl.p.splits * g->K * pl.p.Kred * (int64_t)sizeof(float);
}

/* out4 = {tile rows, tile cols, split-K factor, shape index | float4-dY<<8 | tap-mask kind<<9} */
int gca_conv_wgrad_cfg(const gca_conv_geom* g, int32_t* out4) {
  if (!geom_ok(g) || !out4) return GCA_EINVAL;
  WgradPlan pl{};
  wgrad_plan(g, pl);
  if (pl.ts) {
    out4[0] = pl.ts == 12 ? 64 : 32; out4[1] = 32 * taps(g); out4[2] = pl.p.splits;
    out4[3] = pl.ts | (pl.p.math << 12);
    return GCA_OK;
  }
  out4[0] = WGRAD_SHAPES[pl.shape].bm; out4[1] = WGRAD_SHAPES[pl.shape].bn; out4[2] = pl.p.splits;
  out4[3] = pl.shape | (pl.avec << 8) | (pl.fast << 9) | (pl.bvec << 11) | (pl.p.math << 12);
  return GCA_OK;
}

// the split-K kernel of `g` into `slab` (gca_conv_wgrad_ws_bytes); -> splits (> 0) or a negative status
static int wgrad_partial_launch(const gca_conv_geom* g, const void* x_, const void* dy_, const int32_t* table, void* ws,
                                hipStream_t st, const float* in_scale = nullptr, const float* in_shift = nullptr) {
  const float* x = reinterpret_cast<const float*>(x_);      // opaque to the host side: the kernels index in bytes
  const float* dy = reinterpret_cast<const float*>(dy_);
  if (!geom_ok(g) || !x || !dy || !table || !ws) return GCA_EINVAL;
  WgradPlan pl{};
  wgrad_plan(g, pl);
  const WgradParams& p = pl.p;
  const long long nblk = (long long)p.tilesM * p.tilesN * p.splits;
  if (nblk > 0x7fffffffLL) return GCA_EINVAL;
  const int2* t = reinterpret_cast<const int2*>(table);
  float* slab = reinterpret_cast<float*>(ws);
  dim3 grid((unsigned)nblk);
  int rc;
  if (pl.ts == 14) {
    if (in_scale || ((uintptr_t)x % 4) || ((uintptr_t)dy % 16)) return GCA_EINVAL;
    rc = wgrad_stem_launch(g, p.math, p.splits, x_, dy_, slab, st);
    return rc ? rc : p.splits;
  }
  if (pl.ts) {
    // 16-byte DMA pieces; rows of the (1,3,3) kernel start at arbitrary columns anyway, so only dword alignment matters there
    if (pl.ts != 13 && (((uintptr_t)x % 16) || ((uintptr_t)dy % 16))) return GCA_EINVAL;
    if (((uintptr_t)x % 4) || ((uintptr_t)dy % 4)) return GCA_EINVAL;
    rc = wgrad_ts_launch(g, pl.ts, p.math, p.splits, x, dy, slab, st, in_scale, in_shift);
    return rc ? rc : p.splits;
  }
  if (in_scale) return GCA_EINVAL;                       // only the streaming kernels transform x on the fly (gca_conv_xf_ok)
  const int av = !pl.avec || ((uintptr_t)dy % 16) ? 0 : (pl.bvec && ((uintptr_t)x % 16) == 0 ? 2 : 1);
  if (pl.fast == 1) rc = launch_shape<1>(pl.shape, av, grid, st, x, dy, t, slab, p);
  else if (pl.fast == 2) rc = launch_shape<2>(pl.shape, av, grid, st, x, dy, t, slab, p);
  else rc = launch_shape<0>(pl.shape, av, grid, st, x, dy, t, slab, p);
  if (rc) return rc;
  rc = gca_launch_status();
  return rc ? rc : p.splits;
}

int gca_conv_wgrad(const gca_conv_geom* g, const void* x_, const void* dy_, const int32_t* table,
                   float* dw, int accumulate, void* ws, void* stream) {
  return gca_conv_wgrad_xf(g, x_, nullptr, nullptr, dy_, table, dw, accumulate, ws, stream);
}

int gca_conv_wgrad_xf(const gca_conv_geom* g, const void* x_, const float* in_scale, const float* in_shift, const void* dy_,
                      const int32_t* table, float* dw, int accumulate, void* ws, void* stream) {
  if (!dw || ((in_scale == nullptr) != (in_shift == nullptr))) return GCA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int splits = wgrad_partial_launch(g, x_, dy_, table, ws, st, in_scale, in_shift);
  if (splits < 0) return splits;
  float* slab = reinterpret_cast<float*>(ws);
  const long long n = (long long)g->K * g->C * taps(g);
  if (n % 4 == 0 && ((uintptr_t)slab % 16) == 0 && ((uintptr_t)dw % 16) == 0)
    hipLaunchKernelGGL(splitk_reduce4_kernel, dim3((unsigned)gca_ceil_div(n / 4, 64)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(slab), reinterpret_cast<float4*>(dw), n / 4, splits, accumulate ? 1 : 0);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)gca_ceil_div(n, 64)), dim3(256), 0, st, slab, dw, n,
                       splits, accumulate ? 1 : 0);
  return gca_launch_status();
}

int gca_conv_wgrad_partial(const gca_conv_geom* g, const void* x, const float* in_scale, const float* in_shift, const void* dy,
                           const int32_t* table, void* slabs, int32_t* out_splits, void* stream) {
  if (!out_splits || ((in_scale == nullptr) != (in_shift == nullptr))) return GCA_EINVAL;
  const int splits = wgrad_partial_launch(g, x, dy, table, slabs, (hipStream_t)stream, in_scale, in_shift);
  if (splits < 0) return splits;
  *out_splits = splits;
  return GCA_OK;
}

int64_t gca_reduce_jobs_finalize_host(gca_reduce_job* jobs, int64_t njobs) {
  if (!jobs || njobs <= 0) return GCA_EINVAL;
  long long first = 0;
  for (int64_t i = 0; i < njobs; ++i) {
    gca_reduce_job& j = jobs[i];
    if (!j.slabs || !j.dw || j.n <= 0 || j.splits <= 0) return GCA_EINVAL;
    // the vector form needs 16-byte aligned slabs / dw; otherwise n % 4 != 0 must hold for the kernel to take the scalar form
    if (j.n % 4 == 0 && (((uintptr_t)j.slabs % 16) || ((uintptr_t)j.dw % 16))) return GCA_EINVAL;
    const long long nb = j.n % 4 == 0 ? gca_ceil_div(j.n / 4, 64) : gca_ceil_div(j.n, 64);
    if (first + nb > 0x7fffffffLL) return GCA_EINVAL;
    j.first_block = (int32_t)first;
    j.nblocks = (int32_t)nb;
    first += nb;
  }
  return first;
}

int gca_splitk_reduce_batched(const gca_reduce_job* jobs_dev, int64_t njobs, int64_t total_blocks, void* stream) {
  if (!jobs_dev || njobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) return GCA_EINVAL;
  hipLaunchKernelGGL(splitk_reduce_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const ReduceJob*>(jobs_dev), (int)njobs);
  return gca_launch_status();
}

int gca_bias_grad(const float* dy, int64_t N, int64_t K, int64_t SP, float* db, int accumulate, void* stream) {
  if (!dy || !db || N <= 0 || K <= 0 || SP <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, dy, (long long)N,
                     (long long)K, (long long)SP, db, accumulate ? 1 : 0);
  return gca_launch_status();
}

}  // extern "C"
