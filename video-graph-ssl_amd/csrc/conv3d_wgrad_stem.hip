// Weight gradient of the STEM convolutions: <= 4 input channels, stride 2 along H and W, unit stride along D, up to 7 x 7 x 7
// taps (resnet2p1d.py:162-168 and s3d_1.py stem (1,7,7); resnet.py:120-126 (7,7,7)).
//
//   dW[k, c, a, b, e] = sum_{n, od, oh, ow} dY[n, k, od, oh, ow] * X[n, c, od + a - pd, 2 oh + b - ph, 2 ow + e - pw]
//
// As a GEMM this is M = K output channels (64 / 110) x N = C * taps columns (147 / 1029) with the REDUCTION over 1.6 - 6.4
// million output positions.  conv3d_wgrad.hip gathers every x element of the im2col operand on its own (a 2- or 4-byte
// load per element per tap, 64 x 64 tiles when the tap mask does not fit 64 bits): 97 TF/s (bf16x6) and 170 TF/s (fp16) --
// 0.54 ms of the configs[1] step, 5.0 of the 46 ms of configs[4].  Here
//
//   * the MFMA reduction index is 16 consecutive ow of one output row; A fragment = dY[k, ., od, oh, ow..] (lane = channel,
//     8 consecutive positions), B fragment of column (a, b, c, e) = X[c, od + a - pd, 2 oh + b - ph, 2 (ow..) + e - pw];
//   * a stride-2 window is contiguous in the PARITY phases of the row: P_par[i] = x[2 i + par], and column e reads
//     P_par[ow + s] with u = e - pw, par = u & 1, s = (u - par) / 2 in {-2 .. 1}.  Every x row a workgroup needs is staged in
//     LDS ONCE as four arrays -- both phases, each also shifted by one element -- so that every (column, ow) window starts on
//     a 4-byte boundary: A_{par,cp}[j] = P_par[j - 2 + cp], column e reads 8 elements from j = ow + s + 2 - cp of array
//     (par, cp = s & 1), two ds_read2_b32.  In the split-product modes the bf16 hi / mid / lo parts are separate arrays: an x
//     element is split once per staging, not once per tap;
//   * rows roll: output row oh needs input rows 2 oh - ph .. 2 oh - ph + KH - 1, two new ones per oh, kept in a ring of 9
//     row slots per (tap plane, channel); a workgroup walks "units" (clip, od, chunk of output rows) row by row;
//   * dY travels global -> LDS by LDS-DMA in lane = (channel, 8-position half) order, i.e. already as MFMA fragments: a lane
//     reads back the 16 bytes it wrote (fp32: two pieces, split to bf16 parts in registers); one stage (SS steps of 16
//     positions) is in flight behind the one being multiplied;
//   * a tap plane has 5 column tiles of 32 (b, c, e) columns; a wave owns a 2 x 2 block (two row tiles of 32 output channels x
//     two column tiles) plus one tile of the fifth column: 5 accumulator tiles on 2 A and 3 B fragments per step (one row
//     tile x 5 column tiles reads 5 B fragments per step: the kernel was then bound by LDS reads, not by the matrix pipe).
//     The four waves = 2 column halves x (2 row-tile pairs | 2 tap planes of the group | 2 interleaved shares of the steps);
//     tap planes in the zero padding are skipped;
//   * one fp32 slab per workgroup "split" in dW's own layout, folded by the batched split-K reduction like every other weight
//     gradient (fixed order: deterministic).
#include "conv_common.h"

using namespace gca_conv;

namespace {

constexpr int RING = 9;          // row slots per (plane, channel): KH <= 7 live rows + the 2 being staged for the next oh
constexpr int XT = 3;            // staging tasks per thread and round (a round = 2 input rows of every plane and channel)
constexpr int CT = 5;            // column tiles of 32 per tap plane: KH * C * KW <= 160
constexpr int NB = 3;            // B fragments of a wave: its two block columns and the fifth column
constexpr int NT = 5;            // accumulator tiles of a wave: (r0, c0) (r0, c0+1) (r0+1, c0) (r0+1, c0+1) (r0 + half, 4)

struct StwParams {
  int K, C, D, H, W, OD, OH, OW, KD, KH, KW, pd, ph, pw;
  int nrt;                       // row tiles of 32 output channels per workgroup: 2 or 4
  int g;                         // tap planes per workgroup: 2 when KD > 1 and nrt = 2, else 1
  int wk;                        // waves sharing a tile set, interleaved over the steps: 2 when KD = 1 and nrt = 2, else 1
  int ngroups;                   // ceil(KD / g)
  int units, ups, splits;        // unit = (clip, od, chunk of ohl output rows); units per split
  int ohc, ohl;                  // row chunks per (clip, od), rows per chunk
  int steps;                     // ceil(OW / 16)
  int SS, nstg;                  // steps per dY stage, stages per output row
  int ncols;                     // KH * C * KW
  int LQ;                        // dwords of one array: 8 * steps + 1 (odd: consecutive arrays walk all banks)
  int ARRB, RS, PLS, ZB;         // bytes: one array, one row slot (C * NP * 4 arrays), one plane (RING slots), zero region
  int DYB;                       // bytes of one dY stage buffer
  int ntask;                     // staging tasks of a round: 2 * C * g * LQ
  unsigned x_nstride, dy_nstride;   // elements between clips
  unsigned osp;                  // OD * OH * OW
  unsigned x_bytes, dy_bytes, slab_bytes;
  long long nW;                  // elements of dW (slab stride)
};

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// (see conv3d_wgrad_ts.hip: the builtin form of the LDS-DMA makes hipcc drain vmcnt before every later LDS read)
__device__ __forceinline__ void dma16(const i32x4 rs, unsigned voff, unsigned soff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, %1 offen lds"
               : : "v"(voff), "s"(soff), "s"(lds), "s"(rs) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}

template <int MATH>      // 1: bf16x3 (hi, lo), 2: bf16x6 (hi, mid, lo) on fp32 tensors; 3: fp16 tensors, one f16 product
__global__ __launch_bounds__(256) void conv_wgrad_stem_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                              float* __restrict__ slab, const StwParams p) {
  constexpr bool F16 = MATH == 3;
  constexpr int NP = MATH == 2 ? 3 : (MATH == 1 ? 2 : 1);
  constexpr unsigned ES = F16 ? 2u : 4u;
  constexpr int NQ = F16 ? 1 : 2;                      // 1 KB DMA pieces per (step, row tile): 32 rows x 2 halves x 16 bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const XZ = smem;                      // zeros: what the dead columns (>= ncols) of the last tile read
  unsigned char* const XR = smem + p.ZB;               // [g][RING][C][NP][4 arrays][ARRB]
  unsigned char* const DYT = XR + p.g * p.PLS;         // [2][SS][nrt][NQ][1 KB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, ll = lane & 31;

  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);      // the groups of one split sit on one XCD: dY comes through one L2
  const int split = bid / p.ngroups, grp = bid - split * p.ngroups;
  const int u0 = split * p.ups;
  int u1 = u0 + p.ups; if (u1 > p.units) u1 = p.units;

  const int half = wave >> 1, sel = wave & 1;          // column half; row-tile pair | tap plane of the group | share of the steps
  const int r0 = p.nrt == 4 ? 2 * sel : 0;             // first row tile of the wave's pair
  const int kdl = p.g > 1 ? sel : 0;                   // tap plane of this wave inside the group
  const int ksp = p.wk > 1 ? sel : 0;                  // share of the steps
  const int kd = grp * p.g + kdl;

  const i32x4 rx = make_rsrc(x, p.x_bytes), ry = make_rsrc(dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t bx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, p.x_bytes, 0x00020000);
  (void)rx;
  const unsigned dyt_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(lds_void*)DYT);

  // ---- columns of this lane: col = ct * 32 + ll = (b * C + c) * KW + e
  unsigned cB[NB];             // LDS byte offset inside the ring without the row slot (plane, channel, array, shift)
  int rB[NB];                  // ring slot of the lane's input row for the current oh
  bool dead[NB];
  int dwoff[NB];               // element offset of the column inside dW[k] (c, kd, b, e)
  int colb[NB];                // tap row b of the column
#pragma unroll
  for (int ct = 0; ct < NB; ++ct) {
    const int col = (ct < 2 ? 2 * half + ct : 4) * 32 + ll;
    dead[ct] = col >= p.ncols;
    const int cc = dead[ct] ? 0 : col;
    const int b = cc / (p.C * p.KW), rem = cc - b * (p.C * p.KW);
    const int c = rem / p.KW, e = rem - c * p.KW;
    const int u = e - p.pw, par = u & 1, s = (u - par) >> 1, cp = s & 1;
    cB[ct] = (unsigned)(kdl * p.PLS + c * (NP * 4 * p.ARRB) + (par * 2 + cp) * p.ARRB + (s + 2 - cp) * 2);
    rB[ct] = (b - p.ph + 2 * RING) % RING;
    colb[ct] = b;
    dwoff[ct] = ((c * p.KD + kd) * p.KH + b) * p.KW + e;
  }

  // ---- staging tasks of this thread: (row of the round, plane, channel, dword q of the arrays); constant for the kernel
  unsigned t_dst[XT];          // ring byte offset without the row slot
  unsigned t_src[XT];          // element offset inside the clip without (od, round): c * DHW + pl * HW + rowsel * W
  int t_m[XT];                 // first source pair m = 2 q - 2 (pairs m, m + 1, m + 2)
  int t_pl[XT], t_row[XT];     // plane, row of the round; plane < 0: no task
#pragma unroll
  for (int j = 0; j < XT; ++j) {
    const int task = tid + 256 * j;
    const int rr = task / p.LQ, q = task - rr * p.LQ;
    const int rowsel = rr & 1, cpl = rr >> 1;
    const int c = cpl % p.C, pl = cpl / p.C;
    const bool ok = task < p.ntask;
    t_pl[j] = ok ? pl : -1;
    t_row[j] = rowsel;
    t_m[j] = 2 * q - 2;
    t_dst[j] = (unsigned)(pl * p.PLS + c * (NP * 4 * p.ARRB) + q * 4);
    t_src[j] = (unsigned)((c * p.D + pl) * (p.H * p.W) + rowsel * p.W);
  }

  // ---- dY pieces this wave issues per stage: piece = (step of the stage, row tile, q), lane = (row, half)
  const int npc_full = p.SS * p.nrt * NQ;
  // lane part of the source address of piece (.., rtile, q): row k = rtile * 32 + ll, positions 8 lh + 4 q (fp32) | 8 lh
  auto dy_voff = [&](int rtile, int q, int step) __attribute__((always_inline)) -> unsigned {
    const int k = rtile * 32 + ll;
    const int pos = step * 16 + 8 * lh + (F16 ? 0 : 4 * q);
    const bool ok = k < p.K && pos < p.OW && step < p.steps;
    return ok ? ((unsigned)k * p.osp + (unsigned)(8 * lh + (F16 ? 0 : 4 * q))) * ES : 0xffffffffu;
  };

  f32x16 acc[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  unsigned xr[XT][3][F16 ? 1 : 2];       // the three source pairs of a task (fp16: one dword = both phases)

  for (int u = u0; u < u1; ++u) {
    const int ohc = u % p.ohc, nod = u / p.ohc;
    const int n = nod / p.OD, od = nod - n * p.OD;
    const int oh0 = ohc * p.ohl;
    int oh1 = oh0 + p.ohl; if (oh1 > p.OH) oh1 = p.OH;
    const int total = (oh1 - oh0) * p.nstg;
    const int d0 = od + grp * p.g - p.pd;                                  // source plane of local tap plane 0
    const bool wave_on = kd < p.KD && (unsigned)(d0 + kdl) < (unsigned)p.D;
    const unsigned xbase = (unsigned)n * p.x_nstride + (unsigned)(d0 * (p.H * p.W));   // (+ t_src, + 2 t * W)
    const unsigned dybase = ((unsigned)n * p.dy_nstride + (unsigned)od * (unsigned)(p.OH * p.OW)) * ES;

    auto x_issue = [&](int t) __attribute__((always_inline)) {            // rows 2 t, 2 t + 1
#pragma unroll
      for (int j = 0; j < XT; ++j) {
        const int h = 2 * t + t_row[j];
        const bool rok = t_pl[j] >= 0 && h < p.H && (unsigned)(d0 + t_pl[j]) < (unsigned)p.D && grp * p.g + t_pl[j] < p.KD;
        const unsigned rowoff = xbase + t_src[j] + (unsigned)(2 * t * p.W);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int m = t_m[j] + i;
          const bool ok = rok && m >= 0 && 2 * m + 1 < p.W;
          const unsigned v = ok ? (rowoff + (unsigned)(2 * m)) * ES : 0xffffffffu;
          if (F16) xr[j][i][0] = __builtin_amdgcn_raw_buffer_load_b32(bx, (int)v, 0, 0);
          else {
            xr[j][i][0] = __builtin_amdgcn_raw_buffer_load_b32(bx, (int)v, 0, 0);
            xr[j][i][F16 ? 0 : 1] = __builtin_amdgcn_raw_buffer_load_b32(bx, (int)(ok ? v + 4u : 0xffffffffu), 0, 0);
          }
        }
      }
    };
    auto x_store = [&](int t) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < XT; ++j) {
        if (t_pl[j] < 0) continue;
        const int slot = (2 * t + t_row[j]) % RING;
        unsigned char* d = XR + slot * p.RS + t_dst[j];
        if (F16) {
          const unsigned d0_ = xr[j][0][0], d1 = xr[j][1][0], d2 = xr[j][2][0];
          *reinterpret_cast<unsigned*>(d) = (d0_ & 0xffffu) | (d1 << 16);                       // E0: P0[m], P0[m+1]
          *reinterpret_cast<unsigned*>(d + p.ARRB) = (d1 & 0xffffu) | (d2 << 16);               // E1: P0[m+1], P0[m+2]
          *reinterpret_cast<unsigned*>(d + 2 * p.ARRB) = (d0_ >> 16) | (d1 & 0xffff0000u);      // O0
          *reinterpret_cast<unsigned*>(d + 3 * p.ARRB) = (d1 >> 16) | (d2 & 0xffff0000u);       // O1
        } else {
          const float e0 = __uint_as_float(xr[j][0][0]), o0 = __uint_as_float(xr[j][0][F16 ? 0 : 1]);
          const float e1 = __uint_as_float(xr[j][1][0]), o1 = __uint_as_float(xr[j][1][F16 ? 0 : 1]);
          const float e2 = __uint_as_float(xr[j][2][0]), o2 = __uint_as_float(xr[j][2][F16 ? 0 : 1]);
          unsigned w[4][3];
          if (MATH == 2) {
            split_bf16x3(e0, e1, w[0][0], w[0][1], w[0][2]);
            split_bf16x3(e1, e2, w[1][0], w[1][1], w[1][2]);
            split_bf16x3(o0, o1, w[2][0], w[2][1], w[2][2]);
            split_bf16x3(o1, o2, w[3][0], w[3][1], w[3][2]);
          } else {
            split_bf16x2(e0, e1, w[0][0], w[0][1]);
            split_bf16x2(e1, e2, w[1][0], w[1][1]);
            split_bf16x2(o0, o1, w[2][0], w[2][1]);
            split_bf16x2(o1, o2, w[3][0], w[3][1]);
          }
#pragma unroll
          for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int a = 0; a < 4; ++a) *reinterpret_cast<unsigned*>(d + pp * (4 * p.ARRB) + a * p.ARRB) = w[a][pp];
        }
      }
    };
    auto dy_issue = [&](int it) __attribute__((always_inline)) {          // stage `it` of this unit -> buffer it & 1
      const int ohr = it / p.nstg, sg = it - ohr * p.nstg;
      const unsigned rowb = dybase + (unsigned)((oh0 + ohr) * p.OW) * ES;
      for (int pc = wave; pc < npc_full; pc += 4) {
        const int st = pc / (p.nrt * NQ), r2 = pc - st * (p.nrt * NQ);
        const int rtile = r2 / NQ, q = r2 - rtile * NQ;
        const int step = sg * p.SS + st;
        const unsigned v = dy_voff(rtile, q, step);
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(rowb + (unsigned)(step * 16) * ES));
        dma16(ry, v, so, dyt_lds + (unsigned)((it & 1) * p.DYB + pc * 1024));
      }
    };

    // ---- unit prologue: the ring holds the previous unit's rows
    __syncthreads();
    for (int i = tid * 16; i < p.g * p.PLS; i += 256 * 16)
      *reinterpret_cast<uint4*>(XR + i) = make_uint4(0u, 0u, 0u, 0u);     // rows -ph .. -1 (and every slot not staged yet)
    if (u == u0)
      for (int i = tid * 16; i < p.ZB; i += 256 * 16) *reinterpret_cast<uint4*>(XZ + i) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (int t = oh0 - 2; t <= oh0 + 1; ++t)                               // rows 2 oh0 - 4 .. 2 oh0 + 3 (negative rows: the zeros)
      if (t >= 0) { x_issue(t); x_store(t); }
    dy_issue(0);
#pragma unroll
    for (int ct = 0; ct < NB; ++ct) rB[ct] = (2 * oh0 + colb[ct] - p.ph + 2 * RING) % RING;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // One stage in flight: the DMA of stage it + 1 and the x loads of the round that stage's row needs are issued in front of
    // the MFMAs of stage it and drained behind them; the second workgroup of the CU covers the wait.  (Measured and not
    // kept: three stage buffers with the end of the iteration waiting for stage it + 1 only by instruction count -- the
    // third buffer costs the second workgroup per CU, 0.80 vs 0.54 ms on the R(2+1)D stem in bf16x6.)
    for (int it = 0; it < total; ++it) {
      const int ohr = it / p.nstg, sg = it - ohr * p.nstg;
      const int oh = oh0 + ohr;
      if (it + 1 < total) dy_issue(it + 1);
      if (sg == 0) x_issue(oh + 2);
      if (wave_on) {
        const unsigned char* Ab = DYT + (it & 1) * p.DYB + lane * 16;
        unsigned baddr[NB];
#pragma unroll
        for (int ct = 0; ct < NB; ++ct) baddr[ct] = dead[ct] ? 0u : (unsigned)(p.ZB + rB[ct] * p.RS) + cB[ct];
        const int s_lo = sg * p.SS;
        int s_hi = s_lo + p.SS; if (s_hi > p.steps) s_hi = p.steps;
        for (int step = s_lo + ksp; step < s_hi; step += p.wk) {
          const int st = step - s_lo;
          // A fragments of the wave's two row tiles
          uint4 af[2][NP];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (F16) af[i][0] = *reinterpret_cast<const uint4*>(Ab + (st * p.nrt + r0 + i) * 1024);
            else {
              const float4 a0 = *reinterpret_cast<const float4*>(Ab + ((st * p.nrt + r0 + i) * 2) * 1024);
              const float4 a1 = *reinterpret_cast<const float4*>(Ab + ((st * p.nrt + r0 + i) * 2 + 1) * 1024);
              if (MATH == 2) {
                split_bf16x3(a0.x, a0.y, af[i][0].x, af[i][1].x, af[i][NP - 1].x);
                split_bf16x3(a0.z, a0.w, af[i][0].y, af[i][1].y, af[i][NP - 1].y);
                split_bf16x3(a1.x, a1.y, af[i][0].z, af[i][1].z, af[i][NP - 1].z);
                split_bf16x3(a1.z, a1.w, af[i][0].w, af[i][1].w, af[i][NP - 1].w);
              } else {
                split_bf16x2(a0.x, a0.y, af[i][0].x, af[i][NP - 1].x);
                split_bf16x2(a0.z, a0.w, af[i][0].y, af[i][NP - 1].y);
                split_bf16x2(a1.x, a1.y, af[i][0].z, af[i][NP - 1].z);
                split_bf16x2(a1.z, a1.w, af[i][0].w, af[i][NP - 1].w);
              }
            }
          }
          const unsigned boff = (unsigned)(16 * step + 8 * lh) * 2u;
          uint4 bf[NB][NP];
#pragma unroll
          for (int ct = 0; ct < NB; ++ct)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
              const unsigned* src = reinterpret_cast<const unsigned*>(smem + baddr[ct] + boff + pp * (4 * p.ARRB));
              bf[ct][pp] = make_uint4(src[0], src[1], src[2], src[3]);
            }
          auto mma = [&](f32x16& c, const uint4 (&xa)[NP], const uint4 (&yb)[NP]) __attribute__((always_inline)) {
            if (MATH == 3) {
              c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xa[0]), __builtin_bit_cast(f16x8, yb[0]), c, 0, 0, 0);
            } else {
              const bf16x8 xh = __builtin_bit_cast(bf16x8, xa[0]), xl = __builtin_bit_cast(bf16x8, xa[NP - 1]);
              const bf16x8 yh = __builtin_bit_cast(bf16x8, yb[0]), yl = __builtin_bit_cast(bf16x8, yb[NP - 1]);
              if (MATH == 2) {
                const bf16x8 xm = __builtin_bit_cast(bf16x8, xa[1]), ym = __builtin_bit_cast(bf16x8, yb[1]);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, ym, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, yh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ym, c, 0, 0, 0);
              } else {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, c, 0, 0, 0);
              }
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, c, 0, 0, 0);
            }
          };
          mma(acc[0], af[0], bf[0]);
          mma(acc[1], af[0], bf[1]);
          mma(acc[2], af[1], bf[0]);
          mma(acc[3], af[1], bf[1]);
          if (half) mma(acc[4], af[1], bf[2]); else mma(acc[4], af[0], bf[2]);
        }
      }
      if (sg == 0) x_store(oh + 2);
      if (sg == p.nstg - 1) {
#pragma unroll
        for (int ct = 0; ct < NB; ++ct) { rB[ct] += 2; if (rB[ct] >= RING) rB[ct] -= RING; }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- fold the two shares of the steps (KD == 1, two row tiles) through LDS, then one slab per split in dW's layout
  if (p.wk > 1) {
    float* red = reinterpret_cast<float*>(smem);               // [column half][NT][16][64]
    if (sel == 1) {
      float* q = red + half * (NT * 1024);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) q[(t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (sel == 0) {
      const float* q = red + half * (NT * 1024);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += q[(t * 16 + r) * 64 + lane];
    }
  }
  if (ksp == 0 && kd < p.KD) {
    float* sl = slab + (long long)split * p.nW;
    const int per_k = p.C * p.KD * p.KH * p.KW;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ct = t == 4 ? 2 : (t & 1);
      const int rtile = t == 4 ? r0 + half : r0 + (t >> 1);
      if (dead[ct]) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = rtile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (k < p.K) sl[(long long)k * per_k + dwoff[ct]] = acc[t][r];
      }
    }
  }
}

template <int MATH>
int launch_stw(dim3 grid, size_t lds, hipStream_t st, const void* x, const void* dy, float* slab, const StwParams& p) {
  static bool raised = false;
  if (lds > (48u << 10) && !raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_stem_kernel<MATH>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10) != hipSuccess) return GCA_ELAUNCH;
    raised = true;
  }
  hipLaunchKernelGGL((conv_wgrad_stem_kernel<MATH>), grid, dim3(256), lds, st, x, dy, slab, p);
  return gca_launch_status();
}

// units and splits of a geometry: `want` workgroups per group of tap planes (0: a default that fills the part twice over)
void stw_partition(const gca_conv_geom* g, int want, int& ohc, int& ohl, int& units, int& ups, int& splits) {
  const long long base = (long long)g->N * g->OD;
  const int nrt = g->K <= 64 ? 2 : 4;
  const int grp = g->kd > 1 ? (int)gca_ceil_div(g->kd, nrt == 2 ? 2 : 1) : 1;
  long long w = want > 0 ? want : gca_ceil_div(1024, grp);
  long long c = 1;
  if (w > base) {
    c = gca_ceil_div(w, base);
    const long long cmax = g->OH / 8 > 0 ? g->OH / 8 : 1;               // >= 8 rows per chunk: the 4 staging rounds of a chunk start stay small
    if (c > cmax) c = cmax;
  }
  ohl = (int)gca_ceil_div(g->OH, c);
  ohc = (int)gca_ceil_div(g->OH, ohl);
  const long long un = base * ohc;
  if (w > un) w = un;
  if (w < 1) w = 1;
  units = (int)un;
  ups = (int)gca_ceil_div(un, w);
  splits = (int)gca_ceil_div(un, ups);
}

bool stw_params(const gca_conv_geom* g, int math, int want, StwParams& p, size_t& lds) {
  const int np = math == 2 ? 3 : (math == 1 ? 2 : 1);
  const int es = math == 3 ? 2 : 4;
  p.K = g->K; p.C = g->C; p.D = g->D; p.H = g->H; p.W = g->W; p.OD = g->OD; p.OH = g->OH; p.OW = g->OW;
  p.KD = g->kd; p.KH = g->kh; p.KW = g->kw; p.pd = g->pd; p.ph = g->ph; p.pw = g->pw;
  p.nrt = g->K <= 64 ? 2 : 4;
  p.g = g->kd > 1 && p.nrt == 2 ? 2 : 1;               // (one tap plane per workgroup, two shares of the steps: 2.4 vs 1.8 ms on the R3D-50 stem)
  p.wk = p.nrt == 2 && p.g == 1 ? 2 : 1;
  p.ngroups = (int)gca_ceil_div(g->kd, p.g);
  if ((long long)g->N * g->OD * g->OH > 0x3fffffffLL) return false;
  stw_partition(g, want, p.ohc, p.ohl, p.units, p.ups, p.splits);
  p.steps = (int)gca_ceil_div(g->OW, 16);
  p.ncols = g->kh * g->C * g->kw;
  p.LQ = 8 * p.steps + 1;
  p.ARRB = 4 * p.LQ;
  p.RS = g->C * np * 4 * p.ARRB;
  p.PLS = RING * p.RS;
  p.ZB = (int)gca_round_up(np * 4 * p.ARRB, 16);
  // one dY stage: <= 16 pieces of 1 KB (<= 4 DMA instructions per wave)
  const int per_step = p.nrt * (es == 4 ? 2 : 1);
  p.SS = 16 / per_step; if (p.SS > p.steps) p.SS = p.steps;
  if (p.SS < 1) return false;
  p.nstg = (int)gca_ceil_div(p.steps, p.SS);
  p.DYB = p.SS * per_step * 1024;
  p.ntask = 2 * g->C * p.g * p.LQ;
  if (p.ntask > 256 * XT) return false;
  const long long cdhw = (long long)g->C * g->D * g->H * g->W;
  const long long osp = (long long)g->OD * g->OH * g->OW;
  p.x_nstride = (unsigned)(g->x_batch_stride ? g->x_batch_stride : cdhw);
  p.dy_nstride = (unsigned)((long long)g->K * osp);
  p.osp = (unsigned)osp;
  const long long xb = (long long)g->N * p.x_nstride * es, yb = (long long)g->N * p.dy_nstride * es;
  if (xb > 0xfffff000LL || yb > 0xfffff000LL) return false;          // 32-bit byte offsets
  p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
  p.nW = (long long)g->K * g->C * g->kd * g->kh * g->kw;
  p.slab_bytes = 0;
  size_t need = (size_t)p.ZB + (size_t)p.g * p.PLS + 2 * (size_t)p.DYB;
  const size_t red = (size_t)(p.wk - 1) * 2 * NT * 1024 * 4;
  if (red > need) need = red;
  lds = need;
  return lds <= (size_t)(160 << 10) - 512;
}

}  // namespace

namespace gca_conv {

// tune_wgrad_tile 14
bool wgrad_stem_ok(const gca_conv_geom* g, int math) {
  if (math < 1 || math > 3) return false;
  if ((math == 3) != (g->act_f16 != 0)) return false;
  if (g->C > 4 || g->K > 128) return false;
  if (g->sd != 1 || g->sh != 2 || g->sw != 2) return false;
  if (g->kh > RING - 2 || g->kw > 8 || g->kd > 8) return false;
  if (g->kh * g->C * g->kw > 32 * CT) return false;
  if (g->pw > g->kw || g->ph > g->kh || g->pd > g->kd) return false;
  if (g->pw - (g->pw & 1) > 4) return false;                          // shifts s >= -2 (arrays start at phase index -2)
  if (g->kw - 1 - g->pw > 3) return false;                           //        s <= 1
  if (g->W % 2 != 0 || g->OW % 8 != 0) return false;
  if (g->ph > 3) return false;                                        // zeroed row slots: rows -ph .. -1 next to 7 live ones
  StwParams p; size_t lds;
  return stw_params(g, math, 0, p, lds);
}

int wgrad_stem_splits(const gca_conv_geom* g, int want) {
  int ohc, ohl, units, ups, splits;
  stw_partition(g, want, ohc, ohl, units, ups, splits);
  return splits;
}

int wgrad_stem_launch(const gca_conv_geom* g, int math, int splits, const void* x, const void* dy, float* slab, hipStream_t st) {
  StwParams p; size_t lds;
  if (!stw_params(g, math, g->tune_wgrad_splits, p, lds) || p.splits != splits) return GCA_EINVAL;
  const long long nblk = (long long)p.splits * p.ngroups;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return GCA_EINVAL;
  const dim3 grid((unsigned)nblk);
  if (math == 3) return launch_stw<3>(grid, lds, st, x, dy, slab, p);
  if (math == 2) return launch_stw<2>(grid, lds, st, x, dy, slab, p);
  return launch_stw<1>(grid, lds, st, x, dy, slab, p);
}

}  // namespace gca_conv
