// Weight gradient of the TEMPORAL convolutions ((kd,1,1), unit stride: resnet2p1d.py:13-36 conv*_t, s3d_1.py:53-57 conv_t) as a
// streaming, barrier-free MFMA kernel with a rolling register window over the taps.
//
//   dW[ko, c, a] = sum_{n, od, hw} dY[n, ko, od, hw] * X[n, c, od + a - pd, hw]
//
// conv3d_wgrad.hip treats every (channel, tap) pair as its own GEMM row, so each x element is fetched, split into its bf16
// parts and staged once PER TAP, and dy is split once per column tile: 10 VALU instructions per MFMA, matrix pipe 41 % busy
// (profiles/r02f_sq_counters_bf16x6.txt).  For a temporal conv a tap is a shift by whole planes, which needs no re-staging at all:
//
//   * a wave owns a 32*TM x 32 tile of (ko, c) for ALL kd taps: kd * TM accumulator tiles;
//   * it walks "units" = (clip, 16 consecutive positions of a plane) and, inside a unit, the planes d = 0, 1, 2, ...  One step
//     = one plane: the MFMA reduction index is the 16 positions; the A fragment is dY[., od, .], the B fragment of tap a is
//     X[., od + a - pd, .] -- the SAME fragment that tap a+1 used one step earlier.  So the kd B fragments live in a register
//     window that takes in ONE new plane per step; every x / dy element is fetched and split once per wave;
//   * operands travel global -> LDS by LDS-DMA (buffer_load ... lds, 1 KB pieces of 16 rows x 64 B, no VGPRs in flight) into a
//     per-wave ring PF steps deep, and LDS -> registers as fp32 fragments (lane = row, 8 consecutive positions = 2 x
//     ds_read_b128); the bf16 hi/mid/lo split happens on those registers.  The LDS image of a piece is lane-linear, so the
//     bank swizzle sits on the per-lane SOURCE address and on the read (cdna guide rule 21): quad q of row r lands in slot
//     4r + (q ^ ((r >> 1) & 3)) -- eight consecutive rows read eight distinct 16-byte columns;
//   * waves share nothing: no __syncthreads in the loop, counted s_waitcnt vmcnt keeps PF-1 steps of DMA in flight;
//   * taps whose plane lies in the zero padding are SKIPPED (wave-uniform), not multiplied by zeros;
//   * the four waves of a workgroup run the same tile over interleaved units (adjacent 64-byte halves of the same cache
//     lines) and fold their accumulators through LDS once at the end -> one fp32 slab per workgroup (split), reduced by
//     splitk_reduce_kernel like every other weight gradient (fixed order: deterministic).
#include "conv_common.h"

using namespace gca_conv;

namespace {

struct TsParams {
  int K, C, D, OD, HW, pd;
  int Kred;                      // C * KD
  int tilesM, tilesC, splits;
  int units, units_per_split;    // unit = (clip, chunk of 16 positions)
  int chunks;                    // HW / 16
  int S;                         // steps per unit, padded to a multiple of KD
  unsigned x_nstride, dy_nstride;   // elements between clips
  unsigned x_bytes, dy_bytes, slab_bytes;
  gca_magic m_chunks;
};

constexpr int PF = 4;            // ring depth (steps of DMA in flight: PF - 1 behind the one being read)

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer `rs` at per-lane byte offset `voff` to LDS bytes [lds, lds + 1024) in lane
// order.  Inline asm on purpose: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every LDS read that follows the
// builtin form (__builtin_amdgcn_raw_ptr_buffer_load_lds) whenever it cannot prove the two LDS ranges distinct -- a ring
// indexed at run time never can -- which drains the whole prefetch ring every step.  The asm form is invisible to its
// bookkeeping; the kernel counts its own vmcnt.  M0 (the DMA's LDS base) is saved and restored around the statement.
__device__ __forceinline__ void dma16(const i32x4 rs, unsigned voff, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds), "s"(rs) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));        // stride 0, no swizzle
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);                       // num_records (bytes): out-of-range lanes read zeros
  r.w = 0x00020000;
  return r;
}

template <int KD, int TM, int MATH>
__global__ __launch_bounds__(256) void conv_wgrad_ts_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ slab, const TsParams p) {
  static_assert(MATH == 1 || MATH == 2, "split-product arithmetic only");
  constexpr int NP = MATH == 2 ? 3 : 2;
  constexpr int NA = 2 * TM, NB = 2;                   // 1 KB pieces (16 rows x 64 B) of the A / B tile of one step
  constexpr int NL = NA + NB;                          // LDS-DMA instructions per step
  constexpr int STAGE = NL * 1024;
  constexpr int W0 = KD - 1;                           // (W0 - pd) warm-up steps: plane j enters at step j, dY plane od = j - (W0 - pd)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lh = lane >> 5, ll = lane & 31;
  unsigned char* const ring = smem + wave * (PF * STAGE);

  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tilesM * p.tilesC;
  const int split = bid / ntile; bid -= split * ntile;     // tiles of one split run next to each other on one XCD: shared L2 lines
  const int tileM = bid % p.tilesM, tileC = bid / p.tilesM;
  const int ko0 = tileM * (32 * TM), c0 = tileC * 32;

  // units of this wave: u = u0 + 4 i + wave, i = 0 .. nunits - 1
  const int u0 = split * p.units_per_split;
  int u1 = u0 + p.units_per_split; if (u1 > p.units) u1 = p.units;
  const int nunits = (u1 - u0 - wave + 3) >> 2 > 0 ? (u1 - u0 - wave + 3) >> 2 : 0;
  const int T = nunits * p.S;

  const i32x4 rx = make_rsrc(x, p.x_bytes), ry = make_rsrc(dy, p.dy_bytes);
  const unsigned ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(lds_void*)ring);

  // ---- DMA source addressing of this lane: row r = lane / 4 of a 16-row piece, quad kq = swizzled 4-position group
  const int r16 = lane >> 2;
  const int kq = (lane & 3) ^ ((r16 >> 1) & 3);
  const unsigned PL = (unsigned)p.HW;                                        // elements per plane
  unsigned a_row[NA], b_row[NB];                                             // element offset of (row, quad) inside a clip, plane 0
#pragma unroll
  for (int h = 0; h < NA; ++h) a_row[h] = (unsigned)min(ko0 + 16 * h + r16, p.K - 1) * (unsigned)p.OD * PL + 4u * (unsigned)kq;
#pragma unroll
  for (int h = 0; h < NB; ++h) b_row[h] = (unsigned)min(c0 + 16 * h + r16, p.C - 1) * (unsigned)p.D * PL + 4u * (unsigned)kq;

  // load stream: step (li-th unit of this wave, lj)
  int li = 0, lj = 0, lstage = 0;
  unsigned l_a = 0, l_b = 0;                                                  // element offset of (clip, chunk) of the load stream's unit
  auto unit_base = [&](int i) __attribute__((always_inline)) {
    const int u = u0 + 4 * min(i, max(nunits - 1, 0)) + wave;                 // past the end: re-read the last unit (never used)
    const unsigned uc = (unsigned)min(u, p.units - 1);
    const unsigned n = gca_fdiv(uc, p.m_chunks), ch = uc - n * (unsigned)p.chunks;
    l_a = n * p.dy_nstride + ch * 16u;
    l_b = n * p.x_nstride + ch * 16u;
  };
  auto issue_step = [&]() __attribute__((always_inline)) {
    const int od = min(max(lj - (W0 - p.pd), 0), p.OD - 1);                   // (planes outside the tensor are clamped: their taps are skipped)
    const int e = min(lj, p.D - 1);
    const unsigned st = ring_lds + (unsigned)(lstage * STAGE);
#pragma unroll
    for (int h = 0; h < NA; ++h) dma16(ry, (l_a + a_row[h] + (unsigned)od * PL) * 4u, st + h * 1024);
#pragma unroll
    for (int h = 0; h < NB; ++h) dma16(rx, (l_b + b_row[h] + (unsigned)e * PL) * 4u, st + (NA + h) * 1024);
    lstage = lstage + 1 == PF ? 0 : lstage + 1;
    if (++lj == p.S) { lj = 0; ++li; unit_base(li); }
  };

  // ---- fragment reads: lane (ll, lh) = row ll, positions 8 lh .. 8 lh + 7 = quads 2 lh, 2 lh + 1 of its row
  const int fr = ll & 15, fsw = (fr >> 1) & 3;
  const unsigned fo0 = (unsigned)((ll >> 4) * 1024 + (4 * fr + ((2 * lh) ^ fsw)) * 16);
  const unsigned fo1 = (unsigned)((ll >> 4) * 1024 + (4 * fr + ((2 * lh + 1) ^ fsw)) * 16);
  struct Frag { uint4 part[NP]; };
  auto split8 = [&](const float4 v0, const float4 v1, Frag& f) __attribute__((always_inline)) {
    if constexpr (MATH == 2) {
      split_bf16x3(v0.x, v0.y, f.part[0].x, f.part[1].x, f.part[2].x);
      split_bf16x3(v0.z, v0.w, f.part[0].y, f.part[1].y, f.part[2].y);
      split_bf16x3(v1.x, v1.y, f.part[0].z, f.part[1].z, f.part[2].z);
      split_bf16x3(v1.z, v1.w, f.part[0].w, f.part[1].w, f.part[2].w);
    } else {
      split_bf16x2(v0.x, v0.y, f.part[0].x, f.part[1].x);
      split_bf16x2(v0.z, v0.w, f.part[0].y, f.part[1].y);
      split_bf16x2(v1.x, v1.y, f.part[0].z, f.part[1].z);
      split_bf16x2(v1.z, v1.w, f.part[0].w, f.part[1].w);
    }
  };

  f32x16 acc[TM][KD];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int a = 0; a < KD; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][a][r] = 0.f;

  Frag win[KD];                    // B fragments: plane e sits in slot e % KD
  Frag af[TM], afn[TM];            // A fragments of this step / the next
  float4 ra[TM][2], rb[2];         // raw fp32 fragments of the next step

  auto mma = [&](int i, int a, const Frag& A, const Frag& B) __attribute__((always_inline)) {
    const bf16x8 xh = __builtin_bit_cast(bf16x8, A.part[0]), xl = __builtin_bit_cast(bf16x8, A.part[NP - 1]);
    const bf16x8 yh = __builtin_bit_cast(bf16x8, B.part[0]), yl = __builtin_bit_cast(bf16x8, B.part[NP - 1]);
    if constexpr (MATH == 2) {
      const bf16x8 xm = __builtin_bit_cast(bf16x8, A.part[1]), ym = __builtin_bit_cast(bf16x8, B.part[1]);
      acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, ym, acc[i][a], 0, 0, 0);
      acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc[i][a], 0, 0, 0);
      acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc[i][a], 0, 0, 0);
      acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, yh, acc[i][a], 0, 0, 0);
      acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ym, acc[i][a], 0, 0, 0);
    } else {
      acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc[i][a], 0, 0, 0);
      acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc[i][a], 0, 0, 0);
    }
    acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc[i][a], 0, 0, 0);
  };
  // read the raw fragments of the step held by ring stage `rs` (its DMA must have landed), then re-fill that stage
  auto read_raw = [&](int rs) __attribute__((always_inline)) {
    const unsigned char* st = ring + rs * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ra[i][0] = *reinterpret_cast<const float4*>(st + i * 2048 + fo0);
      ra[i][1] = *reinterpret_cast<const float4*>(st + i * 2048 + fo1);
    }
    rb[0] = *reinterpret_cast<const float4*>(st + NA * 1024 + fo0);
    rb[1] = *reinterpret_cast<const float4*>(st + NA * 1024 + fo1);
  };

  if (T > 0) {
    unit_base(0);
    // prologue: PF steps of DMA in flight, then the fragments of step 0
#pragma unroll
    for (int s = 0; s < PF; ++s) issue_step();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * NL) : "memory");
    read_raw(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue_step();                                           // step PF -> stage 0 (just read)
#pragma unroll
    for (int i = 0; i < TM; ++i) split8(ra[i][0], ra[i][1], af[i]);
    split8(rb[0], rb[1], win[0]);

    int j = 0, rstage = 1;
    for (int t = 0; t < T; t += KD) {
#pragma unroll
      for (int b = 0; b < KD; ++b) {
        // step t + b: plane index j (== b mod KD), dY plane od
        const int od = j - (W0 - p.pd);
        const bool live = od >= 0 && od < p.OD;
        // tap a multiplies plane e = j + a - W0, held in slot (b + a + 1) % KD; tap 0 first: the next step's plane takes its slot
        {
          const int e = j - W0;
          if (live && e >= 0 && e < p.D) {
#pragma unroll
            for (int i = 0; i < TM; ++i) mma(i, 0, af[i], win[(b + 1) % KD]);
          }
        }
        // fragments of step t + b + 1 (ring stage rstage): wait for its DMA, read, hand the stage back to the DMA engine
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * NL) : "memory");
        read_raw(rstage);
#pragma unroll
        for (int a = 1; a < KD; ++a) {
          const int e = j + a - W0;
          if (live && e >= 0 && e < p.D) {
#pragma unroll
            for (int i = 0; i < TM; ++i) mma(i, a, af[i], win[(b + a + 1) % KD]);
          }
          if (a == (KD > 3 ? 2 : 1)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_step();
            rstage = rstage + 1 == PF ? 0 : rstage + 1;
            split8(rb[0], rb[1], win[(b + 1) % KD]);
#pragma unroll
            for (int i = 0; i < TM; ++i) split8(ra[i][0], ra[i][1], afn[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = afn[i];
        if (++j == p.S) j = 0;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no DMA may land in the ring once it is reused below

  // ---- fold the four waves' accumulators (fixed order: (0 + 2) + (1 + 3)) and write the workgroup's slab
  constexpr int NR = TM * KD * 16;
  float* red = reinterpret_cast<float*>(smem);               // [2][NR][64]
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int a = 0; a < KD; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 2) * NR + (i * KD + a) * 16 + r) * 64 + lane] = acc[i][a][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int a = 0; a < KD; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][a][r] += red[(wave * NR + (i * KD + a) * 16 + r) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int a = 0; a < KD; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((i * KD + a) * 16 + r) * 64 + lane] = acc[i][a][r];
  }
  __syncthreads();
  if (wave != 0) return;
  float* out = slab + (long long)split * p.K * p.Kred;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, p.slab_bytes, 0x00020000);
  const int c = c0 + ll;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int a = 0; a < KD; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = ko0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = acc[i][a][r] + red[((i * KD + a) * 16 + r) * 64 + lane];
        const unsigned vo = (m < p.K && c < p.C) ? ((unsigned)m * (unsigned)p.Kred + (unsigned)(c * KD + a)) * 4u : 0xffffffffu;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, (int)vo, 0, 0);
      }
}

template <int KD, int TM>
int launch_ts(int math, dim3 grid, size_t lds, hipStream_t st, const float* x, const float* dy, float* slab, const TsParams& p) {
  static bool raised[2] = {false, false};
#define GCA_TS(M)                                                                                                          \
  {                                                                                                                        \
    if (!raised[M - 1]) {                                                                                                  \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_ts_kernel<KD, TM, M>),                             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10) != hipSuccess) return GCA_ELAUNCH;   \
      raised[M - 1] = true;                                                                                                \
    }                                                                                                                      \
    hipLaunchKernelGGL((conv_wgrad_ts_kernel<KD, TM, M>), grid, dim3(256), lds, st, x, dy, slab, p);                       \
  }
  if (math == 1) GCA_TS(1) else GCA_TS(2)
#undef GCA_TS
  return gca_launch_status();
}

}  // namespace

namespace gca_conv {

// tune_wgrad_tile 11 / 12: the streaming temporal kernel with 32 / 64 output channels per wave
bool wgrad_ts_ok(const gca_conv_geom* g, int tile, int math) {
  if (tile != 11 && tile != 12) return false;
  if (math != 1 && math != 2) return false;
  if (g->act_f16) return false;
  if (g->kh != 1 || g->kw != 1 || g->sd != 1 || g->sh != 1 || g->sw != 1 || g->ph != 0 || g->pw != 0) return false;
  if (g->kd != 3 && g->kd != 7) return false;
  if (g->pd > g->kd - 1) return false;
  if ((g->H * g->W) % 16 != 0) return false;
  if (tile == 12 && g->kd == 7) return false;                // (224 accumulator registers per wave: not built)
  return true;
}

int wgrad_ts_splits(const gca_conv_geom* g, int tile, int want) {
  const int tm = tile == 12 ? 2 : 1;
  const long long tiles = gca_ceil_div(g->K, 32 * tm) * gca_ceil_div(g->C, 32);
  const long long units = (long long)g->N * (g->H * g->W / 16);
  long long s = want > 0 ? want : gca_ceil_div(512, tiles);   // ~2 workgroups' worth of blocks per CU by default
  if (s > units / 4) s = units / 4;                           // every wave of a workgroup gets a unit
  if (s < 1) s = 1;
  const long long ups = gca_ceil_div(units, s);
  return (int)gca_ceil_div(units, ups);
}

int wgrad_ts_launch(const gca_conv_geom* g, int tile, int math, int splits, const float* x, const float* dy, float* slab,
                    hipStream_t st) {
  const int tm = tile == 12 ? 2 : 1;
  TsParams p;
  p.K = g->K; p.C = g->C; p.D = g->D; p.OD = g->OD; p.HW = g->H * g->W; p.pd = g->pd;
  p.Kred = g->C * g->kd;
  p.tilesM = (int)gca_ceil_div(g->K, 32 * tm);
  p.tilesC = (int)gca_ceil_div(g->C, 32);
  p.chunks = p.HW / 16;
  p.units = g->N * p.chunks;
  p.units_per_split = (int)gca_ceil_div(p.units, splits);
  p.splits = (int)gca_ceil_div(p.units, p.units_per_split);
  if (p.splits != splits) return GCA_EINVAL;
  const int w0 = g->kd - 1 - g->pd;
  const int s = (g->D > g->OD + w0 ? g->D : g->OD + w0);
  p.S = (int)gca_round_up(s, g->kd);
  const long long cdhw = (long long)g->C * g->D * p.HW;
  p.x_nstride = (unsigned)(g->x_batch_stride ? g->x_batch_stride : cdhw);
  p.dy_nstride = (unsigned)((long long)g->K * g->OD * p.HW);
  const long long xb = (long long)g->N * p.x_nstride * 4, yb = (long long)g->N * p.dy_nstride * 4;
  const long long sb = (long long)g->K * p.Kred * 4;
  p.x_bytes = xb > 0xfffff000LL ? 0xfffff000u : (unsigned)xb;
  p.dy_bytes = yb > 0xfffff000LL ? 0xfffff000u : (unsigned)yb;
  p.slab_bytes = sb > 0xfffff000LL ? 0xfffff000u : (unsigned)sb;
  p.m_chunks = gca_make_magic((unsigned)p.chunks);
  const long long nblk = (long long)p.tilesM * p.tilesC * p.splits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return GCA_EINVAL;
  const size_t ring = (size_t)4 * PF * (2 * tm + 2) * 1024;
  const size_t red = (size_t)2 * tm * g->kd * 16 * 64 * 4;
  const size_t lds = ring > red ? ring : red;
  const dim3 grid((unsigned)nblk);
  if (g->kd == 7) return launch_ts<7, 1>(math, grid, lds, st, x, dy, slab, p);
  if (tm == 2) return launch_ts<3, 2>(math, grid, lds, st, x, dy, slab, p);
  return launch_ts<3, 1>(math, grid, lds, st, x, dy, slab, p);
}

}  // namespace gca_conv
