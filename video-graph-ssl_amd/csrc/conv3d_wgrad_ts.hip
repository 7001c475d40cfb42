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
  // optional input transform: x stands for relu(x * in_scale[c] + in_shift[c]) (the producer's BatchNorm + ReLU, never
  // materialised); a lane's fragment row IS a channel, so the pair sits in two registers for the whole kernel
  const float* in_scale;
  const float* in_shift;
};

constexpr int PF = 4;            // ring depth (steps of DMA in flight: PF - 1 behind the one being read)

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// The LDS-DMA pieces of one step: NA + NB loads of 64 lanes x 16 bytes each, buffer `rsA` / `rsB`, per-lane byte offsets
// voff[] (the lane's row and swizzled quad: constant for the whole kernel) + the wave-uniform byte offset of the step in an
// SGPR, landing at LDS bytes lds, lds + 1024, ... in lane order.  Inline asm on purpose: hipcc (ROCm 7.2) puts
// `s_waitcnt vmcnt(0)` in front of every LDS read that follows the builtin form (__builtin_amdgcn_raw_ptr_buffer_load_lds)
// whenever it cannot prove the two LDS ranges distinct -- a ring indexed at run time never can -- which drains the whole
// prefetch ring every step.  The asm form is invisible to its bookkeeping; the kernel counts its own vmcnt.  M0 (the DMA's
// LDS base) is a reserved register that hipcc itself re-loads before every use, so it is written here without being saved.
__device__ __forceinline__ void dma16(const i32x4 rs, unsigned voff, unsigned soff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %3, %1 offen lds"
               : : "v"(voff), "s"(soff), "s"(lds), "s"(rs) : "memory");
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
// bank swizzle of a 16-row x 64-byte piece: quad q of row r sits in 16-byte slot 4 r + (q ^ swz(r)).  Eight consecutive
// rows (one ds_read_b128 pass on a 32-bank LDS) hit eight distinct 16-byte columns, sixteen consecutive rows sixteen
// distinct ones (64 banks).
__device__ __forceinline__ int swz(int r) { return ((r >> 1) & 3) ^ ((r >> 3) & 1); }

template <int KD, int TM, int MATH>
__global__ __launch_bounds__(256) void conv_wgrad_ts_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ slab, const TsParams p) {
  static_assert(MATH == 1 || MATH == 2, "split-product arithmetic only");
  constexpr int NP = MATH == 2 ? 3 : 2;
  constexpr int NA = 2 * TM, NB = 2;                   // 1 KB pieces (16 rows x 64 B) of the A / B tile of one step
  constexpr int NL = NA + NB;                          // LDS-DMA instructions per step
  constexpr int STAGE = NL * 1024;
  constexpr int W0 = KD - 1;                           // plane j enters at step j; dY plane od = j - (W0 - pd); tap a reads plane j + a - W0
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // (scalar: everything a wave derives from it stays in SGPRs)
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
  const int kq = (lane & 3) ^ swz(r16);
  const unsigned PL = (unsigned)p.HW;                                        // elements per plane
  unsigned a_row[NA], b_row[NB];                                             // BYTE offset of this lane's (row, quad) inside a clip, plane 0
#pragma unroll
  for (int h = 0; h < NA; ++h) a_row[h] = ((unsigned)min(ko0 + 16 * h + r16, p.K - 1) * (unsigned)p.OD * PL + 4u * (unsigned)kq) * 4u;
#pragma unroll
  for (int h = 0; h < NB; ++h) b_row[h] = ((unsigned)min(c0 + 16 * h + r16, p.C - 1) * (unsigned)p.D * PL + 4u * (unsigned)kq) * 4u;

  // load stream (all scalar): step (li-th unit of this wave, lj)
  int li = 0, lj = 0, lstage = 0;
  unsigned l_a = 0, l_b = 0;                                                  // element offset of (clip, chunk) of the load stream's unit
  auto unit_base = [&](int i) __attribute__((always_inline)) {
    const int u = u0 + 4 * min(i, max(nunits - 1, 0)) + wave;                 // past the end: re-read the last unit (never used)
    const unsigned uc = (unsigned)min(u, p.units - 1);
    const unsigned n = uc / (unsigned)p.chunks, ch = uc - n * (unsigned)p.chunks;
    l_a = n * p.dy_nstride + ch * 16u;
    l_b = n * p.x_nstride + ch * 16u;
  };
  auto issue_step = [&]() __attribute__((always_inline)) {
    const int od = min(max(lj - (W0 - p.pd), 0), p.OD - 1);                   // (planes outside the tensor are clamped: their taps are skipped)
    const int e = min(lj, p.D - 1);
    const unsigned st = ring_lds + (unsigned)(lstage * STAGE);
    const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane((int)((l_a + (unsigned)od * PL) * 4u));
    const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((l_b + (unsigned)e * PL) * 4u));
#pragma unroll
    for (int h = 0; h < NA; ++h) dma16(ry, a_row[h], sa, st + h * 1024);
#pragma unroll
    for (int h = 0; h < NB; ++h) dma16(rx, b_row[h], sb, st + (NA + h) * 1024);
    lstage = lstage + 1 == PF ? 0 : lstage + 1;
    if (++lj == p.S) { lj = 0; ++li; unit_base(li); }
  };

  // ---- fragment reads: lane (ll, lh) = row ll, positions 8 lh .. 8 lh + 7 = quads 2 lh, 2 lh + 1 of its row
  const int fr = ll & 15, fsw = swz(fr);
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

  const bool xf = p.in_scale != nullptr;
  const float xsc = xf ? p.in_scale[min(c0 + ll, p.C - 1)] : 1.f, xsh = xf ? p.in_shift[min(c0 + ll, p.C - 1)] : 0.f;
  auto xform = [&](float4& v) __attribute__((always_inline)) {      // bn.hip bn_apply_kernel's arithmetic
    v.x = fmaxf(v.x * xsc + xsh, 0.f); v.y = fmaxf(v.y * xsc + xsh, 0.f);
    v.z = fmaxf(v.z * xsc + xsh, 0.f); v.w = fmaxf(v.w * xsc + xsh, 0.f);
  };
  Frag win[KD];                    // B fragments: plane e sits in slot e % KD
  Frag af[TM], afn[TM];            // A fragments of this step / the next
  float4 ra[TM][2], rb[2];         // raw fp32 fragments of the next step

  // one of the 3 / 6 products of a (row tile, tap) pair; `q` walks them so that consecutive MFMAs of the fast path hit
  // DIFFERENT accumulators (a dependent MFMA waits for the previous one to leave the pipe)
  auto mma1 = [&](int i, int a, int q, const Frag& A, const Frag& B) __attribute__((always_inline)) {
    constexpr int XA[6] = {1, 2, 0, 1, 0, 0}, XB[6] = {1, 0, 2, 0, 1, 0};      // bf16x6: mm, lh, hl, mh, hm, hh (small terms first)
    constexpr int YA[3] = {1, 0, 0}, YB[3] = {0, 1, 0};                        // bf16x3: lh, hl, hh
    const int pa = MATH == 2 ? XA[q] : YA[q], pb = MATH == 2 ? XB[q] : YB[q];
    acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A.part[pa]), __builtin_bit_cast(bf16x8, B.part[pb]),
                                                        acc[i][a], 0, 0, 0);
  };
  constexpr int NQ = MATH == 2 ? 6 : 3;
  auto mma = [&](int i, int a, const Frag& A, const Frag& B) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) mma1(i, a, q, A, B);
  };
  // read the raw fragments of the step held by ring stage `rs` (its DMA must have landed)
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
    if (xf) { xform(rb[0]); xform(rb[1]); }
    split8(rb[0], rb[1], win[0]);

    int j = 0, rstage = 1;
    for (int t = 0; t < T; t += KD) {
#pragma unroll
      for (int b = 0; b < KD; ++b) {
        // step t + b: plane index j (== b mod KD), dY plane od.  Tap a multiplies plane e = j + a - W0, held in window slot
        // (b + a + 1) % KD; it is live iff 0 <= e < D (and the dY plane exists): one bit per tap, wave-uniform.
        const int od = j - (W0 - p.pd);
        const int alo = max(W0 - j, 0), ahi = min(p.D - 1 + W0 - j, KD - 1);
        const unsigned mask = (od >= 0 && od < p.OD && alo <= ahi) ? (((2u << ahi) - 1u) & ~((1u << alo) - 1u)) : 0u;
        // (one code path with a uniform branch per tap: a second, branch-free copy of the step for interior planes made hipcc
        // keep two sets of accumulators -- 256 AGPRs, one wave per SIMD -- and lost more than the interleaving gained)
        if (mask & 1u) {
#pragma unroll
          for (int i = 0; i < TM; ++i) mma(i, 0, af[i], win[(b + 1) % KD]);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * NL) : "memory");
        read_raw(rstage);
#pragma unroll
        for (int a = 1; a < KD; ++a) {
          if (mask & (1u << a)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) mma(i, a, af[i], win[(b + a + 1) % KD]);
          }
          if (a == (KD > 3 ? 2 : 1)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_step();
            if (xf) { xform(rb[0]); xform(rb[1]); }
            split8(rb[0], rb[1], win[(b + 1) % KD]);
#pragma unroll
            for (int i = 0; i < TM; ++i) split8(ra[i][0], ra[i][1], afn[i]);
          }
        }
        rstage = rstage + 1 == PF ? 0 : rstage + 1;
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

// ---------------------------------------------------------------------------------------------------------------------------
// The same streaming scheme for the SPATIAL (1,3,3) convolutions (unit stride, pad 1: resnet2p1d.py conv*_s, s3d_1.py conv_s):
//
//   dW[ko, c, (bh, bw)] = sum_{n, d, oh, ow} dY[n, ko, d, oh, ow] * X[n, c, d, oh + bh - 1, ow + bw - 1]
//
// A unit is (clip, plane d, chunk of 16 output columns); a step is one row.  The window holds THREE rows (bh) of THREE
// fragments each (bw): the three column-shifted fragments of a new row are cut out of ONE 16-float-per-lane read of the row
// segment [w0 - 4, w0 + 20) -- positions 3..10, 4..11, 5..12 of it -- so a shift by one column costs no shuffle and no
// misaligned LDS access.  Zero padding is the DMA's range check: a 4-position quad of a row lies entirely inside or entirely
// outside [0, W) when W % 4 == 0, and an outside quad (or a whole row outside [0, H), or a channel tail) is fetched through
// an all-ones offset / an empty buffer descriptor and lands in LDS as zeros.  So all nine taps are always multiplied (the
// two padding rows cost 2 / (3 H) of the MFMAs) and the step body has no branches: products are issued interleaved over
// the nine accumulators.
struct SsParams {
  int K, C, D, H, W;
  int Kred;                      // C * 9
  int tilesM, tilesC, splits;
  int units, units_per_split;    // unit = (clip, plane, chunk of 16 columns)
  int chunks;                    // ceil(W / 16)
  int S;                         // steps per unit: rows 0 .. H (one past the end), padded to a multiple of 3
  unsigned x_nstride, dy_nstride;
  unsigned x_bytes, dy_bytes, slab_bytes;
};

// quads of a 32-row x 6-quad B piece set: slot s = 6 r + (q ^ sw6(r)); eight / sixteen consecutive rows read distinct columns
__device__ __forceinline__ int sw6(int r) { return ((r >> 2) ^ (r >> 3)) & 1; }

// MASK: rows narrower than one chunk with W % 4 != 0 (W = 14: the layer-2 maps of R(2+1)D-18).  The last quad of a row then
// also holds the first elements of the NEXT row; those columns (>= W) are zeroed in registers after the fragment read --
// the same lanes and registers every step, since there is one chunk per row (w0 = 0).
template <int TM, int MATH, bool MASK>
__global__ __launch_bounds__(256) void conv_wgrad_ss_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ slab, const SsParams p) {
  static_assert(MATH == 1 || MATH == 2, "split-product arithmetic only");
  constexpr int NP = MATH == 2 ? 3 : 2;
  constexpr int KH = 3, KW = 3;
  constexpr int NA = 2 * TM, NB = 3;                   // A: 16 rows x 4 quads per piece; B: 32 rows x 6 quads = 192 slots = 3 pieces
  constexpr int NL = NA + NB;
  constexpr int STAGE = NL * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, ll = lane & 31;
  unsigned char* const ring = smem + wave * (PF * STAGE);

  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tilesM * p.tilesC;
  const int split = bid / ntile; bid -= split * ntile;
  const int tileM = bid % p.tilesM, tileC = bid / p.tilesM;
  const int ko0 = tileM * (32 * TM), c0 = tileC * 32;

  const int u0 = split * p.units_per_split;
  int u1 = u0 + p.units_per_split; if (u1 > p.units) u1 = p.units;
  const int nunits = (u1 - u0 - wave + 3) >> 2 > 0 ? (u1 - u0 - wave + 3) >> 2 : 0;
  const int T = nunits * p.S;

  const i32x4 rx = make_rsrc(x, p.x_bytes), ry = make_rsrc(dy, p.dy_bytes);
  i32x4 rnull = rx; rnull.z = 0;                                             // empty buffer: every lane reads zeros
  const unsigned ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(lds_void*)ring);
  const unsigned PLN = (unsigned)(p.H * p.W);

  // ---- DMA source addressing.  A piece h: lane -> (row 16 h + lane / 4, swizzled quad).  B piece h: slot 64 h + lane -> (row, quad of 6).
  const int r16 = lane >> 2;
  const int aq = (lane & 3) ^ swz(r16);
  unsigned a_lane[NA];                                                        // byte offset of the lane's row inside a clip (plane 0, row 0), or all-ones
#pragma unroll
  for (int h = 0; h < NA; ++h) {
    const int ko = ko0 + 16 * h + r16;
    a_lane[h] = ko < p.K ? (unsigned)ko * (unsigned)p.D * PLN * 4u : 0xffffffffu;
  }
  int b_q[NB];
  unsigned b_lane[NB];
#pragma unroll
  for (int h = 0; h < NB; ++h) {
    const int sidx = 64 * h + lane, r = sidx / 6, q = sidx - 6 * r;
    b_q[h] = q ^ sw6(r);
    const int c = c0 + r;
    b_lane[h] = c < p.C ? (unsigned)c * (unsigned)p.D * PLN * 4u : 0xffffffffu;
  }

  // load stream (scalar except the per-unit lane offsets): step (li-th unit of this wave, row lj)
  int li = 0, lj = 0, lstage = 0;
  unsigned l_a = 0, l_b = 0;                                                  // element offset of (clip, plane) + chunk start
  unsigned a_voff[NA], b_voff[NB];                                            // lane offsets of the load stream's unit (quad validity folded in)
  auto unit_base = [&](int i) __attribute__((always_inline)) {
    const int u = u0 + 4 * min(i, max(nunits - 1, 0)) + wave;
    const unsigned uc = (unsigned)min(u, p.units - 1);
    const unsigned nd = uc / (unsigned)p.chunks, ch = uc - nd * (unsigned)p.chunks;
    const unsigned n = nd / (unsigned)p.D, d = nd - n * (unsigned)p.D;
    const int w0 = (int)ch * 16;
    // the B segment starts 4 columns left of the chunk.  Offsets must stay non-negative on both sides (the scalar part is
    // zero-extended, the lane part is range-checked as unsigned): the 4 columns come off the scalar part where it can
    // afford them (w0 > 0) and off the lane part otherwise (w0 == 0: the only lanes that would go negative fetch the
    // columns left of the row, which are outside anyway)
    const int adj = w0 > 0 ? 4 : 0;
    l_a = n * p.dy_nstride + d * PLN + (unsigned)w0;
    l_b = n * p.x_nstride + d * PLN + (unsigned)(w0 - adj);
#pragma unroll
    for (int h = 0; h < NA; ++h)                                              // quad [w0 + 4 aq, +4) of the row: inside iff it starts before W
      a_voff[h] = (w0 + 4 * aq < p.W && a_lane[h] != 0xffffffffu) ? a_lane[h] + 16u * (unsigned)aq : 0xffffffffu;
#pragma unroll
    for (int h = 0; h < NB; ++h) {                                            // quad [w0 - 4 + 4 q, +4)
      const int w = w0 - 4 + 4 * b_q[h];
      b_voff[h] = (w >= 0 && w < p.W && b_lane[h] != 0xffffffffu) ? b_lane[h] + (unsigned)(16 * b_q[h] - 16 + 4 * adj) : 0xffffffffu;
    }
  };
  auto issue_step = [&]() __attribute__((always_inline)) {
    const int oh = lj - 1;                                                    // dY row of the step that x row lj completes
    const unsigned st = ring_lds + (unsigned)(lstage * STAGE);
    const bool va = oh >= 0 && oh < p.H, vb = lj < p.H;
    const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane((int)((l_a + (unsigned)max(oh, 0) * (unsigned)p.W) * 4u));
    const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((l_b + (unsigned)lj * (unsigned)p.W) * 4u));
    const i32x4 ra_ = va ? ry : rnull, rb_ = vb ? rx : rnull;
#pragma unroll
    for (int h = 0; h < NA; ++h) dma16(ra_, a_voff[h], sa, st + h * 1024);
#pragma unroll
    for (int h = 0; h < NB; ++h) dma16(rb_, b_voff[h], sb, st + (NA + h) * 1024);
    lstage = lstage + 1 == PF ? 0 : lstage + 1;
    if (++lj == p.S) { lj = 0; ++li; unit_base(li); }
  };

  // ---- fragment reads
  const int fr = ll & 15, fsw = swz(fr);
  const unsigned fa0 = (unsigned)((ll >> 4) * 1024 + (4 * fr + ((2 * lh) ^ fsw)) * 16);
  const unsigned fa1 = (unsigned)((ll >> 4) * 1024 + (4 * fr + ((2 * lh + 1) ^ fsw)) * 16);
  unsigned fb[4];                                                             // quads 2 lh .. 2 lh + 3 of row ll of the B tile
#pragma unroll
  for (int k = 0; k < 4; ++k) fb[k] = (unsigned)(NA * 1024 + (6 * ll + ((2 * lh + k) ^ sw6(ll))) * 16);
  struct Frag { uint4 part[NP]; };
  auto split8v = [&](const float (&v)[8], Frag& f) __attribute__((always_inline)) {
    if constexpr (MATH == 2) {
      split_bf16x3(v[0], v[1], f.part[0].x, f.part[1].x, f.part[2].x);
      split_bf16x3(v[2], v[3], f.part[0].y, f.part[1].y, f.part[2].y);
      split_bf16x3(v[4], v[5], f.part[0].z, f.part[1].z, f.part[2].z);
      split_bf16x3(v[6], v[7], f.part[0].w, f.part[1].w, f.part[2].w);
    } else {
      split_bf16x2(v[0], v[1], f.part[0].x, f.part[1].x);
      split_bf16x2(v[2], v[3], f.part[0].y, f.part[1].y);
      split_bf16x2(v[4], v[5], f.part[0].z, f.part[1].z);
      split_bf16x2(v[6], v[7], f.part[0].w, f.part[1].w);
    }
  };

  f32x16 acc[TM][KH * KW];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int a = 0; a < KH * KW; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][a][r] = 0.f;

  Frag win[KH][KW];               // x row e sits in slot e % 3; [bw] = the fragment shifted by bw - 1 columns
#pragma unroll
  for (int a = 0; a < KH; ++a)
#pragma unroll
    for (int c = 0; c < KW; ++c)
#pragma unroll
      for (int q = 0; q < NP; ++q) win[a][c].part[q] = make_uint4(0u, 0u, 0u, 0u);     // (row -1 of the first unit)
  Frag af[TM], afn[TM];
  float ra[TM][8], rb[16];

  auto mma1 = [&](int i, int a, int q, const Frag& A, const Frag& B) __attribute__((always_inline)) {
    constexpr int XA[6] = {1, 2, 0, 1, 0, 0}, XB[6] = {1, 0, 2, 0, 1, 0};
    constexpr int YA[3] = {1, 0, 0}, YB[3] = {0, 1, 0};
    const int pa = MATH == 2 ? XA[q] : YA[q], pb = MATH == 2 ? XB[q] : YB[q];
    acc[i][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A.part[pa]), __builtin_bit_cast(bf16x8, B.part[pb]),
                                                        acc[i][a], 0, 0, 0);
  };
  constexpr int NQ = MATH == 2 ? 6 : 3;
  auto read_raw = [&](int rs) __attribute__((always_inline)) {
    const unsigned char* st = ring + rs * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float4 v0 = *reinterpret_cast<const float4*>(st + i * 2048 + fa0), v1 = *reinterpret_cast<const float4*>(st + i * 2048 + fa1);
      ra[i][0] = v0.x; ra[i][1] = v0.y; ra[i][2] = v0.z; ra[i][3] = v0.w; ra[i][4] = v1.x; ra[i][5] = v1.y; ra[i][6] = v1.z; ra[i][7] = v1.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(st + fb[k]);
      rb[4 * k] = v.x; rb[4 * k + 1] = v.y; rb[4 * k + 2] = v.z; rb[4 * k + 3] = v.w;
    }
    if constexpr (MASK) {                       // column of raw x element r: 8 lh + r - 4; of dY element e: 8 lh + e
#pragma unroll
      for (int r = 0; r < 16; ++r) rb[r] = (8 * lh + r - 4 < p.W) ? rb[r] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) ra[i][e] = (8 * lh + e < p.W) ? ra[i][e] : 0.f;
    }
  };
  auto split_next = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int bw = 0; bw < KW; ++bw) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rb[3 + bw + e];                  // raw index r <-> column w0 - 4 + 8 lh + r
      split8v(v, win[slot][bw]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) split8v(ra[i], afn[i]);
  };

  if (T > 0) {
    unit_base(0);
#pragma unroll
    for (int s = 0; s < PF; ++s) issue_step();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * NL) : "memory");
    read_raw(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue_step();
    split_next(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = afn[i];

    int j = 0, rstage = 1;
    for (int t = 0; t < T; t += KH) {
#pragma unroll
      for (int b = 0; b < KH; ++b) {
        // step t + b: x row j (== b mod 3) has just entered the window; dY row oh = j - 1; tap (bh, bw) reads row j + bh - 2 = slot (b + bh + 1) % 3
        const int oh = j - 1;
        const bool live = oh >= 0 && oh < p.H;
        if (live) {
          // tap row bh = 0 first: the next step's row takes its slot
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int bw = 0; bw < KW; ++bw)
#pragma unroll
              for (int i = 0; i < TM; ++i) mma1(i, bw, q, af[i], win[(b + 1) % KH][bw]);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * NL) : "memory");
        read_raw(rstage);
        if (live) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int bw = 0; bw < KW; ++bw)
#pragma unroll
              for (int i = 0; i < TM; ++i) mma1(i, KW + bw, q, af[i], win[(b + 2) % KH][bw]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_step();
        split_next((b + 1) % KH);
        if (live) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int bw = 0; bw < KW; ++bw)
#pragma unroll
              for (int i = 0; i < TM; ++i) mma1(i, 2 * KW + bw, q, af[i], win[b % KH][bw]);
        }
        rstage = rstage + 1 == PF ? 0 : rstage + 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = afn[i];
        if (++j == p.S) j = 0;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  constexpr int NT = KH * KW;
  constexpr int NR = TM * NT * 16;
  float* red = reinterpret_cast<float*>(smem);
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 2) * NR + (i * NT + a) * 16 + r) * 64 + lane] = acc[i][a][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][a][r] += red[(wave * NR + (i * NT + a) * 16 + r) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((i * NT + a) * 16 + r) * 64 + lane] = acc[i][a][r];
  }
  __syncthreads();
  if (wave != 0) return;
  float* out = slab + (long long)split * p.K * p.Kred;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, p.slab_bytes, 0x00020000);
  const int c = c0 + ll;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = ko0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = acc[i][a][r] + red[((i * NT + a) * 16 + r) * 64 + lane];
        const unsigned vo = (m < p.K && c < p.C) ? ((unsigned)m * (unsigned)p.Kred + (unsigned)(c * NT + a)) * 4u : 0xffffffffu;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, (int)vo, 0, 0);
      }
}

template <int TM, bool MASK>
int launch_ss(int math, dim3 grid, size_t lds, hipStream_t st, const float* x, const float* dy, float* slab, const SsParams& p) {
  static bool raised[2] = {false, false};
#define GCA_SS(M)                                                                                                          \
  {                                                                                                                        \
    if (!raised[M - 1]) {                                                                                                  \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_ss_kernel<TM, M, MASK>),                                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10) != hipSuccess) return GCA_ELAUNCH;   \
      raised[M - 1] = true;                                                                                                \
    }                                                                                                                      \
    hipLaunchKernelGGL((conv_wgrad_ss_kernel<TM, M, MASK>), grid, dim3(256), lds, st, x, dy, slab, p);                           \
  }
  if (math == 1) GCA_SS(1) else GCA_SS(2)
#undef GCA_SS
  return gca_launch_status();
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

// tune_wgrad_tile 11 / 12: the streaming temporal kernel with 32 / 64 output channels per wave; 13: the streaming (1,3,3) kernel
bool wgrad_ts_ok(const gca_conv_geom* g, int tile, int math) {
  if (tile == 13) {
    if (math != 1 && math != 2) return false;
    if (g->act_f16) return false;
    if (g->kd != 1 || g->kh != 3 || g->kw != 3 || g->sd != 1 || g->sh != 1 || g->sw != 1) return false;
    if (g->pd != 0 || g->ph != 1 || g->pw != 1) return false;
    if (g->H < 2) return false;
    if (g->W % 4 != 0 && g->W > 16) return false;                 // whole quads outside the row, or ONE chunk per row with its tail masked
    return true;
  }
  if (tile != 11 && tile != 12) return false;
  if (math != 1 && math != 2) return false;
  if (g->act_f16) return false;
  if (g->kh != 1 || g->kw != 1 || g->sd != 1 || g->sh != 1 || g->sw != 1 || g->ph != 0 || g->pw != 0) return false;
  if (g->kd != 3 && g->kd != 7) return false;
  if (g->pd > g->kd - 1) return false;
  if ((g->H * g->W) % 16 != 0) return false;
  return true;
}

int wgrad_ts_splits(const gca_conv_geom* g, int tile, int want) {
  const int tm = tile == 12 ? 2 : 1;
  const long long tiles = gca_ceil_div(g->K, 32 * tm) * gca_ceil_div(g->C, 32);
  const long long units = tile == 13 ? (long long)g->N * g->D * gca_ceil_div(g->W, 16) : (long long)g->N * (g->H * g->W / 16);
  long long s = want > 0 ? want : gca_ceil_div(512, tiles);   // ~2 workgroups' worth of blocks per CU by default
  if (s > units / 4) s = units / 4;                           // every wave of a workgroup gets a unit
  if (s < 1) s = 1;
  const long long ups = gca_ceil_div(units, s);
  return (int)gca_ceil_div(units, ups);
}

static int wgrad_ss_launch(const gca_conv_geom* g, int math, int splits, const float* x, const float* dy, float* slab, hipStream_t st) {
  SsParams p;
  p.K = g->K; p.C = g->C; p.D = g->D; p.H = g->H; p.W = g->W;
  p.Kred = g->C * 9;
  p.tilesM = (int)gca_ceil_div(g->K, 32);
  p.tilesC = (int)gca_ceil_div(g->C, 32);
  p.chunks = (int)gca_ceil_div(g->W, 16);
  p.units = g->N * g->D * p.chunks;
  p.units_per_split = (int)gca_ceil_div(p.units, splits);
  p.splits = (int)gca_ceil_div(p.units, p.units_per_split);
  if (p.splits != splits) return GCA_EINVAL;
  p.S = (int)gca_round_up(g->H + 1, 3);
  const long long cdhw = (long long)g->C * g->D * g->H * g->W;
  p.x_nstride = (unsigned)(g->x_batch_stride ? g->x_batch_stride : cdhw);
  p.dy_nstride = (unsigned)((long long)g->K * g->D * g->H * g->W);
  const long long xb = (long long)g->N * p.x_nstride * 4, yb = (long long)g->N * p.dy_nstride * 4;
  const long long sb = (long long)g->K * p.Kred * 4;
  p.x_bytes = xb > 0xfffff000LL ? 0xfffff000u : (unsigned)xb;
  p.dy_bytes = yb > 0xfffff000LL ? 0xfffff000u : (unsigned)yb;
  p.slab_bytes = sb > 0xfffff000LL ? 0xfffff000u : (unsigned)sb;
  const long long nblk = (long long)p.tilesM * p.tilesC * p.splits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return GCA_EINVAL;
  const size_t ring = (size_t)4 * PF * (2 + 3) * 1024;
  const size_t red = (size_t)2 * 9 * 16 * 64 * 4;
  if (g->W % 4) return launch_ss<1, true>(math, dim3((unsigned)nblk), ring > red ? ring : red, st, x, dy, slab, p);
  return launch_ss<1, false>(math, dim3((unsigned)nblk), ring > red ? ring : red, st, x, dy, slab, p);
}

int wgrad_ts_launch(const gca_conv_geom* g, int tile, int math, int splits, const float* x, const float* dy, float* slab,
                    hipStream_t st, const float* in_scale, const float* in_shift) {
  if (tile == 13) return in_scale ? GCA_EINVAL : wgrad_ss_launch(g, math, splits, x, dy, slab, st);
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
  p.in_scale = in_scale; p.in_shift = in_shift;
  const long long nblk = (long long)p.tilesM * p.tilesC * p.splits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return GCA_EINVAL;
  const size_t ring = (size_t)4 * PF * (2 * tm + 2) * 1024;
  const size_t red = (size_t)2 * tm * g->kd * 16 * 64 * 4;
  const size_t lds = ring > red ? ring : red;
  const dim3 grid((unsigned)nblk);
  if (g->kd == 7) return tm == 2 ? launch_ts<7, 2>(math, grid, lds, st, x, dy, slab, p) : launch_ts<7, 1>(math, grid, lds, st, x, dy, slab, p);
  if (tm == 2) return launch_ts<3, 2>(math, grid, lds, st, x, dy, slab, p);
  return launch_ts<3, 1>(math, grid, lds, st, x, dy, slab, p);
}

}  // namespace gca_conv
