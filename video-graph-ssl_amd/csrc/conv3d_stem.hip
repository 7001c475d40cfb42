// Stem convolutions (3 input channels, stride 2 along W, <= 8 taps along W) for gfx950: forward only.
//
// The first conv of every encoder on the path -- R(2+1)D conv1_s (1,7,7)/(1,2,2) (resnet2p1d.py:162-168), S3D's SepConv3d
// spatial half (1,7,7)/(1,2,2) (s3d_1.py:8,35-38), 3D-ResNet conv1 (7,7,7)/(1,2,2) (resnet.py:120-126) -- has C = 3: the
// LDS-halo kernels need 16-channel chunks, and the gather kernels fetch every one of its C*taps = 147..1029 reduction
// elements per output column individually (36 TF/s on the R(2+1)D stem, 207 TF/s in fp16 on the 3D-ResNet stem).
//
// Here the reduction index is (c, kd-tap a, kh-tap b | kw-tap t): one MFMA k-HALF (8 k of the 32x32x16 shapes) is the W
// window of one (c, a, b).  With stride 2 the window of output column w is x[2w - pw + t], t = 0..kw-1: the even taps are
// 4 CONSECUTIVE elements of the even-phase row E[i] = x[2i + base], the odd taps 3-4 consecutive elements of the odd-phase
// row O[i] = x[2i + 1 + base], both starting at i = w - w0.  The tile's input halo is therefore staged in LDS ONCE, split by
// W phase (and into bf16 parts), each phase row twice -- as is and shifted by one element -- so that every lane's 4-element
// run starts on a 4-byte boundary whatever the parity of its column:
//
//     Hs[(c, zd, zh)][phase][part][copy][wp]   16-bit elements
//
// and the B fragment of a k-step is two ds_read2_b32 per part (E run | O run) at  row(c,a,b) + lane position  -- no
// per-element gathers, no operand split in the loop.  k = 7 (the pad of the odd run) multiplies a zero weight.  Weights are
// packed [step][row][part][16] in the same k order, pre-split (conv_halo.h, fmt 2); two (c,a,b) rows per step (the lane
// halves); a wave owns one 32-row tile and reads its A fragments straight from the L2-resident pack, one step ahead -- no LDS
// copy of the weights and no barrier in the k loop.  A tile is a box of 256 output positions x 64 output channels (waves
// 2 x 2) or 128 positions x 128 channels (4 x 1, more than 64 output channels); epilogue as in conv3d_halo.hip.
#include <cstring>
#include "conv_igemm_host.h"
#include "conv_halo.h"

using namespace gca_conv;

namespace {

constexpr int rowb(int math) { return math == 3 ? 32 : (math == 2 ? 96 : 64); }     // bytes of one packed 16-k row
constexpr int nparts(int math) { return math == 3 ? 1 : (math == 2 ? 3 : 2); }
__device__ __forceinline__ unsigned short f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }

template <int MATH, int WM>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(
    const void* __restrict__ src, const unsigned char* __restrict__ apack, const float* __restrict__ bias,
    void* __restrict__ dst, float* __restrict__ psum, float* __restrict__ psq, const StemParams sp) {
  static_assert(MATH >= 1 && MATH <= 3, "bf16x3 / bf16x6 / fp16 storage");
  constexpr bool F16 = MATH == 3;
  constexpr unsigned ES = F16 ? 2u : 4u;
  // waves WM (rows) x WN (columns), 32 rows x 128 columns each: 64 x 256 (output channels <= 64) or 128 x 128 per workgroup
  static_assert(WM == 2 || WM == 4, "wave grid");
  constexpr int WN = 4 / WM, BM = 32 * WM, TN = 4;
  constexpr int ROWB = rowb(MATH), NP = nparts(MATH);
  const IgemmParams& p = sp.g;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* const rowoff = reinterpret_cast<int*>(smem);                       // [2 * nsteps] LDS byte offset of halo row (c, a, b)
  unsigned* const rowbase = reinterpret_cast<unsigned*>(smem) + 2 * sp.nsteps;   // [C*hd*hh] byte offset of the input row in x (or ~0)
  unsigned char* const Hs = smem + sp.rowoff_bytes;                       // halo rows (>= 2 KB: also the statistics scratch of the epilogue)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN, lh = lane >> 5, ll = lane & 31;
  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);
  const int tileM = bid % p.tilesM, tileN = bid / p.tilesM;
  const int per_img = sp.nbd * sp.nbh * sp.nbw;
  const int img = tileN / per_img;
  int tb = tileN - img * per_img;
  const int tbd = tb / (sp.nbh * sp.nbw); tb -= tbd * (sp.nbh * sp.nbw);
  const int tbh = tb / sp.nbw, tbw = tb - tbh * sp.nbw;
  const int q0d = tbd * sp.bd, q0h = tbh * sp.bh, q0w = tbw * sp.bw;

  const int wp = sp.wp;                                  // elements of one phase row copy (even)
  // halo row = [phase][part][copy][cd dwords]; copy stride and row pitch come from the host (stem_copy_dwords / stem_row_bytes)
  const unsigned COPYB = (unsigned)sp.cd * 4u, PARTB = 2u * COPYB, PHASEB = PARTB * NP, HROWB = (unsigned)sp.hrowb;
  const int khd = sp.kd * sp.kh;
  // (m_w2 divides by wp here: the dword tasks of one halo row)
  for (int r = tid; r < 2 * sp.nsteps; r += 256) {
    int off = 0;
    if (r < sp.nrows) {
      const int c = r / khd, ab = r - c * khd, a = ab / sp.kh, b = ab - a * sp.kh;
      off = ((c * sp.hd + a) * sp.hh + b) * (int)HROWB;
    }
    rowoff[r] = off;                                     // rows past the end multiply zero weights: any finite data will do
  }

  // ---- A fragments come STRAIGHT from the packed weights (L1/L2-resident): a wave owns ONE 32-row tile (the two wave rows
  // split M, so the weights cross the L1 once per wave pair instead of once per wave) and reads its 16 bytes per part at
  // step*astep + (tileM*BM + 32 wm + ll)*ROWB + 32 part + 16 lh.  One step ahead in registers; no LDS copy, no barrier in the k loop.
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(apack), 0, sp.pack_bytes, 0x00020000);
  const unsigned avoff = (unsigned)(tileM * BM + 32 * wm + ll) * ROWB + (unsigned)lh * 16u;
  const unsigned astep = (unsigned)sp.Mrows * ROWB;
  float4 afn[NP];
  auto a_issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(avoff + 32u * q), (int)((unsigned)s * astep), 0));
      afn[q] = make_float4(f.x, f.y, f.z, f.w);
    }
  };
  a_issue(0);

  // ---- stage the halo.  A task is one DWORD (two consecutive elements) of a phase row in both copies: it loads S[2j], S[2j+1],
  // S[2j+2] of its phase (input columns s0w + 2(2j + e) + phase), and writes the pair (S[2j], S[2j+1]) to copy 0 and the pair
  // (S[2j+1], S[2j+2]) to copy 1 -- the bf16 split works on pairs anyway, the LDS stores are conflict-free b32s, and the only
  // division per task is the one by the row length (row origins come from a table built once per tile).
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, p.src_bytes, 0x00020000);
    const int nhrows = p.SC * sp.hd * sp.hh;
    const int s0d = q0d * sp.sd - sp.pd, s0h = q0h * sp.sh - sp.ph, s0w = q0w * 2 - sp.pw;
    const unsigned ibase = (unsigned)((long long)img * p.src_nstride);
    const int hdh = sp.hd * sp.hh;
    for (int hr = tid; hr < nhrows; hr += 256) {
      const int c = (int)gca_fdiv((unsigned)hr, sp.m_hdh), rem = hr - c * hdh;
      const int zd = (int)gca_fdiv((unsigned)rem, sp.m_hh), zh = rem - zd * sp.hh;
      const int id = s0d + zd, ih = s0h + zh;
      const bool ok = (unsigned)id < (unsigned)p.SD && (unsigned)ih < (unsigned)p.SH;
      rowbase[hr] = ok ? (ibase + (unsigned)((id * p.SH + ih) * p.SW)) * ES + (unsigned)c * sp.cs_bytes : 0xffffffffu;
    }
    __syncthreads();
    const int hw = wp >> 1;                               // dwords per phase-row copy
    const int ntasks = nhrows * wp;
    constexpr int UF = 4;
    for (int t0 = tid; t0 < ntasks; t0 += 256 * UF) {
      float v[UF][3];
      unsigned wo[UF];
#pragma unroll
      for (int k = 0; k < UF; ++k) {
        const int task = t0 + 256 * k;
        const int hr = (int)gca_fdiv((unsigned)task, sp.m_w2), slot = task - hr * wp;
        const int ph = slot >= hw ? 1 : 0, j = slot - ph * hw;
        const unsigned rb = task < ntasks ? rowbase[hr] : 0xffffffffu;
        const int iw0 = s0w + 4 * j + ph;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const int iw = iw0 + 2 * e;
          const unsigned vo = (rb == 0xffffffffu || (unsigned)iw >= (unsigned)p.SW) ? 0xffffffffu : rb + (unsigned)iw * ES;
          if (F16) v[k][e] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)vo, 0, 0));
          else v[k][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)vo, 0, 0));
        }
        wo[k] = task < ntasks ? (unsigned)hr * HROWB + (unsigned)ph * PHASEB + (unsigned)j * 4u : 0xffffffffu;
      }
#pragma unroll
      for (int k = 0; k < UF; ++k) {
        if (wo[k] == 0xffffffffu) continue;
        unsigned char* d = Hs + wo[k];
        if (F16) {
          const unsigned a = __float_as_uint(v[k][0]), b = __float_as_uint(v[k][1]), c = __float_as_uint(v[k][2]);
          *reinterpret_cast<unsigned*>(d) = a | (b << 16);
          *reinterpret_cast<unsigned*>(d + COPYB) = b | (c << 16);
        } else if (MATH == 2) {
          unsigned h, m, l;
          split_bf16x3(v[k][0], v[k][1], h, m, l);
          *reinterpret_cast<unsigned*>(d) = h; *reinterpret_cast<unsigned*>(d + PARTB) = m; *reinterpret_cast<unsigned*>(d + 2 * PARTB) = l;
          split_bf16x3(v[k][1], v[k][2], h, m, l);
          *reinterpret_cast<unsigned*>(d + COPYB) = h; *reinterpret_cast<unsigned*>(d + COPYB + PARTB) = m;
          *reinterpret_cast<unsigned*>(d + COPYB + 2 * PARTB) = l;
        } else {
          unsigned h, l;
          split_bf16x2(v[k][0], v[k][1], h, l);
          *reinterpret_cast<unsigned*>(d) = h; *reinterpret_cast<unsigned*>(d + PARTB) = l;
          split_bf16x2(v[k][1], v[k][2], h, l);
          *reinterpret_cast<unsigned*>(d + COPYB) = h; *reinterpret_cast<unsigned*>(d + COPYB + PARTB) = l;
        }
      }
    }
  }

  // ---- per-lane column state: column tile j of this wave = columns wn*128 + 32 j + ll of the box
  unsigned bpos[TN];
  bool cval[TN];
  int cq[TN][3];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cidx = wn * 128 + j * 32 + ll;
    const int zw = cidx & (sp.bw - 1), t1 = cidx >> sp.lbw;
    const int zh = t1 & (sp.bh - 1), zd = t1 >> sp.lbh;
    cq[j][0] = q0d + zd; cq[j][1] = q0h + zh; cq[j][2] = q0w + zw;
    cval[j] = cq[j][0] < p.QD && cq[j][1] < p.QH && cq[j][2] < p.QW;
    bpos[j] = (unsigned)((zd * sp.sd) * sp.hh + zh * sp.sh) * HROWB + (unsigned)(zw & 1) * COPYB + (unsigned)(zw & ~1) * 2u;
  }

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  auto load_b = [&](float4 (&bf)[NP], int j, unsigned ro) __attribute__((always_inline)) {
    const unsigned char* bp = Hs + bpos[j] + ro;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      // 4-byte aligned runs: two dwords each (ds_read2_b32), never one 8-byte access
      const unsigned* e = reinterpret_cast<const unsigned*>(bp + q * PARTB);            // even-phase run: taps 0, 2, 4, 6
      const unsigned* o = reinterpret_cast<const unsigned*>(bp + PHASEB + q * PARTB);   // odd-phase run:  taps 1, 3, 5, pad
      bf[q] = make_float4(__uint_as_float(e[0]), __uint_as_float(e[1]), __uint_as_float(o[0]), __uint_as_float(o[1]));
    }
  };
  auto mma = [&](f32x16& c, const float4 (&af)[NP], const float4 (&bf)[NP]) __attribute__((always_inline)) {
    if constexpr (MATH == 3) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[0]), c, 0, 0, 0);
    } else {
      const bf16x8 xh = __builtin_bit_cast(bf16x8, af[0]), xl = __builtin_bit_cast(bf16x8, af[NP - 1]);
      const bf16x8 yh = __builtin_bit_cast(bf16x8, bf[0]), yl = __builtin_bit_cast(bf16x8, bf[NP - 1]);
      if (MATH == 2) {
        const bf16x8 xm = __builtin_bit_cast(bf16x8, af[1]), ym = __builtin_bit_cast(bf16x8, bf[1]);
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

  __syncthreads();                                       // publishes rowoff and the halo (read-only from here on)
  unsigned ro = (unsigned)rowoff[lh];
  float4 bfc[NP], bfn[NP];
  load_b(bfc, 0, ro);
  for (int s = 0; s < sp.nsteps; ++s) {
    const unsigned ro_next = (unsigned)rowoff[2 * min(s + 1, sp.nsteps - 1) + lh];
    float4 af[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) af[q] = afn[q];
    a_issue(min(s + 1, sp.nsteps - 1));                  // next step's weights (clamped: unconditional loads, exact waits)
#pragma unroll
    for (int j = 0; j < TN; ++j) {                       // the next column tile's fragments (or the next step's first) behind these MFMAs
      if (j + 1 < TN) load_b(bfn, j + 1, ro); else load_b(bfn, 0, ro_next);
      mma(acc[j], af, bfc);
#pragma unroll
      for (int q = 0; q < NP; ++q) bfc[q] = bfn[q];
    }
    ro = ro_next;
  }
  __syncthreads();                                       // (the epilogue's statistics scratch overlays the halo)

  // ---- epilogue (C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)); see conv3d.hip
  const int mbase = tileM * BM + 32 * wm;                  // first row of this wave
  const int rows_left = p.DK - mbase - 4 * lh;
  const int DHW = p.DH * p.DW;
  const unsigned DSP = (unsigned)(p.DD * DHW);
  const unsigned rowb_ = DSP * ES;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, p.dst_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const unsigned dsp = (unsigned)(cq[j][0] * DHW + cq[j][1] * p.DW + cq[j][2]);
    const unsigned vb = cval[j] ? (((unsigned)img * (unsigned)p.DK + (unsigned)(mbase + 4 * lh)) * DSP + dsp) * ES : 0xffffffffu;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2);
      const unsigned vo = rr < rows_left ? vb : 0xffffffffu;
      float v = acc[j][r];
      if (bias) v += bias[min(mbase + rr + 4 * lh, p.DK - 1)];
      if (F16) __builtin_amdgcn_raw_buffer_store_b16(f16_bits(v), rd, (int)vo, (int)((unsigned)rr * rowb_), 0);
      else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rd, (int)vo, (int)((unsigned)rr * rowb_), 0);
    }
  }
  if (psum) {
    float* red = reinterpret_cast<float*>(Hs);                   // [WN][BM][2] floats; the halo is dead by now
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) { const float v = cval[j] ? acc[j][r] : 0.f; sm += v; sq += v * v; }
      sm = half_wave_sum_hi(sm);
      sq = half_wave_sum_hi(sq);
      const int rr = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ll == 31) { red[(wn * BM + rr) * 2] = sm; red[(wn * BM + rr) * 2 + 1] = sq; }
    }
    __syncthreads();
    if (tid < BM && tileM * BM + tid < p.DK) {
      float sm = red[tid * 2], sq = red[tid * 2 + 1];
      if (WN == 2) { sm += red[(BM + tid) * 2]; sq += red[(BM + tid) * 2 + 1]; }
      const long long m = tileM * BM + tid;
      psum[m * p.P + tileN] = sm;
      psq[m * p.P + tileN] = sq;
    }
  }
}

}  // namespace

namespace gca_conv {

size_t stem_lds_bytes(const StemParams& sp, int math) {
  const size_t halo = (size_t)sp.g.SC * sp.hd * sp.hh * sp.hrowb;
  return (size_t)sp.rowoff_bytes + (halo > 2048 ? halo : 2048);
}

int stem_launch(int math, const StemParams& sp, const void* src, const unsigned char* apack, const float* bias, void* dst,
                float* psum, float* psq, hipStream_t st) {
  const size_t lds = stem_lds_bytes(sp, math);
  const long long nblk = (long long)sp.g.tilesM * sp.g.tilesN;
  if (nblk <= 0 || nblk > 0x7fffffffLL || lds > (size_t)(160 << 10) - 1024) return GCA_EINVAL;
  static bool raised[4][2] = {};
#define GCA_SK2(M, W)                                                                                                      \
  {                                                                                                                        \
    if (lds > (48u << 10) && !raised[M][W / 4]) {                                                                          \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem_kernel<M, W>),                                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10) != hipSuccess) return GCA_ELAUNCH;   \
      raised[M][W / 4] = true;                                                                                             \
    }                                                                                                                      \
    hipLaunchKernelGGL((conv_stem_kernel<M, W>), dim3((unsigned)nblk), dim3(256), lds, st, src, apack, bias, dst, psum, psq, sp); \
  }
#define GCA_SK(M) { if (sp.wm == 4) GCA_SK2(M, 4) else GCA_SK2(M, 2) }
  switch (math) {
    case 1: GCA_SK(1) break;
    case 2: GCA_SK(2) break;
    case 3: GCA_SK(3) break;
    default: return GCA_EINVAL;
  }
#undef GCA_SK
#undef GCA_SK2
  return gca_launch_status();
}

}  // namespace gca_conv
