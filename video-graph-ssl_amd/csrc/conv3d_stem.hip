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
//     Hs[(c, zd, zh)][phase][copy][part][wp]   16-bit elements
//
// and the B fragment of a k-step is two ds_read2_b32 per part (E run | O run) at  row(c,a,b) + lane position  -- no
// per-element gathers, no operand split in the loop.  k = 7 (the pad of the odd run) multiplies a zero weight.  Weights are
// packed [step][row][part][16] in the same k order, pre-split (conv_halo.h, fmt 2); two (c,a,b) rows per step (the lane
// halves).  A tile is a box of 128 output positions x 64 output channels; epilogue as in conv3d_halo.hip.
#include <cstring>
#include "conv_igemm_host.h"
#include "conv_halo.h"

using namespace gca_conv;

namespace {

constexpr int rowb(int math) { return math == 3 ? 32 : (math == 2 ? 96 : 64); }     // bytes of one packed 16-k row
constexpr int pitchb(int math) { return rowb(math) + 16; }
constexpr int nparts(int math) { return math == 3 ? 1 : (math == 2 ? 3 : 2); }
__device__ __forceinline__ unsigned short f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }

template <int MATH>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(
    const void* __restrict__ src, const unsigned char* __restrict__ apack, const float* __restrict__ bias,
    void* __restrict__ dst, float* __restrict__ psum, float* __restrict__ psq, const StemParams sp) {
  static_assert(MATH >= 1 && MATH <= 3, "bf16x3 / bf16x6 / fp16 storage");
  constexpr bool F16 = MATH == 3;
  constexpr unsigned ES = F16 ? 2u : 4u;
  constexpr int TM = 2, BM = 64;
  constexpr int ROWB = rowb(MATH), PITCH = pitchb(MATH), NP = nparts(MATH);
  constexpr int RPC = ROWB / 16, NPC = BM * RPC, A_PC = (NPC + 255) / 256;
  const IgemmParams& p = sp.g;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* const rowoff = reinterpret_cast<int*>(smem);                       // [2 * nsteps] LDS byte offset of halo row (c, a, b)
  unsigned char* const As = smem + sp.rowoff_bytes;                       // [2][BM][PITCH]
  unsigned char* const Hs = As + 2 * BM * PITCH;                          // halo rows

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, lh = lane >> 5, ll = lane & 31;
  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);
  const int tileM = bid % p.tilesM, tileN = bid / p.tilesM;
  const int per_img = sp.nbd * sp.nbh * sp.nbw;
  const int img = tileN / per_img;
  int tb = tileN - img * per_img;
  const int tbd = tb / (sp.nbh * sp.nbw); tb -= tbd * (sp.nbh * sp.nbw);
  const int tbh = tb / sp.nbw, tbw = tb - tbh * sp.nbw;
  const int q0d = tbd * sp.bd, q0h = tbh * sp.bh, q0w = tbw * sp.bw;

  const int wp = sp.wp;                                  // elements of one phase row copy (even)
  const unsigned PARTB = (unsigned)wp * 2u;              // bytes: one part of one copy
  const unsigned COPYB = PARTB * NP, PHASEB = 2u * COPYB, HROWB = 2u * PHASEB;
  const int khd = sp.kd * sp.kh;
  for (int r = tid; r < 2 * sp.nsteps; r += 256) {
    int off = 0;
    if (r < sp.nrows) {
      const int c = r / khd, ab = r - c * khd, a = ab / sp.kh, b = ab - a * sp.kh;
      off = ((c * sp.hd + a) * sp.hh + b) * (int)HROWB;
    }
    rowoff[r] = off;                                     // rows past the end multiply zero weights: any finite data will do
  }

  // ---- A pieces of this thread (as conv3d_halo.hip)
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(apack), 0, sp.pack_bytes, 0x00020000);
  unsigned avoff[A_PC], awoff[A_PC];
#pragma unroll
  for (int u = 0; u < A_PC; ++u) {
    const int i = tid + 256 * u;
    const int row = i / RPC, sub = i - row * RPC;
    avoff[u] = i < NPC ? (unsigned)(tileM * BM) * ROWB + (unsigned)i * 16u : 0xffffffffu;
    awoff[u] = (unsigned)(i < NPC ? row : 0) * PITCH + (unsigned)sub * 16u;
  }
  const unsigned astep = (unsigned)sp.Mrows * ROWB;
  uint4 areg[A_PC];
  auto a_issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A_PC; ++u)
      areg[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)avoff[u], (int)((unsigned)s * astep), 0));
  };
  auto a_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A_PC; ++u)
      if (u + 1 < A_PC || NPC % 256 == 0 || tid + 256 * u < NPC)
        *reinterpret_cast<uint4*>(As + buf * (BM * PITCH) + awoff[u]) = areg[u];
  };
  a_issue(0);

  // ---- stage the halo: task = (halo row hr = (c, zd, zh), u = column offset from the tile's first input column); element
  // x[c][s0d + zd][s0h + zh][s0w + u] goes to phase u & 1, index u >> 1 of copy 0 and index (u >> 1) - 1 of copy 1
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, p.src_bytes, 0x00020000);
    const int W2 = 2 * wp;
    const int nhrows = p.SC * sp.hd * sp.hh;
    const int ntasks = nhrows * W2;
    const int s0d = q0d * sp.sd - sp.pd, s0h = q0h * sp.sh - sp.ph, s0w = q0w * 2 - sp.pw;
    const unsigned ibase = (unsigned)((long long)img * p.src_nstride);
    const int hdh = sp.hd * sp.hh;
    constexpr int UF = 8;
    for (int t0 = tid; t0 < ntasks; t0 += 256 * UF) {
      float v[UF];
      int hrw[UF], uu[UF];
#pragma unroll
      for (int k = 0; k < UF; ++k) {
        const int task = t0 + 256 * k;
        const int hr = (int)gca_fdiv((unsigned)task, sp.m_w2), u = task - hr * W2;
        const int c = (int)gca_fdiv((unsigned)hr, sp.m_hdh), rem = hr - c * hdh;
        const int zd = (int)gca_fdiv((unsigned)rem, sp.m_hh), zh = rem - zd * sp.hh;
        const int id = s0d + zd, ih = s0h + zh, iw = s0w + u;
        const bool ok = task < ntasks && (unsigned)id < (unsigned)p.SD && (unsigned)ih < (unsigned)p.SH && (unsigned)iw < (unsigned)p.SW;
        const unsigned e = ibase + (unsigned)((id * p.SH + ih) * p.SW + iw);
        const unsigned vo = ok ? e * ES + (unsigned)c * sp.cs_bytes : 0xffffffffu;      // (the channel differs per lane: not an soffset)
        if (F16) v[k] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)vo, 0, 0));
        else v[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)vo, 0, 0));
        hrw[k] = task < ntasks ? hr : -1;
        uu[k] = u;
      }
#pragma unroll
      for (int k = 0; k < UF; ++k) {
        if (hrw[k] < 0) continue;
        unsigned short part[3];
        if (F16) part[0] = (unsigned short)__float_as_uint(v[k]);
        else if (MATH == 2) {
          unsigned h, m, l;
          split_bf16x3(v[k], 0.f, h, m, l);
          part[0] = (unsigned short)h; part[1] = (unsigned short)m; part[2] = (unsigned short)l;
        } else {
          unsigned h, l;
          split_bf16x2(v[k], 0.f, h, l);
          part[0] = (unsigned short)h; part[1] = (unsigned short)l;
        }
        const int ph = uu[k] & 1, idx = uu[k] >> 1;
        unsigned char* rowp = Hs + (unsigned)hrw[k] * HROWB + (unsigned)ph * PHASEB;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          *reinterpret_cast<unsigned short*>(rowp + q * PARTB + idx * 2) = part[q];
          if (idx > 0) *reinterpret_cast<unsigned short*>(rowp + COPYB + q * PARTB + (idx - 1) * 2) = part[q];
        }
      }
    }
    // the last element of copy 1 (index wp - 1) has no source: it is only ever multiplied by the zero pad weight, but must be finite
    for (int r = tid; r < nhrows * 2 * NP; r += 256) {
      const int hr = r / (2 * NP), rem = r - hr * (2 * NP), ph = rem / NP, q = rem - ph * NP;
      *reinterpret_cast<unsigned short*>(Hs + (unsigned)hr * HROWB + (unsigned)ph * PHASEB + COPYB + q * PARTB + (wp - 1) * 2) = 0;
    }
  }

  // ---- per-lane column state
  const int cidx = wn * 32 + ll;
  const int zw = cidx & (sp.bw - 1), t1 = cidx >> sp.lbw;
  const int zh = t1 & (sp.bh - 1), zd = t1 >> sp.lbh;
  const int cq0 = q0d + zd, cq1 = q0h + zh, cq2 = q0w + zw;
  const bool cval = cq0 < p.QD && cq1 < p.QH && cq2 < p.QW;
  const unsigned bpos = (unsigned)((zd * sp.sd) * sp.hh + zh * sp.sh) * HROWB + (unsigned)(zw & 1) * COPYB + (unsigned)(zw & ~1) * 2u;
  const unsigned aoff = (unsigned)ll * PITCH + (unsigned)lh * 16u;

  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  a_store(0);
  __syncthreads();                                       // publishes rowoff, the halo and A tile 0
  a_issue(min(1, sp.nsteps - 1));
  int buf = 0;
  unsigned ro = (unsigned)rowoff[lh];
  for (int s = 0; s < sp.nsteps; ++s) {
    const unsigned ro_next = (unsigned)rowoff[2 * min(s + 1, sp.nsteps - 1) + lh];
    float4 bf[NP], af[TM][NP];
    const unsigned char* bp = Hs + bpos + ro;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      // 4-byte aligned runs: two dwords each (ds_read2_b32), never one 8-byte access
      const unsigned* e = reinterpret_cast<const unsigned*>(bp + q * PARTB);            // even-phase run: taps 0, 2, 4, 6
      const unsigned* o = reinterpret_cast<const unsigned*>(bp + PHASEB + q * PARTB);   // odd-phase run:  taps 1, 3, 5, pad
      bf[q] = make_float4(__uint_as_float(e[0]), __uint_as_float(e[1]), __uint_as_float(o[0]), __uint_as_float(o[1]));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < NP; ++q)
        af[i][q] = *reinterpret_cast<const float4*>(As + buf * (BM * PITCH) + aoff + i * (32 * PITCH) + 32 * q);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (MATH == 3) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, bf[0]), acc[i], 0, 0, 0);
      } else {
        const bf16x8 xh = __builtin_bit_cast(bf16x8, af[i][0]), xl = __builtin_bit_cast(bf16x8, af[i][NP - 1]);
        const bf16x8 yh = __builtin_bit_cast(bf16x8, bf[0]), yl = __builtin_bit_cast(bf16x8, bf[NP - 1]);
        if (MATH == 2) {
          const bf16x8 xm = __builtin_bit_cast(bf16x8, af[i][1]), ym = __builtin_bit_cast(bf16x8, bf[1]);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, ym, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, yh, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ym, acc[i], 0, 0, 0);
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc[i], 0, 0, 0);
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc[i], 0, 0, 0);
      }
    }
    a_store(buf ^ 1);                                    // tile s+1 -> the other buffer; fetch s+2 (clamped: exact vmcnt waits)
    a_issue(min(s + 2, sp.nsteps - 1));
    __syncthreads();
    buf ^= 1;
    ro = ro_next;
  }

  // ---- epilogue (C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)); see conv3d.hip
  const int mbase = tileM * BM;
  const int rows_left = p.DK - mbase - 4 * lh;
  const int DHW = p.DH * p.DW;
  const unsigned DSP = (unsigned)(p.DD * DHW);
  const unsigned rowb_ = DSP * ES;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, p.dst_bytes, 0x00020000);
  {
    const unsigned dsp = (unsigned)(cq0 * DHW + cq1 * p.DW + cq2);
    const unsigned vb = cval ? (((unsigned)img * (unsigned)p.DK + (unsigned)(mbase + 4 * lh)) * DSP + dsp) * ES : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
        const unsigned vo = rr < rows_left ? vb : 0xffffffffu;
        float v = acc[i][r];
        if (bias) v += bias[min(mbase + rr + 4 * lh, p.DK - 1)];
        if (F16) __builtin_amdgcn_raw_buffer_store_b16(f16_bits(v), rd, (int)vo, (int)((unsigned)rr * rowb_), 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rd, (int)vo, (int)((unsigned)rr * rowb_), 0);
      }
  }
  if (psum) {
    float* red = reinterpret_cast<float*>(As);                   // [4][BM][2] floats; the operand tiles are dead by now
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = cval ? acc[i][r] : 0.f;
        const float sm = half_wave_sum_hi(v), sq = half_wave_sum_hi(v * v);
        const int rr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ll == 31) { red[(wn * BM + rr) * 2] = sm; red[(wn * BM + rr) * 2 + 1] = sq; }
      }
    }
    __syncthreads();
    if (tid < BM && mbase + tid < p.DK) {
      const float sm = red[tid * 2] + red[(BM + tid) * 2] + red[(2 * BM + tid) * 2] + red[(3 * BM + tid) * 2];
      const float sq = red[tid * 2 + 1] + red[(BM + tid) * 2 + 1] + red[(2 * BM + tid) * 2 + 1] + red[(3 * BM + tid) * 2 + 1];
      const long long m = mbase + tid;
      psum[m * p.P + tileN] = sm;
      psq[m * p.P + tileN] = sq;
    }
  }
}

}  // namespace

namespace gca_conv {

size_t stem_lds_bytes(const StemParams& sp, int math) {
  const size_t hrow = (size_t)4 * nparts(math) * sp.wp * 2;
  return (size_t)sp.rowoff_bytes + 2 * 64 * pitchb(math) + (size_t)sp.g.SC * sp.hd * sp.hh * hrow;
}

int stem_launch(int math, const StemParams& sp, const void* src, const unsigned char* apack, const float* bias, void* dst,
                float* psum, float* psq, hipStream_t st) {
  const size_t lds = stem_lds_bytes(sp, math);
  const long long nblk = (long long)sp.g.tilesM * sp.g.tilesN;
  if (nblk <= 0 || nblk > 0x7fffffffLL || lds > (size_t)(160 << 10) - 1024) return GCA_EINVAL;
  static bool raised[4] = {false, false, false, false};
#define GCA_SK(M)                                                                                                          \
  {                                                                                                                        \
    if (lds > (48u << 10) && !raised[M]) {                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem_kernel<M>),                                         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10) != hipSuccess) return GCA_ELAUNCH;   \
      raised[M] = true;                                                                                                    \
    }                                                                                                                      \
    hipLaunchKernelGGL((conv_stem_kernel<M>), dim3((unsigned)nblk), dim3(256), lds, st, src, apack, bias, dst, psum, psq, sp); \
  }
  switch (math) {
    case 1: GCA_SK(1) break;
    case 2: GCA_SK(2) break;
    case 3: GCA_SK(3) break;
    default: return GCA_EINVAL;
  }
#undef GCA_SK
  return gca_launch_status();
}

}  // namespace gca_conv
