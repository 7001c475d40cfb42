// Interface between conv3d.hip (problem classes, launch configuration, C ABI) and conv3d_halo.hip (the LDS-halo kernels).
#pragma once
#include "conv_igemm_host.h"

namespace gca_conv {

// Kernel parameter block of conv_halo_kernel: the class description of conv3d.hip + the box tiling.
struct HaloParams {
  IgemmParams g;
  int bd, bh, bw;        // box of iteration positions per tile, w fastest; powers of two, bd*bh*bw = tile columns
  int lbh, lbw;          // log2(bh), log2(bw)
  int hd, hh, hw;        // halo box (source positions) of a tile
  int h0d, h0h, h0w;     // halo origin = box origin * m + h0   (h0 = o + smallest tap delta)
  int dmin_d, dmin_h, dmin_w;   // smallest tap delta per dimension
  int nbd, nbh, nbw;     // boxes per clip
  int P;                 // hd*hh*hw valid halo positions (<= halo_max_positions()); LDS slot P is the dump slot
  int nchunks, chunks_per_split;   // 16-channel chunks of the reduction
  int Mrows;             // rows of one packed A step
  unsigned cs_bytes;     // channel stride of the gathered tensor (bytes)
  unsigned pack_bytes;   // extent of this class's packed weights
  // optional input transform (forward, fp32 storage): the conv consumes relu(x * in_scale[c] + in_shift[c]) -- the BatchNorm +
  // ReLU of the producing layer, applied where the halo is staged (once per staged element), so that tensor is never
  // written or read.  Zero padding applies to the TRANSFORMED tensor.  Arrays hold >= 16 * nchunks floats.
  const float* in_scale;
  const float* in_shift;
};

struct HaloCfg {
  int bm, bn;            // tile rows (32..160), columns (128 | 256)
  int bd, bh, bw;
  int splits, chunks_per_split;
  int math;
};

// Kernel parameter block of conv_stem_kernel (conv3d_stem.hip): forward class of a conv with <= 4 input channels, stride 2
// and <= 8 taps along W.
struct StemParams {
  IgemmParams g;
  int wm;                       // wave rows: 2 (tile 64 x 256) or 4 (tile 128 x 128)
  int bd, bh, bw, lbh, lbw;     // box of 256 / 128 output positions per tile, w fastest; powers of two
  int nbd, nbh, nbw;            // boxes per clip
  int hd, hh;                   // halo rows: (bd-1)*sd + kd, (bh-1)*sh + kh
  int wp;                       // elements of one phase-row copy: bw + 4 (even)
  int cd, hrowb;                // LDS layout: dwords between the two copies of a phase row (16 mod 32), bytes of one halo row
  int sd, sh, pd, ph, pw;       // strides in d, h (2 in w), padding
  int kd, kh, kw;
  int nrows, nsteps;            // C*kd*kh reduction rows, two per step
  int Mrows;                    // rows of one packed A step
  int rowoff_bytes;             // LDS bytes of the two tables (reduction-row offsets, halo-row origins; multiple of 16)
  unsigned cs_bytes;            // channel stride of x (bytes)
  unsigned pack_bytes;
  gca_magic m_w2, m_hdh, m_hh;  // divisions of the staging loop: wp, hd*hh, hh
};

// One thread per (step, row, pair of k) of the halo pack (PackParams fmt 1) or of the stem pack (fmt 2: k = 8*half + e, half
// -> reduction row (c, kd-tap, kh-tap) = 2*step + half, e -> kw-tap 0,2,4,6 | 1,3,5,7; taps >= KW and rows past the end are 0).
__device__ __forceinline__ void pack_halo_elements(const float* __restrict__ w, unsigned char* __restrict__ packed,
                                                   const PackParams& p, long long first, long long step, long long end) {
  if (p.fmt == 3) {                              // pointwise fp16 GEMM operand: [Mrows][Kpad] halves, k = reduction channel, contiguous
    const int kp2 = p.Kpad >> 1;                 // (items: one per pair of k)
    for (long long i = first; i < end; i += step) {
      const int m = (int)(i / kp2), k = 2 * (int)(i - (long long)m * kp2);
      float v0 = 0.f, v1 = 0.f;
      if (m < p.M) {
        if (k < p.SC) v0 = w[(long long)k * p.s_ch + (long long)m * p.s_m];
        if (k + 1 < p.SC) v1 = w[(long long)(k + 1) * p.s_ch + (long long)m * p.s_m];
      }
      const unsigned lo = __builtin_bit_cast(unsigned short, (_Float16)v0), hi = __builtin_bit_cast(unsigned short, (_Float16)v1);
      reinterpret_cast<unsigned*>(packed)[i] = lo | (hi << 16);
    }
    return;
  }
  const int rowbytes = p.math == 3 ? 32 : (p.math == 2 ? 96 : 64);
  for (long long i = first; i < end; i += step) {
    const int kp = (int)(i & 7);                 // channel pair inside the chunk
    const long long rowi = i >> 3;
    const int m = (int)(rowi % p.Mrows);
    const int s = (int)(rowi / p.Mrows);
    const int chunk = s / p.ntaps, tl = s - chunk * p.ntaps;
    const int a = tl / (p.nb * p.nc), r = tl - a * (p.nb * p.nc), b = r / p.nc, c = r - b * p.nc;
    const int tap = ((p.k0d + p.sd * a) * p.KH + (p.k0h + p.sh * b)) * p.KW + (p.k0w + p.sw * c);
    float v0 = 0.f, v1 = 0.f;
    const int ch = chunk * 16 + 2 * kp;
    if (p.fmt == 2) {
      const int rho = 2 * s + (kp >> 2), e0 = 2 * (kp & 3);                      // reduction row, first of the two k of this item
      if (m < p.M && rho < p.SC * p.ntaps) {                                     // (ntaps = kd*kh rows per channel for fmt 2)
        const int c = rho / p.ntaps, ab = rho - c * p.ntaps;
        const int t0 = e0 < 4 ? 2 * e0 : 2 * (e0 - 4) + 1, t1 = e0 + 1 < 4 ? 2 * (e0 + 1) : 2 * (e0 + 1 - 4) + 1;
        const float* wr = w + (long long)c * p.s_ch + (long long)ab * p.KW + (long long)m * p.s_m;
        if (t0 < p.KW) v0 = wr[t0];
        if (t1 < p.KW) v1 = wr[t1];
      }
    } else if (m < p.M) {
      if (ch < p.SC) v0 = w[(long long)ch * p.s_ch + tap + (long long)m * p.s_m];
      if (ch + 1 < p.SC) v1 = w[(long long)(ch + 1) * p.s_ch + tap + (long long)m * p.s_m];
    }
    unsigned char* row = packed + rowi * rowbytes;
    if (p.math == 3) {                             // fp16 weights for v_mfma_f32_32x32x16_f16 (fp32 masters stay in the arena)
      const unsigned lo = __builtin_bit_cast(unsigned short, (_Float16)v0), hi = __builtin_bit_cast(unsigned short, (_Float16)v1);
      *reinterpret_cast<unsigned*>(row + 4 * kp) = lo | (hi << 16);
    } else if (p.math == 0) {
      *reinterpret_cast<float2*>(row + 8 * kp) = make_float2(v0, v1);
    } else if (p.math == 2) {
      unsigned h, md, l;
      split_bf16x3(v0, v1, h, md, l);
      *reinterpret_cast<unsigned*>(row + 4 * kp) = h;
      *reinterpret_cast<unsigned*>(row + 32 + 4 * kp) = md;
      *reinterpret_cast<unsigned*>(row + 64 + 4 * kp) = l;
    } else {
      unsigned h, l;
      split_bf16x2(v0, v1, h, l);
      *reinterpret_cast<unsigned*>(row + 4 * kp) = h;
      *reinterpret_cast<unsigned*>(row + 32 + 4 * kp) = l;
    }
  }
}

// Kernel parameter block of conv_pw_f16_kernel (conv3d_pw.hip): pointwise conv on fp16 maps = one GEMM per clip.
struct PwParams {
  int DK, Kc, Kpad;             // output channels, reduction channels, reduction padded to 32 (packed row length)
  int SP, N;                    // positions per clip (multiple of 8), clips
  int tilesM, tiles_sp;         // row tiles of 128, column tiles of 128 per clip
  int P;                        // BatchNorm partials per channel = N * tiles_sp
  int accumulate;
  unsigned src_nstride;         // elements between clips of the source
  unsigned src_bytes, dst_bytes, pack_bytes;
};
int pw_launch(const PwParams& p, const void* src, const unsigned char* apack, const float* bias, void* dst, float* psum,
              float* psq, hipStream_t st);

size_t stem_lds_bytes(const StemParams& sp, int math);
int stem_launch(int math, const StemParams& sp, const void* src, const unsigned char* apack, const float* bias, void* dst,
                float* psum, float* psq, hipStream_t st);

inline int halo_row_bytes(int math) { return math == 3 ? 32 : (math == 2 ? 96 : 64); }
size_t halo_lds_bytes(int bm, int math, int P);
int halo_max_positions();
int halo_launch(const HaloCfg& c, const HaloParams& hp, const void* src, const unsigned char* apack, const int* tapdelta,
                const float* bias, void* dst, float* psum, float* psq, float* slab, hipStream_t st);

}  // namespace gca_conv
