// Pointwise (1x1x1, unit stride, no padding) convolutions on fp16 feature maps for gfx950: forward and dgrad.
//
// Two thirds of the convs of a bottleneck 3D-ResNet (resnet.py:45-63: conv1 / conv3 of every block, BASELINE configs[4]) are
// plain GEMMs per clip,  y[m][sp] = sum_c A[m][c] * x[c][sp]  (dgrad: A = W^T), with few channels (64..2048) against
// 10^4..10^6 positions: HBM-bound at 50..100 FLOP per byte.  The gather kernels treat them as general convs (position
// decode, tap masks, one 2-byte load per B element, 2-byte stores) and reach ~0.2 of the HBM roofline on them.  Here:
//
//   * the x tile [32 channels][128 positions] is COPIED into LDS with 16-byte loads / stores, rows = channels as they lie in
//     memory; the MFMA B operand (k = channel, strided in that image) is read with ds_read_b64_tr_b16, CDNA4's transposing
//     LDS read (cdna guide T10: 4 rows x 16 columns per 16-lane group, delivered column-major), through the XOR swizzle of
//     that recipe for 256-byte rows, so stores and transposed reads are both conflict-free;
//   * A = the weights packed fp16 [rows][channels] (pack fmt 3, k contiguous): 16-byte copies, ds_read_b128 fragments;
//   * v_mfma_f32_32x32x16_f16, 128 x 128 tile, waves 2 x 2 (64 x 64 each), double-buffered LDS, one barrier per 32 channels;
//   * epilogue as in the other conv kernels (buffer stores, accumulate, BatchNorm partial sums).  (Measured and not kept: the
//     output tile staged through LDS and written as 16-byte row segments -- no faster, 0.210 vs 0.217 ms on the 64 -> 256
//     layer-1 conv, slower on the small maps: with 2..64 k-tiles per workgroup the kernel is bound by the latency of its
//     one-tile-ahead operand fetch, not by its stores; fetching TWO tiles ahead through a second register set costs a
//     resident workgroup (193 VGPRs) and was slower still, 0.240 ms;
//     round 3: both operand tiles by LDS-DMA (no registers in flight) into a ring of four 16 KB stages, three k-tiles ahead,
//     waited for by instruction count, A rows unpadded with the bank swizzle on the source address -- correct, and 4 % slower
//     in a same-box A/B of the whole fp16 step (0.0675 vs 0.0645 ms per launch): the latency that binds here is not the
//     operand fetch's alone.)
#include <cstring>
#include "conv_igemm_host.h"
#include "conv_halo.h"

using namespace gca_conv;

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int BM = 128, BN = 128, BKC = 32;          // tile rows (output channels), columns (positions), channels per k-tile
constexpr int A_PITCH = 80;                          // bytes per A row in LDS: 64 + 16 (odd number of 16-B slots)
__device__ __forceinline__ unsigned short f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
__device__ __forceinline__ fp16x4_t tr_read(const unsigned char* lds) {       // ds_read_b64_tr_b16 (8-byte aligned LDS address)
  return __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(lds));
}
// byte offset of 16-byte chunk `ch` (0..15) of row `row` in the [32][256 B] image (cdna guide T10, layout (b))
__device__ __forceinline__ unsigned boff(unsigned row, unsigned ch) { return 256u * row + 16u * (ch ^ (((row & 3u) << 2) | ((row >> 2) & 3u))); }

__global__ __launch_bounds__(256, 2) void conv_pw_f16_kernel(
    const void* __restrict__ src, const unsigned char* __restrict__ apack, const float* __restrict__ bias,
    void* __restrict__ dst, float* __restrict__ psum, float* __restrict__ psq, const PwParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char As[2][BM * A_PITCH];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][BKC * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, lh = lane >> 5, ll = lane & 31;
  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);
  const int tileM = bid % p.tilesM, tileN = bid / p.tilesM;
  const int img = tileN / p.tiles_sp, tsp = tileN - img * p.tiles_sp;
  const int sp0 = tsp * BN;

  // ---- global -> register pieces of this thread: two 16-byte chunks of B (rows r, r+16), two of A (rows r, r+64)
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(apack), 0, p.pack_bytes, 0x00020000);
  const int brow = tid >> 4, bch = tid & 15;                               // channel row 0..15 (+16), chunk 0..15
  const bool bcol_ok = sp0 + 8 * bch < p.SP;                               // (SP % 8 == 0: a chunk is all inside or all outside)
  const unsigned rowbytes = (unsigned)p.SP * 2u;
  // per-lane part of the B address (clip, channel row inside the k-tile, position chunk); the k-tile's first channel is the
  // wave-uniform soffset
  const unsigned bvoff = bcol_ok ? ((unsigned)img * p.src_nstride + (unsigned)(sp0 + 8 * bch)) * 2u + (unsigned)brow * rowbytes : 0xffffffffu;
  const int arow = tid >> 2, ach = tid & 3;
  const unsigned avoff = ((unsigned)(tileM * BM + arow) * (unsigned)p.Kpad + 8u * ach) * 2u;
  const unsigned astride64 = 64u * (unsigned)p.Kpad * 2u;
  uint4 breg[2], areg[2];
  auto issue = [&](int kt) __attribute__((always_inline)) {
    const int c0 = kt * BKC;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned v = c0 + brow + 16 * i < p.Kc ? bvoff : 0xffffffffu;      // channels past the end: zeros
      breg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)v, (int)((unsigned)(c0 + 16 * i) * rowbytes), 0));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      areg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(avoff + (unsigned)i * astride64), c0 * 2, 0));
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(&Bs[buf][boff((unsigned)(brow + 16 * i), (unsigned)bch)]) = breg[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(&As[buf][(arow + 64 * i) * A_PITCH + 16 * ach]) = areg[i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addresses.  B: 16-lane group g = lane >> 4 reads the 4-row x 16-column block at rows 16 ks + 8 h + 4 rd,
  // columns 64 wn + 32 j + 16 (g & 1): lane 4q + p of the group supplies row q, chunk (p >> 1), byte 8 (p & 1) of it
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, p4 = l16 & 3;
  unsigned btr[2][2];                                                      // [j][rd] for k-step 0 of a buffer; k-step 1 = rows + 16
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const unsigned row = (unsigned)(8 * (g >> 1) + 4 * rd + q4);
      const unsigned ch = (unsigned)((64 * wn + 32 * j) / 8 + 2 * (g & 1) + (p4 >> 1));
      btr[j][rd] = boff(row, ch) + 8u * (p4 & 1);
    }
  // rows 16..31 of the image: (row & 3) and ((row >> 2) & 3) are those of row - 16, so the swizzle is the same: + 16 * 256 bytes
  const unsigned aoff = (unsigned)(64 * wm + ll) * A_PITCH + 16u * (unsigned)lh;

  const int nkt = p.Kpad / BKC;
  issue(0);
  store(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    issue(min(kt + 1, nkt - 1));                                           // clamped: unconditional loads, exact waits
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float4 af[2];
      uint2 bl[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][aoff + (unsigned)(32 * i) * A_PITCH + 32u * ks]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const fp16x4_t v = tr_read(&Bs[buf][btr[j][rd] + 4096u * ks]);
          bl[j][rd] = __builtin_bit_cast(uint2, v);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint4 bb = make_uint4(bl[j][0].x, bl[j][0].y, bl[j][1].x, bl[j][1].y);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i]), __builtin_bit_cast(f16x8, bb), acc[i][j], 0, 0, 0);
        }
    }
    store(buf ^ 1);                                                        // tile kt+1 -> the other buffer (last one: rewritten, unread)
    __syncthreads();
  }

  // ---- epilogue (C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)); see conv3d.hip
  const int mbase = tileM * BM + 64 * wm;
  const int rows_left = p.DK - mbase - 4 * lh;
  const unsigned orow = (unsigned)p.SP * 2u;
  const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc(dst, 0, p.dst_bytes, 0x00020000);
  bool cval[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = sp0 + 64 * wn + 32 * j + ll;
    cval[j] = n < p.SP;
    const unsigned vb = cval[j] ? (((unsigned)img * (unsigned)p.DK + (unsigned)(mbase + 4 * lh)) * (unsigned)p.SP + (unsigned)n) * 2u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float old[16];
      if (p.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
          const unsigned vo = ro < rows_left ? vb : 0xffffffffu;
          old[r] = (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rd_, (int)vo, (int)((unsigned)ro * orow), 0));
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
        const unsigned vo = ro < rows_left ? vb : 0xffffffffu;
        float v = acc[i][j][r];
        if (bias) v += bias[min(mbase + ro + 4 * lh, p.DK - 1)];
        if (p.accumulate) v += old[r];
        __builtin_amdgcn_raw_buffer_store_b16(f16_bits(v), rd_, (int)vo, (int)((unsigned)ro * orow), 0);
      }
    }
  }
  if (psum) {
    float* red = reinterpret_cast<float*>(&As[0][0]);                     // [2 (wn)][BM][2] floats; the operand tiles are dead by now
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) { const float v = cval[j] ? acc[i][j][r] : 0.f; sm += v; sq += v * v; }
        sm = half_wave_sum_hi(sm);
        sq = half_wave_sum_hi(sq);
        const int ro = 64 * wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ll == 31) { red[(wn * BM + ro) * 2] = sm; red[(wn * BM + ro) * 2 + 1] = sq; }
      }
    __syncthreads();
    if (tid < BM && tileM * BM + tid < p.DK) {
      const long long m = tileM * BM + tid;
      psum[m * p.P + tileN] = red[tid * 2] + red[(BM + tid) * 2];
      psq[m * p.P + tileN] = red[tid * 2 + 1] + red[(BM + tid) * 2 + 1];
    }
  }
}

}  // namespace

namespace gca_conv {

int pw_launch(const PwParams& p, const void* src, const unsigned char* apack, const float* bias, void* dst, float* psum,
              float* psq, hipStream_t st) {
  const long long nblk = (long long)p.tilesM * p.N * p.tiles_sp;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return GCA_EINVAL;
  hipLaunchKernelGGL(conv_pw_f16_kernel, dim3((unsigned)nblk), dim3(256), 0, st, src, apack, bias, dst, psum, psq, p);
  return gca_launch_status();
}

}  // namespace gca_conv
