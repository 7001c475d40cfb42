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
//   * v_mfma_f32_32x32x16_f16, 128 x 128 tile, waves 2 x 2 (64 x 64 each), one barrier per 32 channels;
//   * round 3: both operand tiles travel global -> LDS by LDS-DMA (no registers in flight) into a ring of NST stages, NST - 1
//     k-tiles ahead of the MFMAs, waited for by instruction count: with 2..64 k-tiles per workgroup the register version was
//     bound by the latency of its one-tile-ahead fetch.  A DMA lays 64 x 16 bytes down in lane order, so the A rows sit
//     unpadded (64 bytes) and the bank swizzle moves to the source address: chunk c of row r lands in slot c ^ ((r >> 2) & 3);
//   * epilogue as in the other conv kernels (buffer stores, accumulate, BatchNorm partial sums).  (Measured and not kept: the
//     output tile staged through LDS and written as 16-byte row segments -- no faster, 0.210 vs 0.217 ms on the 64 -> 256
//     layer-1 conv, slower on the small maps: with 2..64 k-tiles per workgroup the kernel is bound by the latency of its
//     one-tile-ahead operand fetch, not by its stores; fetching TWO tiles ahead through a second register set costs a
//     resident workgroup (193 VGPRs) and was slower still, 0.240 ms.)
#include <cstring>
#include "conv_igemm_host.h"
#include "conv_halo.h"

using namespace gca_conv;

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int BM = 128, BN = 128, BKC = 32;          // tile rows (output channels), columns (positions), channels per k-tile
constexpr int NST = 4;                               // LDS stages of (A 8 KB + B 8 KB): three k-tiles in flight
constexpr int STAGE = 16384;
typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));
// (inline asm on purpose, see conv3d_wgrad_ts.hip: the builtin form makes hipcc drain vmcnt in front of every later LDS read)
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
__device__ __forceinline__ unsigned short f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
__device__ __forceinline__ fp16x4_t tr_read(const unsigned char* lds) {       // ds_read_b64_tr_b16 (8-byte aligned LDS address)
  return __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(lds));
}
// byte offset of 16-byte chunk `ch` (0..15) of row `row` in the [32][256 B] image (cdna guide T10, layout (b))
__device__ __forceinline__ unsigned boff(unsigned row, unsigned ch) { return 256u * row + 16u * (ch ^ (((row & 3u) << 2) | ((row >> 2) & 3u))); }

__global__ __launch_bounds__(256, 2) void conv_pw_f16_kernel(
    const void* __restrict__ src, const unsigned char* __restrict__ apack, const float* __restrict__ bias,
    void* __restrict__ dst, float* __restrict__ psum, float* __restrict__ psq, const PwParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [NST][A: 128 rows x 64 B | B: 32 rows x 256 B]
  const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1, lh = lane >> 5, ll = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);
  const int tileM = bid % p.tilesM, tileN = bid / p.tilesM;
  const int img = tileN / p.tiles_sp, tsp = tileN - img * p.tiles_sp;
  const int sp0 = tsp * BN;

  // ---- DMA pieces of this wave per k-tile: B pieces w, w + 4 (4 channel rows x 16 chunks each), A pieces w, w + 4 (16 rows x
  // 4 chunks each); a lane's LDS slot is fixed by the DMA (lane order), so the swizzles pick the SOURCE chunk
  const i32x4 rb = make_rsrc(src, p.src_bytes), ra = make_rsrc(apack, p.pack_bytes);
  const unsigned rowbytes = (unsigned)p.SP * 2u;
  unsigned bvoff[2], avoff[2];
  int brow_[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = wave + 4 * i;
    const unsigned row = (unsigned)(4 * pc + (lane >> 4)), slot = (unsigned)(lane & 15);
    const unsigned ch = slot ^ (((row & 3u) << 2) | ((row >> 2) & 3u));          // boff(): slot = ch ^ f(row)
    const bool ok = sp0 + 8 * (int)ch < p.SP;                                    // (SP % 8 == 0: a chunk is all inside or all outside)
    bvoff[i] = ok ? ((unsigned)img * p.src_nstride + (unsigned)(sp0 + 8 * (int)ch)) * 2u + row * rowbytes : 0xffffffffu;
    brow_[i] = (int)row;
    const unsigned arow = (unsigned)(16 * pc + (lane >> 2)), aslot = (unsigned)(lane & 3);
    const unsigned ac = aslot ^ ((arow >> 2) & 3u);
    avoff[i] = ((unsigned)(tileM * BM) + arow) * (unsigned)p.Kpad * 2u + 16u * ac;
  }
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(lds_void*)smem);
  const int nkt = p.Kpad / BKC;
  auto issue = [&](int kt) __attribute__((always_inline)) {               // always 4 DMA instructions per wave: the waits count them
    const bool live = kt < nkt;
    const int c0 = kt * BKC;
    const unsigned st = lds0 + (unsigned)((kt % NST) * STAGE);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned v = live && c0 + brow_[i] < p.Kc ? bvoff[i] : 0xffffffffu;  // channels past the end: zeros
      dma16(rb, v, (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)c0 * rowbytes)), st + 8192u + (unsigned)((wave + 4 * i) * 1024));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      dma16(ra, live ? avoff[i] : 0xffffffffu, (unsigned)__builtin_amdgcn_readfirstlane(c0 * 2), st + (unsigned)((wave + 4 * i) * 1024));
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
  unsigned btr[2][2];                                                      // [j][rd] for k-step 0 of a stage; k-step 1 = rows + 16
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const unsigned row = (unsigned)(8 * (g >> 1) + 4 * rd + q4);
      const unsigned ch = (unsigned)((64 * wn + 32 * j) / 8 + 2 * (g & 1) + (p4 >> 1));
      btr[j][rd] = 8192u + boff(row, ch) + 8u * (p4 & 1);
    }
  // rows 16..31 of the image: (row & 3) and ((row >> 2) & 3) are those of row - 16, so the swizzle is the same: + 16 * 256 bytes
  unsigned aoff[2][2];                                                     // [i][ks]: row 64 wm + 32 i + ll, chunk 2 ks + lh
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned row = (unsigned)(64 * wm + 32 * i + ll);
      aoff[i][ks] = row * 64u + 16u * ((unsigned)(2 * ks + lh) ^ ((row >> 2) & 3u));
    }

#pragma unroll
  for (int kt = 0; kt < NST - 1; ++kt) issue(kt);
  for (int kt = 0; kt < nkt; ++kt) {
    // stage kt has landed once at most the (NST - 2) * 4 younger DMA instructions of this wave are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * 4) : "memory");
    __syncthreads();                                                       // every wave's pieces are in; stage kt - 1 is free
    issue(kt + NST - 1);
    const unsigned char* St = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float4 af[2];
      uint2 bl[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(St + aoff[i][ks]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const fp16x4_t v = tr_read(St + btr[j][rd] + 4096u * ks);
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
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // (the dead tail stages: nothing may land after the epilogue took the LDS)
  __syncthreads();

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
    float* red = reinterpret_cast<float*>(smem);                          // [2 (wn)][BM][2] floats; the operand tiles are dead by now
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
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            NST * STAGE) != hipSuccess) return GCA_ELAUNCH;
    raised = true;
  }
  hipLaunchKernelGGL(conv_pw_f16_kernel, dim3((unsigned)nblk), dim3(256), NST * STAGE, st, src, apack, bias, dst, psum, psq, p);
  return gca_launch_status();
}

}  // namespace gca_conv
