// 3D convolution forward / dgrad for gfx950 as a TAP-OUTER implicit GEMM with the input window staged in LDS.
//
// conv3d.hip gathers every (channel, tap) element of the im2col operand from L1/L2 once per tap and, in the split-product
// arithmetic modes, splits both operands into bf16 parts inside the k loop.  Here the reduction runs
//
//     for 16-channel chunk c:    stage the HALO of the tile -- every source position any tap of any output of the tile
//                                touches, 16 channels deep -- in LDS ONCE: [position][part][16 ch], already split into
//                                bf16 hi/mid/lo parts (or kept fp32)
//        for tap t:              A = packed weights of (c, t), PRE-SPLIT at pack time; B = the same LDS halo read at the
//                                tap's offset  ->  one 32x32x16 MFMA step per (row tile, column tile)
//
// so an input element is fetched from memory and split once per (tile, chunk) instead of once per tap (7x fewer gathers and
// operand splits on the (7,1,1) stem conv, 9x on (1,3,3), 27x on 3x3x3), the weight operand is never split in the loop, and
// the k loop is LDS reads + MFMAs.  A tile is a BOX of output positions (bd x bh x bw = 128 or 256, w fastest) inside one clip
// so that its halo is compact; activations stay NCDHW and lanes run along W both for the staging loads and for the stores.
// Same problem classes (forward, unit-stride dgrad, one class per stride residue) and the same epilogue (buffer stores,
// BatchNorm partial sums by DPP, split-K slabs) as conv3d.hip.  Reference call sites: every nn.Conv3d with >= 16 input
// channels (resnet2p1d.py:13-36,169-174; s3d_1.py:53-57; resnet.py:14-22,77-78).
#include <cstring>
#include <type_traits>
#include "conv_igemm_host.h"
#include "conv_halo.h"

using namespace gca_conv;

namespace {

constexpr int HR = 3;                 // staging rounds per thread: the halo holds up to 128*HR positions
constexpr int PPAD = 128 * HR;        // staging tasks per channel half (tasks beyond the halo write the dump slot)

// arithmetic 3 = fp16 STORAGE: activations are _Float16 in HBM, products on v_mfma_f32_32x32x16_f16, fp32 accumulation
constexpr int rowb(int math) { return math == 3 ? 32 : (math == 2 ? 96 : 64); }   // bytes of one 16-k row: fp16 / fp32 / hi+lo / hi+mid+lo
constexpr int pitchb(int math) { return rowb(math) + 16; }            // 48 / 80 / 112 B: odd number of 16-B slots -> conflict-free b128 reads
constexpr int nparts(int math) { return (math == 0 || math == 3) ? 1 : (math == 2 ? 3 : 2); }
__device__ __forceinline__ unsigned short f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
__device__ __forceinline__ float f16_val(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// (second launch bound = waves per SIMD: two workgroups per CU whenever the accumulators leave room for it)
template <int TM, int TN, int MATH>
__global__ __launch_bounds__(256, (TM * TN <= 6 ? 2 : 1)) void conv_halo_kernel(
    const void* __restrict__ src, const unsigned char* __restrict__ apack, const int* __restrict__ tapdelta,
    const float* __restrict__ bias, void* __restrict__ dst, float* __restrict__ psum, float* __restrict__ psq,
    float* __restrict__ slab, const HaloParams hp) {
  constexpr bool F16 = MATH == 3;                               // activations (src, dst) are fp16
  constexpr unsigned ES = F16 ? 2u : 4u;
  constexpr int BM = 32 * TM;                                   // x 128*TN columns: TN column tiles of 32 per wave
  constexpr int ROWB = rowb(MATH), PITCH = pitchb(MATH), NP = nparts(MATH);
  constexpr int RPC = ROWB / 16;                                // 16-byte pieces per packed row
  constexpr int NPC = BM * RPC;                                 // pieces of one A step
  constexpr int A_PC = (NPC + 255) / 256;
  const IgemmParams& p = hp.g;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* const tapoff = reinterpret_cast<int*>(smem);            // [64] byte offsets of the taps inside the halo
  unsigned char* const As = smem + 256;                         // [2][BM][PITCH]
  unsigned char* const Hs = As + 2 * BM * PITCH;                // [P + 1][PITCH]; slot P takes the stores of idle staging tasks

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, lh = lane >> 5, ll = lane & 31;
  int bid = gca_xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid % p.splits; bid /= p.splits;
  const int tileM = bid % p.tilesM, tileN = bid / p.tilesM;
  // tile -> clip, box origin (iteration sub-grid coordinates)
  const int per_img = hp.nbd * hp.nbh * hp.nbw;
  const int img = tileN / per_img;
  int tb = tileN - img * per_img;
  const int tbd = tb / (hp.nbh * hp.nbw); tb -= tbd * (hp.nbh * hp.nbw);
  const int tbh = tb / hp.nbw, tbw = tb - tbh * hp.nbw;
  const int q0d = tbd * hp.bd, q0h = tbh * hp.bh, q0w = tbw * hp.bw;

  if (tid < 64) {
    int off = 0;
    if (tid < p.ntaps) {
      const int pk = tapdelta[tid];
      const int dd = (pk << 16) >> 24, dh = (pk << 8) >> 24, dw = pk >> 24;
      off = (((dd - hp.dmin_d) * hp.hh + (dh - hp.dmin_h)) * hp.hw + (dw - hp.dmin_w)) * PITCH;
    }
    tapoff[tid] = off;
  }

  // ---- staging tasks of this thread: (halo position, channel half) per round, fixed for the whole K loop
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, p.src_bytes, 0x00020000);
  unsigned hvoff[HR];          // byte offset of channel 0 of the chunk at this position, all-ones when outside the tensor
  unsigned hwoff[HR];          // LDS byte offset of the task's slot
  int hhalf[HR];
  {
    const int hhw = hp.hh * hp.hw;
    const int s0d = q0d * p.m_d + hp.h0d, s0h = q0h * p.m_h + hp.h0h, s0w = q0w * p.m_w + hp.h0w;
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      const int task = tid + 256 * r;
      const int half = task >= PPAD ? 1 : 0;
      const int pos = task - half * PPAD;
      const int zd = pos / hhw, rem = pos - zd * hhw;
      const int zh = rem / hp.hw, zw = rem - zh * hp.hw;
      const int sd = s0d + zd, sh = s0h + zh, sw = s0w + zw;
      const bool ok = pos < hp.P && (unsigned)sd < (unsigned)p.SD && (unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW;
      const unsigned e = (unsigned)((long long)img * p.src_nstride) + (unsigned)((sd * p.SH + sh) * p.SW + sw);
      hvoff[r] = ok ? e * ES + (unsigned)half * 8u * hp.cs_bytes : 0xffffffffu;
      hwoff[r] = (unsigned)(pos < hp.P ? pos : hp.P) * PITCH + (unsigned)half * (MATH == 0 ? 32u : 16u);
      hhalf[r] = half;
    }
  }
  // ---- A pieces of this thread
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(apack), 0, hp.pack_bytes, 0x00020000);
  unsigned avoff[A_PC], awoff[A_PC];
#pragma unroll
  for (int u = 0; u < A_PC; ++u) {
    const int i = tid + 256 * u;
    const int row = i / RPC, sub = i - row * RPC;
    avoff[u] = i < NPC ? (unsigned)(tileM * BM) * ROWB + (unsigned)i * 16u : 0xffffffffu;
    awoff[u] = (unsigned)(i < NPC ? row : 0) * PITCH + (unsigned)sub * 16u;
  }
  const unsigned astep = (unsigned)hp.Mrows * ROWB;             // bytes between consecutive (chunk, tap) steps of the pack

  float hreg[HR][8];
  uint4 areg[A_PC];
  int hchunk = 0;               // chunk whose halo sits in hreg (for the input transform)
  auto halo_issue = [&](int c) __attribute__((always_inline)) {
    const int c0 = c * 16;
    hchunk = c;
#pragma unroll
    for (int r = 0; r < HR; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned v = (c0 + 8 * hhalf[r] + j < p.SC) ? hvoff[r] : 0xffffffffu;
        if (F16) hreg[r][j] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)v, (int)((unsigned)(c0 + j) * hp.cs_bytes), 0));
        else hreg[r][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)v, (int)((unsigned)(c0 + j) * hp.cs_bytes), 0));
      }
  };
  auto halo_store = [&]() __attribute__((always_inline)) {
    if constexpr (!F16) {
      if (hp.in_scale) {
        // relu(x * scale[c] + shift[c]) on the staged values (the producer's BatchNorm + ReLU, bn.hip bn_apply_kernel's
        // arithmetic); positions outside the tensor and channels past the end stay exactly zero (the padding pads the
        // TRANSFORMED tensor; a garbage scale must not turn a zero-weight product into NaN)
#pragma unroll
        for (int r = 0; r < HR; ++r) {
          const int hu = __builtin_amdgcn_readfirstlane(hhalf[r]);          // (wave-uniform: 384 = 6 waves of tasks per half)
          const int cb = hchunk * 16 + 8 * hu;
          const bool okp = hvoff[r] != 0xffffffffu;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float sc = hp.in_scale[cb + j], sf = hp.in_shift[cb + j];
            const float v = fmaxf(hreg[r][j] * sc + sf, 0.f);
            hreg[r][j] = (okp && cb + j < p.SC) ? v : 0.f;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      unsigned char* d = Hs + hwoff[r];
      if (F16) {                                  // 8 channels of one position: 8 halves = one 16-byte store
        auto u = [&](int j) __attribute__((always_inline)) { return __float_as_uint(hreg[r][j]); };
        *reinterpret_cast<uint4*>(d) = make_uint4(u(0) | (u(1) << 16), u(2) | (u(3) << 16), u(4) | (u(5) << 16), u(6) | (u(7) << 16));
      } else if (MATH == 0) {
        *reinterpret_cast<float4*>(d) = make_float4(hreg[r][0], hreg[r][1], hreg[r][2], hreg[r][3]);
        *reinterpret_cast<float4*>(d + 16) = make_float4(hreg[r][4], hreg[r][5], hreg[r][6], hreg[r][7]);
      } else if (MATH == 2) {
        uint4 h, m, l;
        split_bf16x3(hreg[r][0], hreg[r][1], h.x, m.x, l.x);
        split_bf16x3(hreg[r][2], hreg[r][3], h.y, m.y, l.y);
        split_bf16x3(hreg[r][4], hreg[r][5], h.z, m.z, l.z);
        split_bf16x3(hreg[r][6], hreg[r][7], h.w, m.w, l.w);
        *reinterpret_cast<uint4*>(d) = h;
        *reinterpret_cast<uint4*>(d + 32) = m;
        *reinterpret_cast<uint4*>(d + 64) = l;
      } else {
        uint4 h, l;
        split_bf16x2(hreg[r][0], hreg[r][1], h.x, l.x);
        split_bf16x2(hreg[r][2], hreg[r][3], h.y, l.y);
        split_bf16x2(hreg[r][4], hreg[r][5], h.z, l.z);
        split_bf16x2(hreg[r][6], hreg[r][7], h.w, l.w);
        *reinterpret_cast<uint4*>(d) = h;
        *reinterpret_cast<uint4*>(d + 32) = l;
      }
    }
  };
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

  // ---- per-lane fragment addresses
  unsigned bpos[TN];           // halo byte offset of this lane's column of column tile j (tap offset and part added per read)
  bool cval[TN];
  int cq[TN][3];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cidx = wn * (TN * 32) + j * 32 + ll;
    const int zw = cidx & (hp.bw - 1), t1 = cidx >> hp.lbw;
    const int zh = t1 & (hp.bh - 1), zd = t1 >> hp.lbh;
    bpos[j] = (unsigned)(((zd * p.m_d) * hp.hh + zh * p.m_h) * hp.hw + zw * p.m_w) * PITCH + (unsigned)lh * 16u;
    cq[j][0] = q0d + zd; cq[j][1] = q0h + zh; cq[j][2] = q0w + zw;
    cval[j] = cq[j][0] < p.QD && cq[j][1] < p.QH && cq[j][2] < p.QW;
  }
  const unsigned aoff = (unsigned)ll * PITCH + (unsigned)lh * 16u;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragments.  A fragment read is one ds_read_b128 of [row][part q][8 k of lane half]; for fp32 (MATH 0) the two
  // reads q = 0, 1 are the k-steps 0..7 / 8..15 of the row (a lane of half h holds k = 8q + 4h + e).
  constexpr int NQ = MATH == 0 ? 2 : NP;
  float4 bfc[TN][NQ], afc[NQ], afn[NQ];
  auto load_b = [&](float4 (&d)[TN][NQ], unsigned toff) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < NQ; ++q) d[j][q] = *reinterpret_cast<const float4*>(Hs + bpos[j] + toff + 32 * q);
  };
  auto load_a = [&](float4 (&d)[NQ], int buf, int i) __attribute__((always_inline)) {
    const unsigned char* Ab = As + buf * (BM * PITCH) + aoff + i * (32 * PITCH);
#pragma unroll
    for (int q = 0; q < NQ; ++q) d[q] = *reinterpret_cast<const float4*>(Ab + 32 * q);
  };
  auto mma_group = [&](int i, const float4 (&af)[NQ], const float4 (&bf)[TN][NQ]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (MATH == 3) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[j][0]),
                                                          acc[i][j], 0, 0, 0);
      } else if constexpr (MATH == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bf[j][q].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bf[j][q].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bf[j][q].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bf[j][q].w, acc[i][j], 0, 0, 0);
        }
      } else {
        // bf16x3: x*y ~= xh*yh + xh*yl + xl*yh; bf16x6: hh + hm + mh + hl + lh + mm (conv3d.hip); fp32 accumulation
        const bf16x8 xh = __builtin_bit_cast(bf16x8, af[0]), xl = __builtin_bit_cast(bf16x8, af[NQ - 1]);
        const bf16x8 yh = __builtin_bit_cast(bf16x8, bf[j][0]), yl = __builtin_bit_cast(bf16x8, bf[j][NQ - 1]);
        if (MATH == 2) {
          const bf16x8 xm = __builtin_bit_cast(bf16x8, af[1]), ym = __builtin_bit_cast(bf16x8, bf[j][1]);
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
    }
  };

  // ---- K loop: chunks [c0, c1) of this split, `ntaps` steps each; the A tiles form ONE sequence s = chunk*ntaps + tap that
  // alternates between two LDS buffers (tile s+1 goes from registers to LDS behind the MFMAs of step s, the global loads of
  // tile s+2 are issued right after; one barrier per step).  After the last tap of a chunk the next chunk's halo -- fetched
  // while this chunk was being multiplied -- is split and written over the current one.
  // (Measured and not kept: three A buffers with the fragments of step s+1 read during the MFMAs of step s and the halo
  // re-staged inside the last step -- 40 more VGPRs, 50 % more LDS for A, 5-10 % slower on every layer shape.)
  // (Also measured and not kept: for tiles of <= 64 rows, A fragments straight from the L2-resident packed weights, one step
  // ahead in registers -- no LDS copy of A, no barrier per step.  Every one of the four waves then pulls the whole A tile
  // through the L1: 155 vs 160 TF/s on the 64 x 256 bf16x6 tile, 23.1 vs 22.7 ms per step.  It pays in conv3d_stem.hip, whose
  // waves split the rows.)
  const int c0 = split * hp.chunks_per_split;
  int c1 = c0 + hp.chunks_per_split; if (c1 > hp.nchunks) c1 = hp.nchunks;
  const int nt = p.ntaps;
  const int s_end = c1 * nt;
  if (c0 < c1) {
    int s = c0 * nt;
    halo_issue(c0);
    a_issue(s);
    halo_store();
    a_store(0);
    __syncthreads();                       // also publishes tapoff[]
    a_issue(min(s + 1, s_end - 1));        // unconditional fetches with a clamped index: exact vmcnt waits, no branches
    halo_issue(min(c0 + 1, c1 - 1));
    // ONE flat loop over the steps (a nested chunk / tap loop makes hipcc keep two copies of the accumulators and move
    // all of them at every chunk boundary).  The tap offset of the NEXT step is read one step ahead, so no step starts
    // with a dependent LDS read in front of its fragment reads.
    int t = 0, c = c0, buf = 0;
    unsigned toff = (unsigned)tapoff[0];
    for (; s < s_end; ++s) {
      const bool last_tap = t == nt - 1;
      const unsigned toff_next = (unsigned)tapoff[last_tap ? 0 : t + 1];
      load_b(bfc, toff);
      load_a(afc, buf, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i + 1 < TM) load_a(afn, buf, i + 1);
        mma_group(i, afc, bfc);
        if (i + 1 < TM) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) afc[q] = afn[q];
        }
      }
      a_store(buf ^ 1);                    // tile s+1 (fetched one step ago) -> the other buffer
      a_issue(min(s + 2, s_end - 1));
      __syncthreads();
      if (last_tap && c + 1 < c1) {        // every wave is done reading this chunk's halo
        halo_store();
        halo_issue(min(c + 2, c1 - 1));
        __syncthreads();
      }
      buf ^= 1;
      toff = toff_next;
      if (last_tap) { t = 0; ++c; } else ++t;
    }
  }

  // ---- epilogue (C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)); see conv3d.hip
  const int mbase = tileM * BM;
  const int rows_left = p.DK - mbase - 4 * lh;
  if (p.splits > 1) {
    // partial tile -> slab[split][m][n], n = the flat column index of conv3d.hip (conv_splitk_finish_kernel finishes it)
    float* sl = slab + (long long)split * p.DK * p.Ntot;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(sl, 0, p.slab_bytes, 0x00020000);
    const unsigned rowb_ = (unsigned)p.Ntot * 4u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const unsigned n = (unsigned)(((img * p.QD + cq[j][0]) * p.QH + cq[j][1]) * p.QW + cq[j][2]);
      const unsigned vb = cval[j] ? (n + (unsigned)(mbase + 4 * lh) * (unsigned)p.Ntot) * 4u : 0xffffffffu;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
          const unsigned vo = ro < rows_left ? vb : 0xffffffffu;
          const float v = acc[i][j][r];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rd, (int)vo, (int)((unsigned)ro * rowb_), 0);
        }
    }
    return;
  }
  const int DHW = p.DH * p.DW;
  const unsigned DSP = (unsigned)(p.DD * DHW);
  const unsigned rowb_ = DSP * ES;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, p.dst_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const unsigned dsp = (unsigned)((cq[j][0] * p.dm_d + p.do_d) * DHW + (cq[j][1] * p.dm_h + p.do_h) * p.DW + (cq[j][2] * p.dm_w + p.do_w));
    const unsigned vb = cval[j] ? (((unsigned)img * (unsigned)p.DK + (unsigned)(mbase + 4 * lh)) * DSP + dsp) * ES : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float old[16];
      if (p.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
          const unsigned vo = ro < rows_left ? vb : 0xffffffffu;
          if (F16) old[r] = f16_val((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rd, (int)vo, (int)((unsigned)ro * rowb_), 0));
          else old[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, (int)vo, (int)((unsigned)ro * rowb_), 0));
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
        const unsigned vo = ro < rows_left ? vb : 0xffffffffu;
        float v = acc[i][j][r];
        if (bias) v += bias[min(mbase + ro + 4 * lh, p.DK - 1)];
        if (p.accumulate) v += old[r];
        if (F16) __builtin_amdgcn_raw_buffer_store_b16(f16_bits(v), rd, (int)vo, (int)((unsigned)ro * rowb_), 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rd, (int)vo, (int)((unsigned)ro * rowb_), 0);
      }
    }
  }
  if (psum) {
    // per-channel partial sum / sum of squares of this tile.  A column of the box that lies outside the iteration grid is
    // not an output (its window may still cover real input): it is masked out here.
    float* red = reinterpret_cast<float*>(As);                   // [4][BM][2] floats; the operand tiles are dead by now
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) { const float v = cval[j] ? acc[i][j][r] : 0.f; sm += v; sq += v * v; }
        sm = half_wave_sum_hi(sm);
        sq = half_wave_sum_hi(sq);
        const int ro = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ll == 31) { red[(wn * BM + ro) * 2] = sm; red[(wn * BM + ro) * 2 + 1] = sq; }
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

template <int TM, int TN>
int launch_halo_math(int math, dim3 grid, size_t lds, hipStream_t st, const void* src, const unsigned char* apack,
                     const int* tapdelta, const float* bias, void* dst, float* ps, float* pq, float* slab,
                     const HaloParams& hp) {
  // dynamic LDS beyond the default 64 KB window needs the attribute (once per instantiation; not a stream operation)
  static bool raised[4] = {false, false, false, false};
#define GCA_HK(M)                                                                                                          \
  {                                                                                                                        \
    if (lds > (48u << 10) && !raised[M]) {                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<TM, TN, M>),                                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10) != hipSuccess) return GCA_ELAUNCH;   \
      raised[M] = true;                                                                                                    \
    }                                                                                                                      \
    hipLaunchKernelGGL((conv_halo_kernel<TM, TN, M>), grid, dim3(256), lds, st, src, apack, tapdelta, bias, dst, ps, pq,   \
                       slab, hp);                                                                                          \
  }
  switch (math) {
    case 0: GCA_HK(0) break;
    case 1: GCA_HK(1) break;
    case 3: GCA_HK(3) break;
    default: GCA_HK(2) break;
  }
#undef GCA_HK
  return gca_launch_status();
}

}  // namespace

namespace gca_conv {

size_t halo_lds_bytes(int bm, int math, int P) { return (size_t)(2 * bm + P + 1) * pitchb(math) + 64 * sizeof(int); }
int halo_max_positions() { return PPAD; }

int halo_launch(const HaloCfg& c, const HaloParams& hp_in, const void* src, const unsigned char* apack, const int* tapdelta,
                const float* bias, void* dst, float* psum, float* psq, float* slab, hipStream_t st) {
  HaloParams hp = hp_in;
  const size_t lds = halo_lds_bytes(c.bm, c.math, hp.P);
  const long long nblk = (long long)hp.g.tilesM * hp.g.tilesN * hp.g.splits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return GCA_EINVAL;
  const dim3 grid((unsigned)nblk);
  const int tn = c.bn / 128, tm = c.bm / 32;
#define GCA_HL(A, B) if (tm == A && tn == B) return launch_halo_math<A, B>(c.math, grid, lds, st, src, apack, tapdelta, bias, dst, psum, psq, slab, hp)
  GCA_HL(1, 1); GCA_HL(2, 1); GCA_HL(3, 1); GCA_HL(4, 1); GCA_HL(5, 1);
  GCA_HL(1, 2); GCA_HL(2, 2); GCA_HL(3, 2); GCA_HL(4, 2);
#undef GCA_HL
  return GCA_EINVAL;
}

}  // namespace gca_conv

// Diagnostic: workgroups of conv_halo_kernel<tm, tn, math> the runtime co-schedules per CU at `lds_bytes` of dynamic LDS
// (registers + LDS as the hardware allocates them); -1 for a shape that is not instantiated.
extern "C" int gca_conv_halo_occupancy(int tm, int tn, int math, int64_t lds_bytes) {
  int n = -1;
#define GCA_HO(A, B, M)                                                                                                   \
  if (tm == A && tn == B && math == M) {                                                                                  \
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<A, B, M>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                        160 << 10);                                                                                       \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_halo_kernel<A, B, M>, 256, (size_t)lds_bytes) != hipSuccess) \
      n = -1;                                                                                                             \
  }
  GCA_HO(2, 2, 2) GCA_HO(2, 2, 0) GCA_HO(5, 1, 2) GCA_HO(4, 1, 2) GCA_HO(3, 1, 2) GCA_HO(2, 1, 2) GCA_HO(1, 1, 2)
#undef GCA_HO
  return n;
}

