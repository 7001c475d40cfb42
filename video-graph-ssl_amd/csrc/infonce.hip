// MoCo queue + InfoNCE on gfx950.
//
//   logits[i,0]   = q_i . k_i / T
//   logits[i,1+j] = q_i . queue_j / T          one batched clip x queue GEMM on the fp32 matrix cores,
//                                              streaming the (K,D) queue from HBM exactly once
//   loss          = mean_i ( logsumexp(logits_i) - logits[i,0] )
//   dq            = ( dlogits[:,0] * k + dlogits[:,1:] @ queue ) / T     split over queue slices
//
// HBM-bound (AI ~ 12.7 F/B at b=32): algorithmic bytes = K*D*4 (queue) + b*(K+1)*4 (logits).
// Reference: lib/memory/mem_moco.py:14-49,60-88; lib/memory/criterion.py:34-45.
#include <cstdint>
#include <cstdlib>
#include "gca_common.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int DCH = 64;              // feature chunk staged per pass
constexpr int QLD = DCH + 4;         // LDS row stride of the staged queue tile (float4-aligned)

// One wave = 32 queue rows x all batch rows (in tiles of 32).  WAVES waves per workgroup share
// the staged q tile.  MFMA orientation: A = q (M = batch rows), B = queue^T (N = queue rows), so
// the 32 lanes of a half-wave hold 32 consecutive logits of one batch row -> coalesced stores.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void moco_logits_kernel(
    const float* __restrict__ q, const float* __restrict__ kpos, const float* __restrict__ queue,
    int b, long long K, int D, float inv_T, float* __restrict__ logits) {
  __shared__ float Qs[DCH][32];                  // q tile, k-major: A operand reads are row-contiguous
  __shared__ float Ns[WAVES][32][QLD];           // per-wave queue tile [row][feature]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lh = lane >> 5, ll = lane & 31;
  const long long row0 = ((long long)blockIdx.x * WAVES + wave) * 32;
  const long long ld = K + 1;
  const int mtiles = (b + 31) / 32;

  for (int mt = 0; mt < mtiles; ++mt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int d0 = 0; d0 < D; d0 += DCH) {
      __syncthreads();
      // stage q[mt*32 .. +32][d0 .. d0+DCH) transposed
      for (int i = tid; i < 32 * DCH; i += WAVES * 64) {
        const int m = i / DCH, kk = i % DCH;
        const int row = mt * 32 + m;
        Qs[kk][m] = (row < b && d0 + kk < D) ? q[(long long)row * D + d0 + kk] : 0.f;
      }
      // stage this wave's 32 queue rows: lanes along the feature axis (coalesced float4)
      for (int i = lane; i < 32 * (DCH / 4); i += 64) {
        const int r = i / (DCH / 4), c4 = i % (DCH / 4);
        const long long row = row0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < K && d0 + c4 * 4 < D) v = *reinterpret_cast<const float4*>(queue + row * D + d0 + c4 * 4);
        *reinterpret_cast<float4*>(&Ns[wave][r][c4 * 4]) = v;
      }
      __syncthreads();
#pragma unroll 8
      for (int kk = 0; kk < DCH; kk += 2) {
        const float a = Qs[kk + lh][ll];
        const float bb = Ns[wave][ll][kk + lh];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
      }
    }
    const long long j = row0 + ll;
    if (j < K) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (i < b) logits[(long long)i * ld + 1 + j] = acc[r] * inv_T;
      }
    }
  }
  // positive column: block 0 does the b row dots (wave per row)
  if (blockIdx.x == 0) {
    for (int i = wave; i < b; i += WAVES) {
      float s = 0.f;
      for (int d = lane; d < D; d += 64) s += q[(long long)i * D + d] * kpos[(long long)i * D + d];
      s = gca_wave_sum(s);
      if (lane == 0) logits[(long long)i * ld] = s * inv_T;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused forward for D <= 128 (the MoCo feature width): logits + per-(row, 32-column block) log-sum-exp partials +
// rank counts in ONE pass over the queue.  One wave = 32 queue rows; a workgroup's waves share the staged q tile.
// Every global load of a wave is issued up front (float4, lanes along the feature axis), the operands sit in LDS
// as [row][k] with a (D+4)-float pitch and are read as conflict-free 128-bit fragments (4 k-steps per read, same
// k-permutation on both operands as in conv3d.hip).  lse_finish_kernel then folds the partials (and the positive
// column) per batch row, so the (b, K+1) logits are written once and never read back.
// ---------------------------------------------------------------------------------------------------------------
constexpr int FD = 128;              // max feature width of the fused kernel
constexpr int FP = FD + 4;           // LDS pitch

__device__ __forceinline__ float half_wave_max_hi(float v) {      // max over the 32 lanes of each wave half -> lanes 16..31 / 48..63
#define GCA_DPPM(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (x)), __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xf, false))
  v = fmaxf(v, GCA_DPPM(v, 0xB1, 0xf));
  v = fmaxf(v, GCA_DPPM(v, 0x4E, 0xf));
  v = fmaxf(v, GCA_DPPM(v, 0x141, 0xf));
  v = fmaxf(v, GCA_DPPM(v, 0x140, 0xf));
  v = fmaxf(v, GCA_DPPM(v, 0x142, 0xa));
#undef GCA_DPPM
  return v;
}
__device__ __forceinline__ float half_wave_sum_hi_f(float v) {
#define GCA_DPPS(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xf, false))
  v += GCA_DPPS(v, 0xB1, 0xf);
  v += GCA_DPPS(v, 0x4E, 0xf);
  v += GCA_DPPS(v, 0x141, 0xf);
  v += GCA_DPPS(v, 0x140, 0xf);
  v += GCA_DPPS(v, 0x142, 0xa);
#undef GCA_DPPS
  return v;
}

// One wave = 32 queue rows; the waves of a workgroup share the staged q tile.  All global loads are float4 with lanes
// along the feature axis (whole 512-byte rows), issued before their first use; operands sit in LDS as [row][k] with a
// (D+4)-float pitch and are read as conflict-free 128-bit k-permuted fragments (4 k-steps per read, as in conv3d.hip).
// (Tried and dropped: fetching q / k / queue fragments lane-per-row straight into registers -- no LDS, no barriers --
// which needs ~280 registers and 32 cache lines per load instruction: 2x slower at K = 65536.)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void moco_logits_fused_kernel(
    const float* __restrict__ q, const float* __restrict__ kpos, const float* __restrict__ queue,
    int b, long long K, int D, float inv_T, float* __restrict__ logits, int ncb,
    float* __restrict__ pm, float* __restrict__ ps, float* __restrict__ pc, gca_magic md4) {
  constexpr int NT = FD / 8;
  __shared__ __attribute__((aligned(16))) float Qs[32 * FP];           // [32][FP]   q tile, k contiguous
  __shared__ float L0[32];                                             // positive logits of the tile's batch rows
  __shared__ __attribute__((aligned(16))) float Ns[WAVES][32 * FP];    // per-wave queue tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lh = lane >> 5, ll = lane & 31;
  const int blk = blockIdx.x;
  const long long cb = (long long)blk * WAVES + wave;                  // 32-column block of this wave
  const long long row0 = cb * 32;
  const long long ld = K + 1;
  const int mtiles = (b + 31) / 32;
  const int d4 = D >> 2, nt = D >> 3;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float* Nw = Ns[wave];
  {  // this wave's 32 queue rows, once: all loads in flight, then LDS (rows past K read as zeros)
    float4 v[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int i = lane + 64 * it, r = (int)gca_fdiv((unsigned)i, md4), c4 = i - r * d4;
      v[it] = zero4;
      if (i < 32 * d4 && row0 + r < K) v[it] = *reinterpret_cast<const float4*>(queue + (row0 + r) * D + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int i = lane + 64 * it, r = (int)gca_fdiv((unsigned)i, md4), c4 = i - r * d4;
      if (i < 32 * d4) *reinterpret_cast<float4*>(&Nw[r * FP + c4 * 4]) = v[it];
    }
  }
  for (int mt = 0; mt < mtiles; ++mt) {
    __syncthreads();                              // previous tile's Qs / L0 readers are done
    for (int i = tid; i < 32 * d4; i += WAVES * 64) {
      const int m = (int)gca_fdiv((unsigned)i, md4), c4 = i - m * d4, row = mt * 32 + m;
      *reinterpret_cast<float4*>(&Qs[m * FP + c4 * 4]) = row < b ? *reinterpret_cast<const float4*>(q + (long long)row * D + c4 * 4) : zero4;
    }
    // positive logits of these 32 batch rows (every workgroup needs them for the rank counts): wave 0, one batch row
    // per lane pair -- lane (m, half) walks half of the features of row m
    __syncthreads();
    {   // positive logits: 32 rows over WAVES waves, 64/(32/WAVES) lanes per row, float4 pieces strided over the lanes
      constexpr int RPW = 32 / WAVES;                 // rows per wave
      constexpr int LPR = 64 / RPW;                   // lanes per row
      const int m = wave * RPW + lane / LPR, part = lane % LPR;
      const int row = mt * 32 + m;
      float sdot = 0.f;
      for (int c4 = part; c4 < d4; c4 += LPR) {
        const float4 a = *reinterpret_cast<const float4*>(&Qs[m * FP + c4 * 4]);
        const float4 kq = row < b ? *reinterpret_cast<const float4*>(kpos + (long long)row * D + c4 * 4) : zero4;
        sdot += a.x * kq.x + a.y * kq.y + a.z * kq.z + a.w * kq.w;
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
      if (part == 0) L0[m] = sdot * inv_T;
    }
    __syncthreads();
    if (cb == 0 && lane < 32 && mt * 32 + lane < b) logits[(long long)(mt * 32 + lane) * ld] = L0[lane];
    if (row0 >= K) continue;                                           // (workgroup tail: keeps the barriers uniform)

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int t = 0; t < nt; ++t) {
      const float4 a = *reinterpret_cast<const float4*>(&Qs[ll * FP + 8 * t + 4 * lh]);
      const float4 bq = *reinterpret_cast<const float4*>(&Nw[ll * FP + 8 * t + 4 * lh]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, acc, 0, 0, 0);
    }
    const long long j = row0 + ll;
    const bool jv = j < K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int i = mt * 32 + m;
      const float v = acc[r] * inv_T;
      if (jv && i < b) logits[(long long)i * ld + 1 + j] = v;
      if (pm) {
        const float vm = jv ? v : -INFINITY;
        const float mx = half_wave_max_hi(vm);                         // valid in lanes 16..31 / 48..63
        // every lane of the half needs the block maximum: two scalar lane reads instead of an LDS-crossbar shuffle
        const float mxa = lh ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 63))
                             : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 31));
        const float e = jv && mxa > -INFINITY ? expf(v - mxa) : 0.f;
        const float se = half_wave_sum_hi_f(e);
        const float ge = half_wave_sum_hi_f(jv && v >= L0[m] ? 1.f : 0.f);
        if (ll == 31 && i < b) { pm[cb * b + i] = mxa; ps[cb * b + i] = se; pc[cb * b + i] = ge; }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Whole InfoNCE forward in ONE launch for b <= 32 (a batch tile), D <= 128: logits, row log-sum-exp, top-k rank and the loss.
//
// Persistent waves: wave w of the launch owns the 32-row queue blocks w, w + nwaves, ...; the tile of the NEXT block is in
// flight (16 float4 per lane) while the MFMAs of the current one run out of a wave-private LDS tile (no workgroup barrier in
// the loop).  The soft-max statistics are kept PER LANE across all blocks of the wave -- running reference, sum of
// exponentials and the count of negatives >= the positive, for each of the lane's 16 batch rows -- so the cross-lane
// reductions (DPP) run once per wave, not once per block.  Workgroups then publish one (max, sum, count) triple per batch
// row with write-through stores and draw a ticket from a device-scope counter; the workgroup that draws the last ticket
// folds all partials (L1-bypassing loads), writes lse / rank / loss and re-arms the counter (cdna guide, Guideline 16:
// counter form, sc1 payload).  `counter` must be zero before the first call and is left zero by every call.
// ---------------------------------------------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void moco_logits_persist_kernel(
    const float* __restrict__ q, const float* __restrict__ kpos, const float* __restrict__ queue,
    int b, long long K, int D, float inv_T, float* __restrict__ logits, int ncb,
    float* part, unsigned* counter, float* __restrict__ lse, int* __restrict__ rank, float* __restrict__ loss, gca_magic md4) {
  constexpr int NT = FD / 8;
  __shared__ __attribute__((aligned(16))) float Qs[32 * FP];
  __shared__ __attribute__((aligned(16))) float Ns[WAVES][32 * FP];
  __shared__ float L0[32];
  __shared__ float Red[2 * WAVES][3][32];
  __shared__ int last_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lh = lane >> 5, ll = lane & 31;
  const int nwg = gridDim.x, nwaves = nwg * WAVES, gw = blockIdx.x * WAVES + wave;
  const long long ld = K + 1;
  const int d4 = D >> 2, nt = D >> 3;
  float* Nw = Ns[wave];

  // first queue tile of this wave: in flight while q is staged.  Every load of the kernel is UNCONDITIONAL: rows past the
  // end go through the buffer resource's range check (all-ones offset -> 0.0), never through a branch -- a predicated load
  // makes hipcc branch around it and wait for each one in turn (16 dependent HBM round trips per tile = 25 us).
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(queue), 0, (unsigned)(K * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q), 0, (unsigned)(b * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kpos), 0, (unsigned)(b * D * 4), 0x00020000);
  auto ld4 = [](const __amdgpu_buffer_rsrc_t& rs, unsigned off) __attribute__((always_inline)) {
    const f32x4v f = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    return make_float4(f.x, f.y, f.z, f.w);
  };
  float4 v[NT];
  unsigned toff[NT];                  // byte offset of piece `it` inside a 32-row tile; all-ones for pieces past 32*d4
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int i = lane + 64 * it, r = (int)gca_fdiv((unsigned)i, md4), c4 = i - r * d4;
    toff[it] = i < 32 * d4 ? (unsigned)(r * D + c4 * 4) * 4u : 0xffffffffu;
  }
  auto fetch = [&](long long cb) __attribute__((always_inline)) {
    const unsigned base = (unsigned)(cb * 32 * D * 4);              // rows past K fall outside the resource: zeros
#pragma unroll
    for (int it = 0; it < NT; ++it) v[it] = ld4(rq, toff[it] == 0xffffffffu ? 0xffffffffu : base + toff[it]);
  };
  long long cb = gw;
  fetch(cb < ncb ? cb : 0);

  {
    constexpr int QPT = (32 * (FD / 4) + WAVES * 64 - 1) / (WAVES * 64);      // q pieces per thread
    float4 qv[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
      const int i = tid + u * WAVES * 64;
      const int m = (int)gca_fdiv((unsigned)i, md4), c4 = i - m * d4;
      qv[u] = ld4(rb, i < 32 * d4 ? (unsigned)(m * D + c4 * 4) * 4u : 0xffffffffu);      // rows >= b: outside the resource
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
      const int i = tid + u * WAVES * 64;
      const int m = (int)gca_fdiv((unsigned)i, md4), c4 = i - m * d4;
      if (i < 32 * d4) *reinterpret_cast<float4*>(&Qs[m * FP + c4 * 4]) = qv[u];
    }
  }
  __syncthreads();
  {   // positive logits: 32 rows over WAVES waves, 64/(32/WAVES) lanes per row; all key loads up front
    constexpr int RPW = WAVES >= 32 ? 1 : 32 / WAVES;
    constexpr int LPR = 64 / RPW;
    constexpr int PPL = (FD / 4 + LPR - 1) / LPR;                   // float4 pieces per lane
    const int m = wave * RPW + lane / LPR, part_ = lane % LPR;
    float4 kq[PPL];
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
      const int c4 = part_ + u * LPR;
      kq[u] = ld4(rk, c4 < d4 ? (unsigned)(m * D + c4 * 4) * 4u : 0xffffffffu);
    }
    float sdot = 0.f;
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
      const int c4 = part_ + u * LPR;
      if (c4 < d4) {
        const float4 a = *reinterpret_cast<const float4*>(&Qs[m * FP + c4 * 4]);
        sdot += a.x * kq[u].x + a.y * kq[u].y + a.z * kq[u].z + a.w * kq[u].w;
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
    if (part_ == 0) L0[m] = sdot * inv_T;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid < 32 && tid < b) logits[(long long)tid * ld] = L0[tid];

  // per-lane statistics of the lane's 16 batch rows (row of register r: m = (r&3) + 8*(r>>2) + 4*lh)
  float l0r[16], mx[16], sm[16], cn[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    l0r[r] = L0[(r & 3) + 8 * (r >> 2) + 4 * lh];
    mx[r] = l0r[r]; sm[r] = 0.f; cn[r] = 0.f;
  }

  for (; cb < ncb; cb += nwaves) {
    // registers -> this wave's LDS tile.  DS operations of one wave execute in order, so the fragment reads below see it
    // and the next iteration's stores come after this iteration's reads: no barrier, no wait beyond the data dependences.
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int i = lane + 64 * it, r = (int)gca_fdiv((unsigned)i, md4), c4 = i - r * d4;
      if (i < 32 * d4) *reinterpret_cast<float4*>(&Nw[r * FP + c4 * 4]) = v[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const long long nxt = cb + nwaves;
    fetch(nxt < ncb ? nxt : cb);                   // unconditional (clamped): the next tile streams in behind the MFMAs
    __builtin_amdgcn_wave_barrier();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int t = 0; t < nt; ++t) {
      const float4 a = *reinterpret_cast<const float4*>(&Qs[ll * FP + 8 * t + 4 * lh]);
      const float4 bq = *reinterpret_cast<const float4*>(&Nw[ll * FP + 8 * t + 4 * lh]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, acc, 0, 0, 0);
    }
    const long long j = cb * 32 + ll;
    const bool jv = j < K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float x = acc[r] * inv_T;
      if (jv && i < b) {
        logits[(long long)i * ld + 1 + j] = x;
        float d = x - mx[r];
        if (d > 24.f) { sm[r] *= __expf(-d); mx[r] = x; d = 0.f; }       // rare: keeps exp(d) far from overflow
        sm[r] += __expf(d);
        cn[r] += x >= l0r[r] ? 1.f : 0.f;
      }
    }
  }

  // ---- one cross-lane fold per wave, then per workgroup
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float mh = half_wave_max_hi(mx[r]);
    const float mxa = lh ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mh), 63))
                         : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mh), 31));
    const float se = half_wave_sum_hi_f(sm[r] * __expf(mx[r] - mxa));
    const float ce = half_wave_sum_hi_f(cn[r]);
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (ll == 31) { Red[wave][0][m] = mxa; Red[wave][1][m] = se; Red[wave][2][m] = ce; }
  }
  __syncthreads();
  if (tid < 32) {
    float M = Red[0][0][tid];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) M = fmaxf(M, Red[w][0][tid]);
    float S = 0.f, C = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { S += Red[w][1][tid] * __expf(Red[w][0][tid] - M); C += Red[w][2][tid]; }
    // write-through (sc1) stores: visible to every other CU once this wave's vmcnt has drained -- no release fence
    __hip_atomic_store(&part[(0 * nwg + blockIdx.x) * 32 + tid], M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&part[(1 * nwg + blockIdx.x) * 32 + tid], S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&part[(2 * nwg + blockIdx.x) * 32 + tid], C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains before the ticket is drawn
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = ticket == (unsigned)nwg - 1u;
  }
  __syncthreads();
  if (!last_flag) return;
  // ---- last workgroup: fold every workgroup's partials (row = tid & 31; the 2*WAVES slices stride over the workgroups).
  // The loads bypass L1 (sc1), which is what makes them see the other CUs' write-through stores without an acquire fence;
  // sixteen workgroups' triples are in flight per thread (issued one dependent round trip at a time they cost 100 us for
  // 256 workgroups).  Measured and not kept: agent-scope acquire + plain loads -- the fence alone cost 5-8 us here.
  {
    const int row = tid & 31, slice = tid >> 5;
    float M = -INFINITY, S = 0.f, C = 0.f;
    constexpr int UF = 16;
    for (int g0 = slice; g0 < nwg; g0 += 2 * WAVES * UF) {
      float pm_[UF], ps_[UF], pc_[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        const int g = g0 + u * 2 * WAVES;
        const int gc = g < nwg ? g : slice;                  // clamped: every load is issued, the surplus is ignored
        pm_[u] = __hip_atomic_load(&part[(0 * nwg + gc) * 32 + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ps_[u] = __hip_atomic_load(&part[(1 * nwg + gc) * 32 + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pc_[u] = __hip_atomic_load(&part[(2 * nwg + gc) * 32 + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        if (g0 + u * 2 * WAVES < nwg) {
          const float Mn = fmaxf(M, pm_[u]);
          S = S * (M > -INFINITY ? __expf(M - Mn) : 0.f) + ps_[u] * __expf(pm_[u] - Mn);
          M = Mn; C += pc_[u];
        }
      }
    }
    Red[slice][0][row] = M; Red[slice][1][row] = S; Red[slice][2][row] = C;
  }
  __syncthreads();
  if (tid < 32) {
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 2 * WAVES; ++w) M = fmaxf(M, Red[w][0][tid]);
    float S = 0.f, C = 0.f;
#pragma unroll
    for (int w = 0; w < 2 * WAVES; ++w) {
      if (Red[w][0][tid] > -INFINITY) S += Red[w][1][tid] * __expf(Red[w][0][tid] - M);
      C += Red[w][2][tid];
    }
    const float l0 = L0[tid];
    const float Mf = fmaxf(M, l0);
    const float lsev = Mf + logf(S * __expf(M - Mf) + __expf(l0 - Mf));       // + the positive column
    const bool rv = tid < b;
    if (rv && lse) lse[tid] = lsev;
    if (rv && rank) rank[tid] = (int)(C + 0.5f);
    float term = rv ? lsev - l0 : 0.f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) term += __shfl_xor(term, o, 64);
    if (tid == 0 && loss) *loss = term / (float)b;
    if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // re-arm for the next call
  }
}

// one workgroup per batch row: fold the column-block partials and the positive column
__global__ __launch_bounds__(256) void lse_finish_kernel(const float* __restrict__ logits, long long ld, int ncb,
                                                         const float* __restrict__ pm, const float* __restrict__ ps,
                                                         const float* __restrict__ pc, float* __restrict__ lse,
                                                         int* __restrict__ rank) {
  __shared__ float sh[4];
  __shared__ float shm[4];
  const long long i = blockIdx.x;
  const float l0 = logits[i * ld];
  float m = l0;
  const long long nb = gridDim.x;                     // partials are [column block][batch row]
  for (int c = threadIdx.x; c < ncb; c += 256) m = fmaxf(m, pm[(long long)c * nb + i]);
  m = gca_wave_max(m);
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
  float s = 0.f, c_ = 0.f;
  for (int c = threadIdx.x; c < ncb; c += 256) {
    const float pmv = pm[(long long)c * nb + i];
    if (pmv > -INFINITY) s += ps[(long long)c * nb + i] * expf(pmv - m);
    c_ += pc[(long long)c * nb + i];
  }
  s = gca_block_sum256(s, sh);
  c_ = gca_block_sum256(c_, sh);
  if (threadIdx.x == 0) {
    if (lse) lse[i] = m + logf(s + expf(l0 - m));
    if (rank) rank[i] = (int)(c_ + 0.5f);
  }
}

// One workgroup per batch row: lse_i = logsumexp(logits_i), rank_i = #{j>=1 : l_ij >= l_i0}
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ logits, long long ncol,
                                                        float* __restrict__ lse, int* __restrict__ rank) {
  __shared__ float sh[4];
  __shared__ float shm[4];
  const float* row = logits + (long long)blockIdx.x * ncol;
  float m = -INFINITY;
  for (long long j = threadIdx.x; j < ncol; j += 256) m = fmaxf(m, row[j]);
  m = gca_wave_max(m);
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
  const float l0 = row[0];
  float s = 0.f, c = 0.f;
  for (long long j = threadIdx.x; j < ncol; j += 256) {
    const float v = row[j];
    s += expf(v - m);
    if (j >= 1 && v >= l0) c += 1.f;
  }
  s = gca_block_sum256(s, sh);
  c = gca_block_sum256(c, sh);
  if (threadIdx.x == 0) {
    if (lse) lse[blockIdx.x] = m + logf(s);
    if (rank) rank[blockIdx.x] = (int)c;
  }
}

__global__ __launch_bounds__(256) void nce_loss_finish_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                              int b, long long ncol, float* __restrict__ loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < b; i += 256) s += lse[i] - logits[(long long)i * ncol];
  s = gca_block_sum256(s, sh);
  if (threadIdx.x == 0) *loss = s / (float)b;
}

__global__ __launch_bounds__(256) void nce_loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                           int b, long long ncol, const float* __restrict__ gs_dev,
                                                           float gs_host, float* __restrict__ dl) {
  const float g = (gs_dev ? *gs_dev : 1.f) * gs_host / (float)b;
  const long long total = (long long)b * ncol;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / ncol, j = i - r * ncol;
    float v = expf(logits[i] - lse[r]);
    if (j == 0) v -= 1.f;
    dl[i] = g * v;
  }
}

// dq partial over a slice of RS queue rows: slab[slice][i][d] = sum_{j in slice} g[i,1+j] * queue[j,d]
// grid (slices, D/128, b/32); 4 waves = 4 x 32 feature columns.
constexpr int BWD_RS = 128;
__global__ __launch_bounds__(256) void moco_dq_kernel(
    const float* __restrict__ dl, const float* __restrict__ logits, const float* __restrict__ lse,
    const float* __restrict__ gs_dev, float gs_host, const float* __restrict__ queue, int b, long long K, int D,
    long long ov_start, const long long* __restrict__ ov_start_dev, long long ov_n,
    const float* __restrict__ ov_rows, float* __restrict__ slab) {
  __shared__ float Gs[BWD_RS][33];
  if (ov_start_dev) ov_start = *ov_start_dev;      // device-resident queue pointer (graph replay)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lh = lane >> 5, ll = lane & 31;
  const long long j0 = (long long)blockIdx.x * BWD_RS;
  const int dcol = blockIdx.y * 128 + wave * 32 + ll;
  const int mt = blockIdx.z;
  const long long ld = K + 1;
  const float g = (gs_dev ? *gs_dev : 1.f) * gs_host / (float)b;
  // stage G^T: lanes along j (coalesced), one batch row per step; all 16 loads of a thread in flight (unconditional buffer
  // loads: see below)
  {
    const float* gsrc = dl ? dl : logits;
    const long long gb = (long long)b * ld * 4;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0,
                                                                        gb > 0xfffff000LL ? 0xfffff000u : (unsigned)gb, 0x00020000);
    constexpr int NG = 32 * BWD_RS / 256;
    float gv[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int i = tid + 256 * u;
      const int m = i / BWD_RS, jj = i % BWD_RS;
      const int row = mt * 32 + m;
      const long long j = j0 + jj;
      const bool ok = row < b && j < K;
      gv[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, ok ? (int)(unsigned)(((long long)row * ld + 1 + j) * 4) : -1, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int i = tid + 256 * u;
      const int m = i / BWD_RS, jj = i % BWD_RS;
      const int row = mt * 32 + m;
      const bool ok = row < b && j0 + jj < K;
      float v = gv[u];
      if (!dl) v = ok ? g * expf(v - lse[min(row, b - 1)]) : 0.f;
      Gs[jj][m] = v;
    }
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool dv = dcol < D;
  // UNCONDITIONAL loads through buffer resources (an out-of-range / not-wanted element is an all-ones offset the hardware
  // zero-fills): a load inside a branch makes hipcc drain vmcnt(0) at every use, i.e. one exposed HBM round trip per row
  // pair -- this loop took 154 us at K = 65536 that way.  Both candidates of a row (queue / saved pre-enqueue copy) are
  // fetched, the copy's offset is valid only inside the overwritten window.
  const unsigned qbytes = K * (long long)D * 4 > 0xfffff000LL ? 0xfffff000u : (unsigned)(K * D * 4);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(queue), 0, qbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ov_rows ? ov_rows : queue), 0,
                                                                      ov_rows ? (unsigned)(ov_n * D * 4) : 0u, 0x00020000);
#pragma unroll 8
  for (int kk = 0; kk < BWD_RS; kk += 2) {
    const long long j = j0 + kk + lh;
    long long rel = j - ov_start; if (rel < 0) rel += K;
    const bool in = dv && j < K, ov = in && rel < ov_n;
    const unsigned vq = in ? (unsigned)((j * D + dcol) * 4) : 0xffffffffu;
    const unsigned vo = ov ? (unsigned)((rel * D + dcol) * 4) : 0xffffffffu;
    const float qv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rq, (int)vq, 0, 0));
    const float sv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, (int)vo, 0, 0));
    const float bb = ov ? sv : qv;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[kk + lh][ll], bb, acc, 0, 0, 0);
  }
  if (dv) {
    float* out = slab + (long long)blockIdx.x * b * D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (i < b) out[(long long)i * D + dcol] = acc[r];
    }
  }
}

// dq[e] = (sum over slices of slab[k][e] + g0 * kpos[e]) / T.  A block finishes 16 elements: 16 groups of 16 threads each
// sum the slices k = group, group + 16, ... (loads 8 deep in flight: the K = 65536 queue makes 512 slices, and one thread
// walking all of them was 100+ us of exposed latency), then the 16 group sums are added in group order -- a fixed tree,
// so the result is deterministic.
__global__ __launch_bounds__(256) void moco_dq_finish_kernel(
    const float* __restrict__ slab, int slices, const float* __restrict__ dl, const float* __restrict__ logits,
    const float* __restrict__ lse, const float* __restrict__ gs_dev, float gs_host, const float* __restrict__ kpos,
    int b, long long K, int D, float inv_T, float* __restrict__ dq) {
  __shared__ float part[16][17];
  const long long total = (long long)b * D;
  const long long ld = K + 1;
  const float g = (gs_dev ? *gs_dev : 1.f) * gs_host / (float)b;
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const long long e = (long long)blockIdx.x * 16 + el;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(slab), 0,
                                                                      (unsigned)((long long)slices * total * 4), 0x00020000);
  float s = 0.f;
  for (int k0 = grp; k0 < slices; k0 += 16 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 16 * u;
      v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (k < slices && e < total) ? (int)(unsigned)(((long long)k * total + e) * 4) : -1, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  part[grp][el] = s;
  __syncthreads();
  if (grp == 0 && e < total) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part[q][el];
    const long long i = e / D;
    float g0;
    if (dl) g0 = dl[i * ld];
    else g0 = g * (expf(logits[i * ld] - lse[i]) - 1.f);
    dq[e] = (t + g0 * kpos[e]) * inv_T;
  }
}

__global__ __launch_bounds__(256) void enqueue_kernel(float* __restrict__ queue, long long K, int D,
                                                      const float* __restrict__ keys, long long n, long long ptr,
                                                      const long long* __restrict__ ptr_dev,
                                                      float* __restrict__ saved) {
  if (ptr_dev) ptr = *ptr_dev;                     // device-resident queue pointer (graph replay)
  const long long total = n * D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i = e / D, d = e - i * D;
    const long long row = (ptr + i) % K;
    if (saved) saved[e] = queue[row * D + d];
    queue[row * D + d] = keys[e];
  }
}

__global__ void queue_advance_kernel(long long* ptr_dev, long long n, long long K) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *ptr_dev = (*ptr_dev + n) % K;
}


// accuracy(output, target, topk) of lib/evaluation/metric.py:44-67 without its top-k sort: the target column is among the
// top k of its row iff fewer than k OTHER columns score >= it (the same ">= the positive" counter the logits kernel fuses
// for label 0).  One workgroup per row.
__global__ __launch_bounds__(256) void rank_ge_kernel(const float* __restrict__ out, const long long* __restrict__ target,
                                                      long long ncol, int* __restrict__ rank) {
  __shared__ float sh[4];
  const long long i = blockIdx.x;
  const long long t = target[i];
  const float* row = out + i * ncol;
  const float ref = row[t];
  float c = 0.f;
  for (long long j = threadIdx.x; j < ncol; j += 256) c += (j != t && row[j] >= ref) ? 1.f : 0.f;
  c = gca_block_sum256(c, sh);
  if (threadIdx.x == 0) rank[i] = (int)(c + 0.5f);
}
}  // namespace

extern "C" {

int64_t gca_infonce_ws_bytes(int64_t b, int64_t K) {
  if (b <= 0 || K <= 0) return GCA_EINVAL;
  // dq partial slabs: one (b x D) slab per BWD_RS-row queue slice, sized for the D <= 256 the
  // backward kernel accepts.
  const int64_t dq = (int64_t)sizeof(float) * gca_ceil_div(K, BWD_RS) * b * 256;
  const int64_t fwd = (int64_t)sizeof(float) * 3 * 32 * 2048;      // forward partials: (max, sum, count) x 32 rows x <= 2048 workgroups
  return dq > fwd ? dq : fwd;
}

int gca_moco_logits_fwd(const float* q, const float* k, const float* queue, int64_t b, int64_t K,
                        int64_t D, float inv_T, float* logits, float* row_lse, int32_t* rank_ge, float* loss,
                        void* ws, uint32_t* sync_counter, void* stream) {
  if (!q || !k || !queue || !logits || b <= 0 || K <= 0 || D <= 0 || (D & 3)) return GCA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool want = row_lse || rank_ge || loss;
  if (loss && !row_lse) return GCA_EINVAL;          // the loss needs the row log-sum-exps (kept for the backward anyway)
  static int persist_on = -1;
  if (persist_on < 0) { const char* e = getenv("GCA_NCE_PERSIST"); persist_on = (e && e[0] == '0') ? 0 : 1; }     // A/B runs
  if (persist_on && b <= 32 && D <= FD && (D & 7) == 0 && ws && sync_counter && K * D * 4 < 0xfffff000LL &&
      (((uintptr_t)q | (uintptr_t)queue | (uintptr_t)k) % 16) == 0) {
    // single launch: persistent waves + last-arriving workgroup folds the statistics (moco_logits_persist_kernel)
    const int ncb = (int)gca_ceil_div(K, 32);
    static int env_waves = -1, env_grid = -1;
    if (env_waves < 0) {
      const char* e = getenv("GCA_NCE_WAVES"); env_waves = e ? atoi(e) : 0;
      const char* g = getenv("GCA_NCE_GRID"); env_grid = g ? atoi(g) : 0;
    }
    // enough waves to cover the queue blocks, at most 4 per CU: past 1024 blocks every wave streams several tiles
    int waves = ncb >= 64 ? 4 : (ncb >= 16 ? 2 : 1);       // (measured: 4 waves per workgroup at K = 4096 and at K = 65536)
    if (env_waves == 1 || env_waves == 2 || env_waves == 4 || env_waves == 8) waves = env_waves;
    long long grid = gca_ceil_div(ncb, waves);
    if (grid > 256) grid = 256;
    if (env_grid > 0) grid = env_grid;
    const gca_magic md4 = gca_make_magic((unsigned)(D >> 2));
    float* part = reinterpret_cast<float*>(ws);
#define GCA_PERSIST(W) hipLaunchKernelGGL((moco_logits_persist_kernel<W>), dim3((unsigned)grid), dim3(64 * W), 0, st, q, k, queue, \
                                          (int)b, (long long)K, (int)D, inv_T, logits, ncb, part, sync_counter, row_lse, rank_ge, loss, md4)
    if (waves == 8) GCA_PERSIST(8);
    else if (waves == 4) GCA_PERSIST(4);
    else if (waves == 2) GCA_PERSIST(2);
    else GCA_PERSIST(1);
#undef GCA_PERSIST
    return gca_launch_status();
  }
  if (D <= FD && (D & 7) == 0 && (!want || ws) && (((uintptr_t)q | (uintptr_t)queue | (uintptr_t)k) % 16) == 0) {
    // fused path: logits + LSE / rank partials in one pass over the queue, then a per-row fold
    const int ncb = (int)gca_ceil_div(K, 32);
    // waves per workgroup: the kernel is one dependent chain per wave (load -> stage -> MFMA -> reduce), so what counts
    // is running ALL column blocks in one round: as many waves per workgroup as it takes to fit 256 workgroups
    const int waves = ncb > 1024 ? 8 : (ncb > 512 ? 4 : (ncb > 256 ? 2 : 1));
    float* pm = want ? reinterpret_cast<float*>(ws) : nullptr;
    float* ps = want ? pm + (long long)b * ncb : nullptr;
    float* pc = want ? ps + (long long)b * ncb : nullptr;
    const gca_magic md4 = gca_make_magic((unsigned)(D >> 2));
    const dim3 grid((unsigned)gca_ceil_div(ncb, waves));
#define GCA_FUSED(W) hipLaunchKernelGGL((moco_logits_fused_kernel<W>), grid, dim3(64 * W), 0, st, q, k, queue, (int)b, \
                                        (long long)K, (int)D, inv_T, logits, ncb, pm, ps, pc, md4)
    if (waves == 8) GCA_FUSED(8);
    else if (waves == 4) GCA_FUSED(4);
    else if (waves == 2) GCA_FUSED(2);
    else GCA_FUSED(1);
#undef GCA_FUSED
    int rc = gca_launch_status();
    if (rc || !want) return rc;
    // (a single-workgroup fold that also reduces the loss was tried for b*K small: 8 us slower than these two launches)
    hipLaunchKernelGGL(lse_finish_kernel, dim3((unsigned)b), dim3(256), 0, st, logits, (long long)(K + 1), ncb, pm, ps, pc,
                       row_lse, rank_ge);
    rc = gca_launch_status();
    if (rc || !loss) return rc;
    hipLaunchKernelGGL(nce_loss_finish_kernel, dim3(1), dim3(256), 0, st, logits, row_lse, (int)b, (long long)(K + 1), loss);
    return gca_launch_status();
  }
  if (K >= 32768)
    hipLaunchKernelGGL((moco_logits_kernel<4>), dim3((unsigned)gca_ceil_div(K, 128)), dim3(256), 0, st, q, k, queue,
                       (int)b, (long long)K, (int)D, inv_T, logits);
  else
    hipLaunchKernelGGL((moco_logits_kernel<1>), dim3((unsigned)gca_ceil_div(K, 32)), dim3(64), 0, st, q, k, queue,
                       (int)b, (long long)K, (int)D, inv_T, logits);
  int rc = gca_launch_status();
  if (rc) return rc;
  if (row_lse || rank_ge) {
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)b), dim3(256), 0, st, logits, (long long)(K + 1), row_lse, rank_ge);
    rc = gca_launch_status();
  }
  if (!rc && loss) {
    hipLaunchKernelGGL(nce_loss_finish_kernel, dim3(1), dim3(256), 0, st, logits, row_lse, (int)b, (long long)(K + 1), loss);
    rc = gca_launch_status();
  }
  return rc;
}

int gca_nce_softmax_loss_fwd(const float* logits, int64_t b, int64_t ncol, const float* row_lse_in,
                             float* row_lse_out, float* loss, void* stream) {
  if (!logits || !loss || b <= 0 || ncol <= 0) return GCA_EINVAL;
  if (!row_lse_in && !row_lse_out) return GCA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float* lse = row_lse_in;
  if (!lse) {
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)b), dim3(256), 0, st, logits, (long long)ncol, row_lse_out,
                       (int*)nullptr);
    lse = row_lse_out;
  }
  hipLaunchKernelGGL(nce_loss_finish_kernel, dim3(1), dim3(256), 0, st, logits, lse, (int)b, (long long)ncol, loss);
  return gca_launch_status();
}

int gca_nce_softmax_loss_bwd(const float* logits, const float* row_lse, int64_t b, int64_t ncol,
                             const float* gscale_dev, float gscale_host, float* dlogits, void* stream) {
  if (!logits || !row_lse || !dlogits || b <= 0 || ncol <= 0) return GCA_EINVAL;
  long long blocks = gca_ceil_div(b * ncol, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nce_loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, row_lse,
                     (int)b, (long long)ncol, gscale_dev, gscale_host, dlogits);
  return gca_launch_status();
}

int gca_moco_logits_bwd(const float* dlogits, const float* logits, const float* row_lse,
                        const float* gscale_dev, float gscale_host,
                        const float* k, const float* queue, int64_t b, int64_t K, int64_t D, float inv_T,
                        int64_t ov_start, const int64_t* ov_start_dev, int64_t ov_n, const float* ov_rows,
                        float* dq, void* ws, void* stream) {
  if (!k || !queue || !dq || !ws || b <= 0 || K <= 0 || D <= 0 || D > 256) return GCA_EINVAL;
  if (!dlogits && (!logits || !row_lse)) return GCA_EINVAL;
  if (ov_n < 0 || ov_n > K || (ov_n > 0 && !ov_rows) || ov_start < 0 || ov_start >= K) return GCA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (K * D * 4 >= 0xfffff000LL || b * (K + 1) * 4 >= 0xfffff000LL) return GCA_EINVAL;      // 32-bit byte offsets in the kernel
  const int slices = (int)gca_ceil_div(K, BWD_RS);
  float* slab = reinterpret_cast<float*>(ws);
  dim3 grid((unsigned)slices, (unsigned)gca_ceil_div(D, 128), (unsigned)gca_ceil_div(b, 32));
  hipLaunchKernelGGL(moco_dq_kernel, grid, dim3(256), 0, st, dlogits, logits, row_lse, gscale_dev, gscale_host, queue,
                     (int)b, (long long)K, (int)D, (long long)ov_start, (const long long*)ov_start_dev, (long long)ov_n, ov_rows,
                     slab);
  int rc = gca_launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(moco_dq_finish_kernel, dim3((unsigned)gca_ceil_div(b * D, 16)), dim3(256), 0, st, slab, slices,
                     dlogits, logits, row_lse, gscale_dev, gscale_host, k, (int)b, (long long)K, (int)D, inv_T, dq);
  return gca_launch_status();
}

int gca_queue_enqueue(float* queue, int64_t K, int64_t D, const float* keys, int64_t n, int64_t ptr,
                      const int64_t* ptr_dev, float* saved_rows, void* stream) {
  if (!queue || !keys || K <= 0 || D <= 0 || n <= 0 || n > K || ptr < 0 || ptr >= K) return GCA_EINVAL;
  long long blocks = gca_ceil_div(n * D, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(enqueue_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, queue, (long long)K,
                     (int)D, keys, (long long)n, (long long)ptr, (const long long*)ptr_dev, saved_rows);
  return gca_launch_status();
}

int gca_rank_ge(const float* output, const int64_t* target, int64_t b, int64_t ncol, int32_t* rank_ge, void* stream) {
  if (!output || !target || !rank_ge || b <= 0 || ncol <= 0 || ncol >= (1LL << 24)) return GCA_EINVAL;   // (exact fp32 counts)
  hipLaunchKernelGGL(rank_ge_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, output, (const long long*)target,
                     (long long)ncol, rank_ge);
  return gca_launch_status();
}

int gca_queue_advance(int64_t* ptr_dev, int64_t n, int64_t K, void* stream) {
  if (!ptr_dev || n < 0 || K <= 0) return GCA_EINVAL;
  hipLaunchKernelGGL(queue_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)ptr_dev, (long long)n,
                     (long long)K);
  return gca_launch_status();
}

}  // extern "C"
