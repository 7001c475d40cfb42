// 3D convolution for gfx950 as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 64 FLOP/clk/SIMD).
//
//   forward : Y[n,ko,o]   = sum_{c,tap} W[ko,c,tap] * X[n,c,o*s-p+tap]        M=Ko   N=n*o      K=c*tap
//   dgrad   : dX[n,c,i]   = sum_{ko,tap} W[ko,c,tap] * dY[n,ko,(i+p-tap)/s]   M=C    N=n*i      K=ko*tap
//   (wgrad lives in conv3d_wgrad.hip)
//
// Activations stay NCDHW: for a fixed GEMM-K row (channel, tap) the GEMM-N direction is the
// W axis of the image, so a wave's 64 lanes read consecutive floats (coalesced along W).
// Nothing is materialised: the im2col window is gathered straight into LDS through a small
// per-row table (element offset + tap id).  Window bounds are resolved ONCE per thread into a
// 64-bit tap-validity mask (its GEMM columns are fixed for the whole K loop), so the gather in the
// hot loop is: bit test, index select, load -- no branches, no per-element compares.
//
// Every pass is expressed as one or more LINEAR gather problems ("classes"):
//   forward            one class:  src = q*stride - pad + tap
//   unit-stride dgrad  one class:  src = q + pad - tap
//   strided dgrad      one class per output-position residue rho (i = s*q + rho): only the taps with
//                      (rho + pad - tap) % s == 0 contribute and src = q + (rho + pad - tap)/s, so no
//                      MFMA work is spent on structurally-zero products and nothing is divided in the loop.
// Convolutions that are pointwise in space (kh = kw = 1: the (k,1,1) temporal convs and 1x1x1, 44 % of
// R(2+1)D's FLOPs) take the VEC variant: a GEMM-K row is one contiguous run of the image, gathered with
// float4 loads into a 64 x 256 tile.  The weight operand is pre-packed k-major and zero padded
// (gca_conv_pack) so its tile loads are unpredicated float4.  Problems whose tile grid cannot fill
// 256 CUs split the K loop over workgroups (fp32 partial slabs + a deterministic finishing pass that
// also emits the BatchNorm sums).
//
// Reference call sites replaced: every nn.Conv3d / nn.Linear on the path (see include/gca_hip.h).
#include <cstring>
#include <type_traits>
#include "conv_halo.h"

#include <cstdlib>
#include <vector>

using namespace gca_conv;

namespace {


// ---------------------------------------------------------------------------------------------
// 256 threads = 4 waves side by side along N: every wave owns all BM = 32*TM rows of the block tile and
// BN/4 of its columns (TN = BN/128 MFMA tiles wide).  LDS tiles are stored [row][k] with an 80-byte row
// pitch, so a lane fetches FOUR k-steps of its operand row with one conflict-free ds_read_b128 and a
// k-tile of 16 is 2 reads per operand tile followed by an uninterrupted chain of 8*TM*TN MFMAs.  (A lane of
// half h holds k = 8t + 4h + e of its row; MFMA step (t, e) therefore sums k = 8t+e and 8t+4+e -- the same
// permutation on both operands.)
// FAST: 1 = tap-mask path with <= 31 taps (one v_bfe_i32 per gathered element), 2 = <= 62 taps, 0 = window
// tests per element (huge kernels, e.g. 7x7x7).  VEC: float4 gathers (BN = 256, every lane owns 4 consecutive
// columns, a wave covers one whole tile row per load).
// ---------------------------------------------------------------------------------------------
constexpr int LDK = BK + 4;     // LDS row pitch in floats (80 B)
// pitch by arithmetic mode: a row of 16 k is 64 B as fp32 or as bf16 hi+lo pairs, 96 B as bf16 hi+mid+lo (+16 B pad each:
// 20 r mod 64 and 28 r mod 64 both walk all 16 four-bank groups over 16 rows, so the b128 fragment reads are conflict-free)
constexpr int lds_pitch(int math) { return math == 2 ? 28 : (math == 3 ? 12 : LDK); }   // (3: 32-byte rows of halves + 16 B: 3 r mod 8 walks all eight 16-byte bank groups)
// LDS buffers of the k-loop: the tall / 256-column bf16x6 tiles (96-byte rows) keep ONE (the next tile waits in registers
// anyway; a second barrier per k-tile) so that 3-5 workgroups stay resident per CU instead of 2 and one workgroup's
// split/store phase overlaps another's MFMAs
constexpr int lds_bufs(int tm, bool vec, int math) { return (math >= 1 || vec || tm >= 4) ? 1 : 2; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// The tile body is a device function so that ONE launch can mix two tile heights (conv_igemm_2phase_kernel below):
// `bid` is the workgroup's logical id inside its phase, As_/Bs_/taptab the workgroup's LDS (sized by the caller).
template <int TM, int BN, int FAST, bool VEC, int MATH, bool H = false>
__device__ __forceinline__ void igemm_tile(
    const void* __restrict__ src, const float* __restrict__ apack, const int2* __restrict__ table,
    const float* __restrict__ bias, void* __restrict__ dst, float* __restrict__ psum,
    float* __restrict__ psq, float* __restrict__ slab, const IgemmParams p, int bid,
    float* __restrict__ As_, float* __restrict__ Bs_, int* __restrict__ taptab) {
  // H: fp16 STORAGE -- the gathered tensor and the destination hold _Float16; every element is widened on load (an fp16
  // value is exactly hi + lo in the bf16x3 split, so the split-product arithmetic loses nothing of the operands) and the
  // fp32 accumulator is rounded once on store.  Weights (packed fp32), statistics and split-K slabs stay fp32.
  constexpr unsigned ES = H ? 2u : 4u;
  constexpr int BM = 32 * TM, TN = BN / 128;
  constexpr int A_F4 = (BM * 4 + 255) / 256;               // float4 loads per thread for the A tile (BM x 16)
  constexpr int B_PER = VEC ? 4 : 8;                       // gathers per thread for the B tile
  static_assert(!VEC || (BN == 256 && FAST == 1), "VEC variant: 256 columns, <= 32-tap mask path");
  static_assert(BN == 128 || BN == 256, "BN");

  constexpr int LDP = lds_pitch(MATH), NP = math_parts(MATH);                // row pitch (floats), bf16 parts per element
  float (*As)[BM][LDP] = reinterpret_cast<float (*)[BM][LDP]>(As_);          // [2][BM][LDP]
  float (*Bs)[BN][LDP] = reinterpret_cast<float (*)[BN][LDP]>(Bs_);          // [2][BN][LDP]

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int split = bid % p.splits; bid /= p.splits;
  const int tileM = bid % p.tilesM, tileN = p.tileN_off + bid / p.tilesM;

  if (FAST) {
    if (tid < 64) taptab[tid] = reinterpret_cast<const int*>(table + p.Kpad)[tid];
    __syncthreads();
  }

  // ---- per-thread column (GEMM-N) state: fixed for the whole K loop
  const int col = VEC ? 4 * lane : tid % BN;
  const int r0 = VEC ? wn * B_PER : (tid / BN) * B_PER;    // first B k-row of this thread inside a k-tile (wave-uniform)
  const long long ng = (long long)tileN * BN + col;
  const bool cvalid = ng < p.Ntot;
  const int QSP = p.QD * p.QH * p.QW, QHW = p.QH * p.QW;
  const int SHW = p.SH * p.SW;
  int id0, ih0, iw0, colbase;                              // 32-bit element index (tensors are < 2^30 elements)
  {
    const long long ngc = cvalid ? ng : 0;
    const int img = (int)(ngc / QSP);
    const int sp = (int)(ngc - (long long)img * QSP);
    const int qd = sp / QHW, r = sp - qd * QHW;
    const int qh = r / p.QW, qw = r - qh * p.QW;
    id0 = qd * p.m_d + p.o_d; ih0 = qh * p.m_h + p.o_h; iw0 = qw * p.m_w + p.o_w;
    colbase = (int)((long long)img * p.src_nstride) + id0 * SHW + ih0 * p.SW + iw0;
  }
  const bool chkD = p.chk & 1, chkH = p.chk & 2, chkW = p.chk & 4;

  // Gathers go through a buffer resource: an out-of-range offset returns 0 in hardware, so an invalid
  // window element costs nothing but an OR of all-ones into its 32-bit byte offset -- no post-load select.
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, p.src_bytes, 0x00020000);
  const unsigned cb4 = (unsigned)colbase * ES;
  // one gathered element / four consecutive ones at byte offset `voff` of the source, as fp32
  // (MATH 3: the half is NOT widened -- its 16 bits travel in the low half of the "float" until they are packed into LDS)
  auto ld1 = [&](unsigned voff) __attribute__((always_inline)) {
    if constexpr (H && MATH == 3) return __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)voff, 0, 0));
    else if constexpr (H) return (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)voff, 0, 0));
    else return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0));
  };
  auto ld4 = [&](unsigned voff) __attribute__((always_inline)) {
    if constexpr (H && MATH == 3) {
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t u = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, 0, 0));
      const unsigned ux = u.x, uy = u.y;
      return make_float4(__uint_as_float(ux & 0xffffu), __uint_as_float(ux >> 16), __uint_as_float(uy & 0xffffu), __uint_as_float(uy >> 16));
    } else if constexpr (H) {
      typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      const f16x4_t hv = __builtin_bit_cast(f16x4_t, __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, 0, 0)));
      return make_float4((float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w);
    } else {
      // NB: bit_cast the WHOLE vector -- __builtin_bit_cast(float, v[i]) on the elements makes hipcc
      // (ROCm 7.2) narrow the load to one dword and replicate it.
      const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0));
      return make_float4(f.x, f.y, f.z, f.w);
    }
  };

  // tap-INvalidity mask: bit t = 1 -> tap t of this column's window falls outside the source tensor
  // (bits of taps the class does not have, and tap id 63 = padded table row, stay 1)
  unsigned ilo = 0xffffffffu, ihi = 0xffffffffu;
  if (FAST) {
    for (int t = 0; t < p.ntaps; ++t) {
      const int pk = taptab[t];
      const int dd = (pk << 16) >> 24, dh = (pk << 8) >> 24, dw = pk >> 24;
      bool ok = cvalid;
      if (chkD) ok = ok & ((unsigned)(id0 + dd) < (unsigned)p.SD);
      if (chkH) ok = ok & ((unsigned)(ih0 + dh) < (unsigned)p.SH);
      if (chkW) ok = ok & ((unsigned)(iw0 + dw) < (unsigned)p.SW);
      if (t < 32) ilo &= ~((unsigned)ok << t); else ihi &= ~((unsigned)ok << (t - 32));
    }
  }

  float breg[VEC ? 1 : B_PER];
  float4 bvec[VEC ? B_PER : 1];
  // A tile (BM x 16 packed weights, k contiguous): up to three float4 slots per thread, kept in named scalars
  // (an indexed array with a predicated tail makes the compiler park it in scratch).  The tail slot is loaded
  // unconditionally from a clamped index and only its LDS store is predicated.
  static_assert(A_F4 <= 3, "A tile slots");
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
  const float* arow = apack + (long long)tileM * BM * p.Kpad;      // packed weights [Mpad][Kpad], k contiguous
  constexpr bool A_TAIL = (BM * 4) % 256 != 0;                      // last slot only partly populated
  const bool a_last_ok = !A_TAIL || tid + (A_F4 - 1) * 256 < BM * 4;
  auto a_slot = [&](int i) __attribute__((always_inline)) {
    int idx = tid + i * 256;
    if (A_TAIL && idx >= BM * 4) idx = BM * 4 - 1;
    return idx;
  };
  const int ao0 = (a_slot(0) >> 2) * p.Kpad + (a_slot(0) & 3) * 4;
  const int ao1 = A_F4 > 1 ? (a_slot(1) >> 2) * p.Kpad + (a_slot(1) & 3) * 4 : 0;
  const int ao2 = A_F4 > 2 ? (a_slot(2) >> 2) * p.Kpad + (a_slot(2) & 3) * 4 : 0;

  // The next tile's loads are issued in PIECES between the MFMA groups of the current tile (below): a vector
  // instruction costs its 4-cycle issue slot inside a 64-cycle MFMA shadow, while the same instructions bunched
  // in front of the MFMA block leave the matrix pipe idle for their whole latency chain (table fetch ->
  // address arithmetic -> load issue).
  int2 e[B_PER];                                           // table rows of the tile being fetched (wave-uniform)
  auto issue_table = [&](int kt) __attribute__((always_inline)) {
    const int2* trow = table + kt * BK + __builtin_amdgcn_readfirstlane(r0);
#pragma unroll
    for (int i = 0; i < B_PER; ++i) e[i] = trow[i];
  };
  auto issue_a = [&](int kt) __attribute__((always_inline)) {
    a0 = *reinterpret_cast<const float4*>(arow + ao0 + kt * BK);
    if (A_F4 > 1) a1 = *reinterpret_cast<const float4*>(arow + ao1 + kt * BK);
    if (A_F4 > 2) a2 = *reinterpret_cast<const float4*>(arow + ao2 + kt * BK);
  };
  auto issue_gather = [&](int i) __attribute__((always_inline)) {      // i: compile-time constant after unrolling
    int inv;                                              // 0 = valid, -1 = invalid
    if (FAST == 1) {
      inv = __builtin_amdgcn_sbfe((int)ilo, e[i].y, 1);              // bit (y & 31) of the mask, sign extended
    } else if (FAST == 2) {
      const unsigned hsel = (unsigned)-((e[i].y >> 5) & 1);            // all-ones for taps 32..62
      const unsigned m = ilo ^ ((ilo ^ ihi) & hsel);
      inv = __builtin_amdgcn_sbfe((int)m, e[i].y, 1);
    } else {
      int off, dd, dh, dw, rvalid;
      decode_row(e[i], off, dd, dh, dw, rvalid);
      bool k = cvalid & (rvalid != 0);
      if (chkD) k = k & ((unsigned)(id0 + dd) < (unsigned)p.SD);
      if (chkH) k = k & ((unsigned)(ih0 + dh) < (unsigned)p.SH);
      if (chkW) k = k & ((unsigned)(iw0 + dw) < (unsigned)p.SW);
      inv = k ? 0 : -1;
    }
    const unsigned voff = (cb4 + (unsigned)(e[i].x >> (H ? 1 : 0))) | (unsigned)inv;  // table offsets are SIGNED fp32 bytes (flipped dgrad taps are negative)
    if (VEC) bvec[VEC ? i : 0] = ld4(voff);
    else breg[VEC ? 0 : i] = ld1(voff);
  };
  // piece g of the next tile's fetch, placed after MFMA group g (G groups per tile): table rows + A tile
  // first, then the gathers spread over the following groups, leaving the last group(s) as latency shadow
  constexpr int G = (MATH ? 1 : 2) * TM * TN;
  constexpr int GSPAN = G > 2 ? G - 2 : 1;                              // groups 1..GSPAN carry the gathers
  constexpr int GCHUNK = (B_PER + GSPAN - 1) / GSPAN;
  auto issue_piece = [&](int g, int kt) __attribute__((always_inline)) {
    if (G == 1) {
      issue_table(kt); issue_a(kt);
#pragma unroll
      for (int i = 0; i < B_PER; ++i) issue_gather(i);
    } else if (g == 0) { issue_table(kt); issue_a(kt); }
    else if (g <= GSPAN) {
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        if (i >= (g - 1) * GCHUNK && i < g * GCHUNK) issue_gather(i);
    }
  };
  auto load_tiles = [&](int kt) __attribute__((always_inline)) {        // everything at once (prologue)
    issue_table(kt);
    issue_a(kt);
#pragma unroll
    for (int i = 0; i < B_PER; ++i) issue_gather(i);
  };
  // MATH >= 1: every fp32 operand x is stored as bf16 parts -- bf16x3: hi = bf16(x), lo = bf16(x - hi); bf16x6: hi, mid, lo
  // (x = hi + mid + lo to 2^-27).  A row of 16 k is two 8-k groups [hi x8 | (mid x8 |) lo x8]: the 16-byte operands that a
  // lane of half h feeds to v_mfma_f32_32x32x16_bf16 sit at float index 4*NP*h + {0, 4, (8)} of the row.
  auto store_split4 = [&](float* row, int chunk, float4 v, bool raw = false) __attribute__((always_inline)) {     // k = 4*chunk .. +3 of `row`
    float* d = row + (chunk >> 1) * (4 * NP) + (chunk & 1) * 2;
    if constexpr (MATH == 3) {      // fp16 operands: weights (fp32, raw = false) are rounded here; gathered halves (raw) are packed
      uint2 o;
      if (raw) {
        o.x = __float_as_uint(v.x) | (__float_as_uint(v.y) << 16);
        o.y = __float_as_uint(v.z) | (__float_as_uint(v.w) << 16);
      } else {
        const f16x2 lo = {(_Float16)v.x, (_Float16)v.y}, hi = {(_Float16)v.z, (_Float16)v.w};
        o.x = __builtin_bit_cast(unsigned, lo);
        o.y = __builtin_bit_cast(unsigned, hi);
      }
      *reinterpret_cast<uint2*>(d) = o;
    } else if (MATH == 2) {
      uint2 h, m, l;
      split_bf16x3(v.x, v.y, h.x, m.x, l.x);
      split_bf16x3(v.z, v.w, h.y, m.y, l.y);
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + 4) = m;
      *reinterpret_cast<uint2*>(d + 8) = l;
    } else {
      uint2 h, l;
      split_bf16x2(v.x, v.y, h.x, l.x);
      split_bf16x2(v.z, v.w, h.y, l.y);
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + 4) = l;
    }
  };
  auto store_tiles = [&](int buf) __attribute__((always_inline)) {
    if constexpr (MATH >= 1) {
      if (A_F4 > 1 || a_last_ok) store_split4(&As[buf][a_slot(0) >> 2][0], a_slot(0) & 3, a0);
      if (A_F4 == 2 ? a_last_ok : A_F4 > 2) store_split4(&As[buf][a_slot(1) >> 2][0], a_slot(1) & 3, a1);
      if (A_F4 > 2 && a_last_ok) store_split4(&As[buf][a_slot(2) >> 2][0], a_slot(2) & 3, a2);
      if (VEC) {
        store_split4(&Bs[buf][col + 0][0], wn, make_float4(bvec[0].x, bvec[1].x, bvec[2].x, bvec[3].x), true);
        store_split4(&Bs[buf][col + 1][0], wn, make_float4(bvec[0].y, bvec[1].y, bvec[2].y, bvec[3].y), true);
        store_split4(&Bs[buf][col + 2][0], wn, make_float4(bvec[0].z, bvec[1].z, bvec[2].z, bvec[3].z), true);
        store_split4(&Bs[buf][col + 3][0], wn, make_float4(bvec[0].w, bvec[1].w, bvec[2].w, bvec[3].w), true);
      } else {          // 8 consecutive k of one column = one whole group: b128 per part
        float* d = &Bs[buf][col][(r0 >> 3) * (4 * NP)];
        constexpr int I1 = VEC ? 0 : 1, I2 = VEC ? 0 : 2, I3 = VEC ? 0 : 3, I4 = VEC ? 0 : 4, I5 = VEC ? 0 : 5,
                      I6 = VEC ? 0 : 6, I7 = VEC ? 0 : 7;
        if constexpr (MATH == 3) {
          *reinterpret_cast<uint4*>(d) = make_uint4(__float_as_uint(breg[0]) | (__float_as_uint(breg[I1]) << 16),
                                                    __float_as_uint(breg[I2]) | (__float_as_uint(breg[I3]) << 16),
                                                    __float_as_uint(breg[I4]) | (__float_as_uint(breg[I5]) << 16),
                                                    __float_as_uint(breg[I6]) | (__float_as_uint(breg[I7]) << 16));
        } else if (MATH == 2) {
          uint4 h, m, l;
          split_bf16x3(breg[0], breg[I1], h.x, m.x, l.x);
          split_bf16x3(breg[I2], breg[I3], h.y, m.y, l.y);
          split_bf16x3(breg[I4], breg[I5], h.z, m.z, l.z);
          split_bf16x3(breg[I6], breg[I7], h.w, m.w, l.w);
          *reinterpret_cast<uint4*>(d) = h;
          *reinterpret_cast<uint4*>(d + 4) = m;
          *reinterpret_cast<uint4*>(d + 8) = l;
        } else {
          uint4 h, l;
          split_bf16x2(breg[0], breg[I1], h.x, l.x);
          split_bf16x2(breg[I2], breg[I3], h.y, l.y);
          split_bf16x2(breg[I4], breg[I5], h.z, l.z);
          split_bf16x2(breg[I6], breg[I7], h.w, l.w);
          *reinterpret_cast<uint4*>(d) = h;
          *reinterpret_cast<uint4*>(d + 4) = l;
        }
      }
      return;
    }
    if (A_F4 > 1 || a_last_ok) *reinterpret_cast<float4*>(&As[buf][a_slot(0) >> 2][(a_slot(0) & 3) * 4]) = a0;
    if (A_F4 == 2 ? a_last_ok : A_F4 > 2) *reinterpret_cast<float4*>(&As[buf][a_slot(1) >> 2][(a_slot(1) & 3) * 4]) = a1;
    if (A_F4 > 2 && a_last_ok) *reinterpret_cast<float4*>(&As[buf][a_slot(2) >> 2][(a_slot(2) & 3) * 4]) = a2;
    if (VEC) {      // 4 k-rows x 4 columns per thread: transpose in registers, one b128 per column
      *reinterpret_cast<float4*>(&Bs[buf][col + 0][r0]) = make_float4(bvec[0].x, bvec[1].x, bvec[2].x, bvec[3].x);
      *reinterpret_cast<float4*>(&Bs[buf][col + 1][r0]) = make_float4(bvec[0].y, bvec[1].y, bvec[2].y, bvec[3].y);
      *reinterpret_cast<float4*>(&Bs[buf][col + 2][r0]) = make_float4(bvec[0].z, bvec[1].z, bvec[2].z, bvec[3].z);
      *reinterpret_cast<float4*>(&Bs[buf][col + 3][r0]) = make_float4(bvec[0].w, bvec[1].w, bvec[2].w, bvec[3].w);
    } else {        // 8 consecutive k-rows of one column
      *reinterpret_cast<float4*>(&Bs[buf][col][r0]) = make_float4(breg[0], breg[1], breg[2], breg[3]);
      *reinterpret_cast<float4*>(&Bs[buf][col][r0 + 4]) = make_float4(breg[4], breg[5], breg[6], breg[7]);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.Kpad / BK;
  const int kt0 = split * p.kt_per_split;
  int kt1 = kt0 + p.kt_per_split; if (kt1 > nk) kt1 = nk;
  const int lh = lane >> 5, ll = lane & 31;
  if (kt0 < kt1) {
    load_tiles(kt0);
    store_tiles(0);
  }
  __syncthreads();
  // One k-tile: 2 x TM x TN groups of four dependent MFMAs with the pieces of the next tile's fetch in between.
  // BUF (LDS buffer) and MORE (is there a next tile) are compile-time, so the steady-state loop below carries no
  // buffer arithmetic and no per-group branches; the last tile(s) are peeled.
  auto tile = [&](auto BUF, auto MORE, int kt_next) __attribute__((always_inline)) {
    constexpr int buf = decltype(BUF)::value;
    constexpr bool more = decltype(MORE)::value;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][i * 32 + ll][8 * t + 4 * lh]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[buf][wn * (TN * 32) + j * 32 + ll][8 * t + 4 * lh]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
          if (more) {
            __builtin_amdgcn_sched_barrier(0);
            issue_piece((t * TM + i) * TN + j, kt_next);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  if constexpr (MATH >= 1) {
    // bf16x3: x*y ~= xh*yh + xh*yl + xl*yh (the dropped xl*yl term and the rounding of the lo parts are ~2^-17 relative).
    // bf16x6: x*y ~= hh + hm + mh + hl + lh + mm (dropped terms and part rounding <= 2^-25).
    // fp32 accumulation in both.  One k-tile = TM*TN groups of 3 / 6 32-cycle MFMAs -- 5.3x / 2.7x less matrix-pipe time
    // than the fp32 MFMA path's 8 x 64 cycles, i.e. SHORTER than a trip to L2/HBM.  The fetch therefore runs TWO tiles
    // ahead: while tile kt is multiplied out of LDS, tile kt+1 is in flight in one register set and the loads of tile
    // kt+2 are issued into the other (the loop is unrolled by two so that both sets are plain registers).
    if constexpr (!VEC && TM <= 3) {
      struct Fetch { float4 a0, a1, a2; float b[VEC ? 1 : B_PER]; float4 bv[VEC ? B_PER : 1]; int2 e[B_PER]; };
      Fetch F0, F1;
      auto f_table = [&](Fetch& F, int kt) __attribute__((always_inline)) {
        const int2* trow = table + kt * BK + __builtin_amdgcn_readfirstlane(r0);
  #pragma unroll
        for (int i = 0; i < B_PER; ++i) F.e[i] = trow[i];
      };
      auto f_a = [&](Fetch& F, int kt) __attribute__((always_inline)) {
        F.a0 = *reinterpret_cast<const float4*>(arow + ao0 + kt * BK);
        if (A_F4 > 1) F.a1 = *reinterpret_cast<const float4*>(arow + ao1 + kt * BK);
        if (A_F4 > 2) F.a2 = *reinterpret_cast<const float4*>(arow + ao2 + kt * BK);
      };
      auto f_gather = [&](Fetch& F, int i) __attribute__((always_inline)) {
        int inv;
        if (FAST == 1) {
          inv = __builtin_amdgcn_sbfe((int)ilo, F.e[i].y, 1);
        } else if (FAST == 2) {
          const unsigned hsel = (unsigned)-((F.e[i].y >> 5) & 1);
          inv = __builtin_amdgcn_sbfe((int)(ilo ^ ((ilo ^ ihi) & hsel)), F.e[i].y, 1);
        } else {
          int off, dd, dh, dw, rvalid;
          decode_row(F.e[i], off, dd, dh, dw, rvalid);
          bool k = cvalid & (rvalid != 0);
          if (chkD) k = k & ((unsigned)(id0 + dd) < (unsigned)p.SD);
          if (chkH) k = k & ((unsigned)(ih0 + dh) < (unsigned)p.SH);
          if (chkW) k = k & ((unsigned)(iw0 + dw) < (unsigned)p.SW);
          inv = k ? 0 : -1;
        }
        const unsigned voff = (cb4 + (unsigned)(F.e[i].x >> (H ? 1 : 0))) | (unsigned)inv;
        if (VEC) F.bv[VEC ? i : 0] = ld4(voff);
        else F.b[VEC ? 0 : i] = ld1(voff);
      };
      auto f_all = [&](Fetch& F, int kt) __attribute__((always_inline)) {
        f_table(F, kt); f_a(F, kt);
  #pragma unroll
        for (int i = 0; i < B_PER; ++i) f_gather(F, i);
      };
      auto f_piece = [&](Fetch& F, int g, int kt) __attribute__((always_inline)) {
        if (G == 1) f_all(F, kt);
        else if (g == 0) { f_table(F, kt); f_a(F, kt); }
        else if (g <= GSPAN) {
  #pragma unroll
          for (int i = 0; i < B_PER; ++i)
            if (i >= (g - 1) * GCHUNK && i < g * GCHUNK) f_gather(F, i);
        }
      };
      auto stash = [&](Fetch& F, int buf) __attribute__((always_inline)) {      // convert + write to LDS (store_tiles reads a0.. / breg / bvec)
        a0 = F.a0; a1 = F.a1; a2 = F.a2;
  #pragma unroll
        for (int i = 0; i < (VEC ? 1 : B_PER); ++i) breg[i] = F.b[i];
  #pragma unroll
        for (int i = 0; i < (VEC ? B_PER : 1); ++i) bvec[i] = F.bv[i];
        store_tiles(buf);
      };
      // tile kt is in LDS[buf]; Fin holds tile kt+1 (in flight); tile kt+2 is fetched into Fnew.  No branch depends on
      // "is there a next tile": past the end the last tile is fetched / stored again (clamped index, nobody reads it), so
      // that the compiler can wait with exact vmcnt counts (a conditional fetch forces s_waitcnt vmcnt(0) at every use).
      auto body = [&](auto BUF, Fetch& Fin, Fetch& Fnew, int kt) __attribute__((always_inline)) {
        constexpr int NB = lds_bufs(TM, VEC, MATH);
        constexpr int buf = NB == 2 ? decltype(BUF)::value : 0;
        const int ktn = min(kt + 2, kt1 - 1);
        float4 bf[TN][NP], afc[NP], afn[NP];
  #pragma unroll
        for (int j = 0; j < TN; ++j)
  #pragma unroll
          for (int q = 0; q < NP; ++q) bf[j][q] = *reinterpret_cast<const float4*>(&Bs[buf][wn * (TN * 32) + j * 32 + ll][4 * NP * lh + 4 * q]);
  #pragma unroll
        for (int q = 0; q < NP; ++q) afc[q] = *reinterpret_cast<const float4*>(&As[buf][ll][4 * NP * lh + 4 * q]);
  #pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (i + 1 < TM) {         // next row tile's fragments, one group ahead of their MFMAs
  #pragma unroll
            for (int q = 0; q < NP; ++q) afn[q] = *reinterpret_cast<const float4*>(&As[buf][(i + 1) * 32 + ll][4 * NP * lh + 4 * q]);
          }
  #pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (MATH == 3) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afc[0]), __builtin_bit_cast(f16x8, bf[j][0]), acc[i][j], 0, 0, 0);
            } else {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, afc[0]), xl = __builtin_bit_cast(bf16x8, afc[NP - 1]);
            const bf16x8 yh = __builtin_bit_cast(bf16x8, bf[j][0]), yl = __builtin_bit_cast(bf16x8, bf[j][NP - 1]);
            if (MATH == 2) {
              const bf16x8 xm = __builtin_bit_cast(bf16x8, afc[1]), ym = __builtin_bit_cast(bf16x8, bf[j][1]);
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
            __builtin_amdgcn_sched_barrier(0);
            f_piece(Fnew, i * TN + j, ktn);
            __builtin_amdgcn_sched_barrier(0);
          }
  #pragma unroll
          for (int q = 0; q < NP; ++q) afc[q] = afn[q];
        }
        if (NB == 1) __syncthreads();                 // every wave is done reading the only buffer
        stash(Fin, NB == 2 ? buf ^ 1 : 0);
        __syncthreads();
      };
      if (kt0 < kt1) {
        f_all(F1, min(kt0 + 1, kt1 - 1));
        for (int kt = kt0; kt < kt1; kt += 2) {
          body(B0{}, F1, F0, kt);
          if (kt + 1 >= kt1) break;
          body(B1{}, F0, F1, kt + 1);
        }
      }
    } else {
      // tall / 256-column tiles: two register sets cost them a resident workgroup (measured slower), and their MFMA
      // phase is long enough for a one-tile-ahead fetch
      for (int kt = kt0; kt < kt1; ++kt) {
        constexpr int NB = lds_bufs(TM, VEC, MATH);
        const int buf = NB == 2 ? (kt - kt0) & 1 : 0;
        const bool more = kt + 1 < kt1;
        float4 af[TM][NP], bf[TN][NP];
  #pragma unroll
        for (int i = 0; i < TM; ++i)
  #pragma unroll
          for (int q = 0; q < NP; ++q) af[i][q] = *reinterpret_cast<const float4*>(&As[buf][i * 32 + ll][4 * NP * lh + 4 * q]);
  #pragma unroll
        for (int j = 0; j < TN; ++j)
  #pragma unroll
          for (int q = 0; q < NP; ++q) bf[j][q] = *reinterpret_cast<const float4*>(&Bs[buf][wn * (TN * 32) + j * 32 + ll][4 * NP * lh + 4 * q]);
  #pragma unroll
        for (int i = 0; i < TM; ++i)
  #pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (MATH == 3) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, bf[j][0]), acc[i][j], 0, 0, 0);
            } else {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, af[i][0]), xl = __builtin_bit_cast(bf16x8, af[i][NP - 1]);
            const bf16x8 yh = __builtin_bit_cast(bf16x8, bf[j][0]), yl = __builtin_bit_cast(bf16x8, bf[j][NP - 1]);
            if (MATH == 2) {
              const bf16x8 xm = __builtin_bit_cast(bf16x8, af[i][1]), ym = __builtin_bit_cast(bf16x8, bf[j][1]);
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
            __builtin_amdgcn_sched_barrier(0);
            if (more) issue_piece(i * TN + j, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        if (NB == 1) __syncthreads();               // every wave is done reading the only buffer
        if (more) store_tiles(NB == 2 ? buf ^ 1 : 0);
        __syncthreads();
      }
    }
  } else if (TM <= 2 && !VEC) {
    // small tiles: the loop is unrolled by two (buffers 0 / 1) and the tail peeled -- the per-tile scalar
    // bookkeeping is a visible share of an iteration that holds only 8-16 MFMAs (measured +2..+8 % here, but a
    // loss for the tall / 256-column tiles, whose bodies are already long: they keep the rolled loop below)
    int kt = kt0;
    for (; kt + 2 < kt1; kt += 2) {
      tile(B0{}, std::true_type{}, kt + 1);
      tile(B1{}, std::true_type{}, kt + 2);
    }
    if (kt + 1 < kt1) {
      tile(B0{}, std::true_type{}, kt + 1);
      tile(B1{}, std::false_type{}, 0);
    } else if (kt < kt1) {
      tile(B0{}, std::false_type{}, 0);
    }
  } else {
    for (int kt = kt0; kt < kt1; ++kt) {
      constexpr int NB = lds_bufs(TM, VEC, MATH);
      const int buf = NB == 2 ? (kt - kt0) & 1 : 0;
      const bool more = kt + 1 < kt1;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][i * 32 + ll][8 * t + 4 * lh]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[buf][wn * (TN * 32) + j * 32 + ll][8 * t + 4 * lh]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) issue_piece((t * TM + i) * TN + j, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      if (NB == 1) __syncthreads();                 // every wave is done reading the only buffer
      if (more) store_tiles(NB == 2 ? buf ^ 1 : 0);
      __syncthreads();
    }
  }

  // ---- epilogue: C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Stores go through a buffer resource: the lane keeps ONE 32-bit byte offset per column tile (its column and
  // its row half), the row part of the address is wave-uniform (soffset), and an invalid row or column is an
  // all-ones offset that the hardware drops -- the epilogue needs a handful of VGPRs instead of 16 pointers.
  const int mbase = tileM * BM;
  const int rows_left = p.DK - mbase - 4 * lh;              // row (i,r) of this lane is real iff i*32 + ro(r) < rows_left
  if (p.splits > 1) {
    // partial tile -> slab[split][m][n]; bias / accumulate / BN sums happen in conv_splitk_finish_kernel
    float* sl = slab + (long long)split * p.DK * p.Ntot;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(sl, 0, p.slab_bytes, 0x00020000);
    const unsigned rowb = (unsigned)p.Ntot * 4u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long long n = (long long)tileN * BN + wn * (TN * 32) + j * 32 + ll;
      const unsigned vb = n < p.Ntot ? ((unsigned)n + (unsigned)(mbase + 4 * lh) * (unsigned)p.Ntot) * 4u : 0xffffffffu;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
          const unsigned vo = ro < rows_left ? vb : 0xffffffffu;
          const float v = acc[i][j][r];     // NB: never bit_cast a vector ELEMENT expression (hipcc 7.2 takes element 0)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rd, (int)vo, (int)((unsigned)ro * rowb), 0);
        }
    }
    return;
  }
  const int DHW = p.DH * p.DW;
  const unsigned DSP = (unsigned)(p.DD * DHW);
  const unsigned rowb = DSP * ES;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, p.dst_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long long n = (long long)tileN * BN + wn * (TN * 32) + j * 32 + ll;
    const bool nv = n < p.Ntot;
    const long long nc = nv ? n : 0;
    const int img = (int)(nc / QSP);
    const int sp = (int)(nc - (long long)img * QSP);
    const int qd = sp / QHW, rr = sp - qd * QHW;
    const int qh = rr / p.QW, qw = rr - qh * p.QW;
    const unsigned dsp = (unsigned)((qd * p.dm_d + p.do_d) * DHW + (qh * p.dm_h + p.do_h) * p.DW + (qw * p.dm_w + p.do_w));
    const unsigned vb = nv ? (((unsigned)img * (unsigned)p.DK + (unsigned)(mbase + 4 * lh)) * DSP + dsp) * ES : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float old[16];
      if (p.accumulate) {                       // all 16 read-modify-write loads in flight together
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
          const unsigned vo = ro < rows_left ? vb : 0xffffffffu;
          if constexpr (H) old[r] = (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rd, (int)vo, (int)((unsigned)ro * rowb), 0));
          else old[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, (int)vo, (int)((unsigned)ro * rowb), 0));
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
        const unsigned vo = ro < rows_left ? vb : 0xffffffffu;
        float v = acc[i][j][r];
        if (bias) v += bias[min(mbase + ro + 4 * lh, p.DK - 1)];
        if (p.accumulate) v += old[r];
        if constexpr (H) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), rd, (int)vo, (int)((unsigned)ro * rowb), 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rd, (int)vo, (int)((unsigned)ro * rowb), 0);
      }
    }
  }
  if (psum) {
    // per-channel partial sum / sum of squares of this TILE (invalid columns hold 0): DPP adds over the 32 lanes of
    // each wave half (totals land in lanes 16..31 / 48..63), then the four waves' totals are combined through LDS
    // (the operand tiles are dead by now) so that BatchNorm's finalize reads one partial per tile, not four.
    float* red = &Bs[0][0][0];                                // [WN][BM][2] floats  (<= 4*160*2 < one B buffer)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) { const float v = acc[i][j][r]; s += v; q += v * v; }
        s = half_wave_sum_hi(s);
        q = half_wave_sum_hi(q);
        const int ro = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ll == 31) { red[(wn * BM + ro) * 2] = s; red[(wn * BM + ro) * 2 + 1] = q; }
      }
    }
    __syncthreads();
    if (tid < BM && mbase + tid < p.DK) {
      const float s = red[tid * 2] + red[(BM + tid) * 2] + red[(2 * BM + tid) * 2] + red[(3 * BM + tid) * 2];
      const float q = red[tid * 2 + 1] + red[(BM + tid) * 2 + 1] + red[(2 * BM + tid) * 2 + 1] + red[(3 * BM + tid) * 2 + 1];
      const long long m = mbase + tid;
      psum[m * p.P + tileN] = s;
      psq[m * p.P + tileN] = q;
    }
  }
}

template <int TM, int BN, int FAST, bool VEC, int MATH, bool H = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const void* __restrict__ src, const float* __restrict__ apack, const int2* __restrict__ table,
    const float* __restrict__ bias, void* __restrict__ dst, float* __restrict__ psum,
    float* __restrict__ psq, float* __restrict__ slab, IgemmParams p) {
  __shared__ __attribute__((aligned(16))) float As_[lds_bufs(TM, VEC, MATH) * 32 * TM * lds_pitch(MATH)];
  __shared__ __attribute__((aligned(16))) float Bs_[lds_bufs(TM, VEC, MATH) * BN * lds_pitch(MATH)];
  __shared__ int taptab[64];
  igemm_tile<TM, BN, FAST, VEC, MATH, H>(src, apack, table, bias, dst, psum, psq, slab, p, gca_xcd_remap(blockIdx.x, gridDim.x), As_, Bs_,
                                   taptab);
}

// Two tile heights in ONE launch: workgroups [0, nA) run TMA-row tiles over the first p.tilesN column tiles, the rest run
// TMB-row tiles over the remaining ones.  Workgroups are dispatched in id order, so the short tiles flow into the CUs
// that the last, partly filled wave of tall workgroups leaves idle (a second LAUNCH would wait for that wave to end).
template <int TMA, int TMB, int FAST, int MATH>
__global__ __launch_bounds__(256) void conv_igemm_2phase_kernel(
    const void* __restrict__ src, const float* __restrict__ apack, const int2* __restrict__ table,
    const float* __restrict__ bias, void* __restrict__ dst, float* __restrict__ psum,
    float* __restrict__ psq, IgemmParams p, int nA, int tilesM_B, int tilesN_B) {
  static_assert(TMB < TMA, "the tail uses the shorter tile");
  __shared__ __attribute__((aligned(16))) float As_[cmax(lds_bufs(TMA, false, MATH) * TMA, lds_bufs(TMB, false, MATH) * TMB) * 32 * lds_pitch(MATH)];
  __shared__ __attribute__((aligned(16))) float Bs_[cmax(lds_bufs(TMA, false, MATH), lds_bufs(TMB, false, MATH)) * 128 * lds_pitch(MATH)];
  __shared__ int taptab[64];
  if ((int)blockIdx.x < nA) {
    igemm_tile<TMA, 128, FAST, false, MATH>(src, apack, table, bias, dst, psum, psq, nullptr, p, gca_xcd_remap(blockIdx.x, nA), As_, Bs_, taptab);
  } else {
    IgemmParams pb = p;
    pb.tileN_off = p.tileN_off + p.tilesN;
    pb.tilesM = tilesM_B; pb.tilesN = tilesN_B;
    igemm_tile<TMB, 128, FAST, false, MATH>(src, apack, table, bias, dst, psum, psq, nullptr, pb,
                                      gca_xcd_remap((int)blockIdx.x - nA, (int)gridDim.x - nA), As_, Bs_, taptab);
  }
}

// Split-K finishing pass, grid (channels, parts): sum the slabs in fixed order, add bias, (+=) store in
// NCDHW (through the class's destination map), and emit the BN partial sums [K][parts].
// gca_conv_fwd_slabs: the split-K classes leave their slabs un-finished (the BatchNorm that follows folds them)
thread_local bool t_leave_slabs = false;
thread_local int t_left_splits = 0;

constexpr int FINISH_CHUNK = 1024;      // columns per finishing block: split-K grids are small, so many short blocks (4 columns per thread;
                                        // issuing the slab loads of all four columns together was measured: 9.1 vs 8.6 us per launch, not kept)
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(
    const float* __restrict__ slab, int splits, const float* __restrict__ bias, T* __restrict__ dst,
    float* __restrict__ psum, float* __restrict__ psq, IgemmParams p) {
  __shared__ double sh[4];
  const int m = blockIdx.x, part = blockIdx.y, P = gridDim.y;
  const float b = bias ? bias[m] : 0.f;
  const long long lo = (long long)part * FINISH_CHUNK;
  long long hi = lo + FINISH_CHUNK; if (hi > p.Ntot) hi = p.Ntot;
  const int QSP = p.QD * p.QH * p.QW, QHW = p.QH * p.QW, DHW = p.DH * p.DW;
  const long long DSP = (long long)p.DD * DHW;
  double s = 0.0, q = 0.0;
  for (long long n = lo + threadIdx.x; n < hi; n += 256) {
    float v = 0.f;
    const float* sp0 = slab + (long long)m * p.Ntot + n;
    const long long sstride = (long long)p.DK * p.Ntot;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {               // four loads in flight, summed in the fixed order 0,1,2,...
      const float l0 = sp0[(long long)k * sstride], l1 = sp0[(long long)(k + 1) * sstride];
      const float l2 = sp0[(long long)(k + 2) * sstride], l3 = sp0[(long long)(k + 3) * sstride];
      v += l0; v += l1; v += l2; v += l3;
    }
    for (; k < splits; ++k) v += sp0[(long long)k * sstride];
    s += (double)v; q += (double)v * (double)v;       // statistics of the conv output proper (no bias on that path)
    v += b;
    const int img = (int)(n / QSP);
    const int sp = (int)(n - (long long)img * QSP);
    const int qd = sp / QHW, rr = sp - qd * QHW;
    const int qh = rr / p.QW, qw = rr - qh * p.QW;
    const long long dsp = (long long)(qd * p.dm_d + p.do_d) * DHW + (qh * p.dm_h + p.do_h) * p.DW + (qw * p.dm_w + p.do_w);
    T* d = dst + ((long long)img * p.DK + m) * DSP + dsp;
    if (p.accumulate) v += (float)*d;
    *d = (T)v;
  }
  if (psum) {
    s = gca_block_sum256_d(s, sh);
    q = gca_block_sum256_d(q, sh);
    if (threadIdx.x == 0) { psum[(long long)m * P + part] = (float)s; psq[(long long)m * P + part] = (float)q; }
  }
}

// packed[m][k] (k contiguous, zero padded to [Mrows][Kpad]), k = (ch, t') over the class's tap grid:
// tap = ((kd0+sd*a)*KH + (kh0+sh*b))*KW + (kw0+sw*c).  fwd (which 0): value = W[m][ch][tap];
// dgrad (which 1): value = W[ch][m][tap].  One thread per packed element, k fastest (coalesced writes; the
// reads are contiguous for the forward layout and tap-strided for dgrad -- weights are L2-resident).
__device__ __forceinline__ void pack_elements(const float* __restrict__ w, float* __restrict__ packed,
                                              const PackParams& p, long long first, long long step, long long end) {
  for (long long i = first; i < end; i += step) {
    const int m = (int)(i / p.Kpad), k = (int)(i - (long long)m * p.Kpad);
    float v = 0.f;
    if (m < p.M && k < p.Kred) {
      const int ch = k / p.ntaps, tl = k - ch * p.ntaps;
      const int a = tl / (p.nb * p.nc), r = tl - a * (p.nb * p.nc), b = r / p.nc, c = r - b * p.nc;
      const int tap = ((p.k0d + p.sd * a) * p.KH + (p.k0h + p.sh * b)) * p.KW + (p.k0w + p.sw * c);
      v = w[(long long)ch * p.s_ch + tap + (long long)m * p.s_m];
    }
    packed[i] = v;
  }
}
__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, PackParams p) {
  if (p.fmt >= 1)
    pack_halo_elements(w, reinterpret_cast<unsigned char*>(packed), p, (long long)blockIdx.x * blockDim.x + threadIdx.x,
                       (long long)gridDim.x * blockDim.x, pack_items(p));
  else
    pack_elements(w, packed, p, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x,
                  pack_items(p));
}

// All convolutions of an encoder in ONE launch.  A job is one problem class of one layer; jobs own consecutive
// block ranges (PACK_CHUNK elements per block) so big and small layers are balanced; a block finds its job by
// binary search over first_block.
constexpr int PACK_CHUNK = 2048;
struct PackJob {
  PackParams p;
  const float* w;
  float* packed;
  int first_block, nblocks;
};
static_assert(sizeof(PackJob) <= GCA_PACK_JOB_BYTES, "gca_hip.h: GCA_PACK_JOB_BYTES too small");
__global__ __launch_bounds__(256) void conv_pack_batched_kernel(const unsigned char* __restrict__ jobs, int njobs) {
  const int bid = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (reinterpret_cast<const PackJob*>(jobs + (size_t)mid * GCA_PACK_JOB_BYTES)->first_block <= bid) lo = mid;
    else hi = mid - 1;
  }
  const PackJob* j = reinterpret_cast<const PackJob*>(jobs + (size_t)lo * GCA_PACK_JOB_BYTES);
  const PackParams p = j->p;
  const long long base = (long long)(bid - j->first_block) * PACK_CHUNK;
  long long end = base + PACK_CHUNK;
  const long long total = pack_items(p);
  if (end > total) end = total;
  if (p.fmt >= 1) pack_halo_elements(j->w, reinterpret_cast<unsigned char*>(j->packed), p, base + threadIdx.x, 256, end);
  else pack_elements(j->w, j->packed, p, base + threadIdx.x, 256, end);
}

// n32 whole 32-bit words, then (fp16 tensors with an odd element count) one trailing 16-bit element
__global__ void zero_fill_kernel(unsigned* __restrict__ p, long long n32, int tail16) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (long long)gridDim.x * blockDim.x) p[i] = 0u;
  if (tail16 && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned short*>(p + n32)[0] = 0;
}

// ---- launch configuration ------------------------------------------------------------------
struct IgemmCfg {
  int bm; int bn; int splits; int kt_per_split; int tail_bm; int main_cols; int math;
  int halo;              // 1: conv3d_halo.hip (LDS halo tiles); kt_per_split then counts 16-channel chunks
  int bd, bh, bw;        // halo: box of a tile
  int h;                 // fp16 storage (gca_set_conv_math(3)): the gather kernels widen on load / round on store and
                         // multiply in bf16x3 (exact for fp16 operands); the halo kernels multiply on the f16 MFMA (math 3)
};


// Heuristic default (the host side may override it per geometry after measuring: gca_conv_geom.tune_*).
// Tile = 32*TM rows (TM = 1..5, picked to minimise padded rows) x 128 columns, or x 256 columns with float4
// gathers for pointwise-in-space classes; the K loop is split when the tile grid cannot occupy the CUs and
// the partial slabs stay small.  tune code: rows | 1024 for the 256-column variant.
inline IgemmCfg choose_cfg(int M, long long Ntot, int nk, bool vec_ok, int force_bm, int force_splits) {
  IgemmCfg best{64, 128, 1, nk, 0, 0, 0, 0, 0, 0, 0, 0};
  double best_cost = 1e300;
  const bool force_vec = force_bm >= 1024;
  const int force_rows = force_bm & 1023;
  for (int v = 0; v < 2; ++v) {
    if (v == 1 && !vec_ok) continue;
    if (force_bm && (v == 1) != (force_vec && vec_ok)) continue;
    for (int tm = 1; tm <= 5; ++tm) {
      const int bm = 32 * tm, bn = v ? 256 : 128;
      if (force_rows && bm != force_rows) continue;
      const long long tiles = gca_ceil_div(M, bm) * gca_ceil_div(Ntot, bn);
      int s = 1;
      if (force_splits > 0) s = force_splits;
      else if (tiles < 2 * NUM_CU && nk >= 8 && (long long)M * Ntot <= (1LL << 20)) {
        long long want = gca_ceil_div(2 * NUM_CU, tiles);
        if (want > nk / 4) want = nk / 4;
        if (want > 16) want = 16;
        if (want > 1) s = (int)want;
      }
      if (s > nk) s = nk;
      if (s < 1) s = 1;
      const int per = (int)gca_ceil_div(nk, s);
      s = (int)gca_ceil_div(nk, per);
      const double wg_per_cu = (double)(tiles * s) / NUM_CU;
      const double occ = wg_per_cu < 1.0 ? 0.55 : (wg_per_cu < 2.0 ? 0.75 : 1.0);   // latency hiding needs >= 2 WGs / CU
      const double rounds = (double)gca_ceil_div(tiles * s, NUM_CU);
      // taller tiles amortise the gather over more MFMAs; the float4 variant gathers 4x cheaper
      const double eff = (0.55 + 0.1 * tm) * (v ? 1.15 : 1.0) * occ;
      const double work = (double)bm * bn * per;
      const double cost = rounds * work / eff + (s > 1 ? 0.05 * rounds * work + 8.0 * bm * bn : 0.0);
      if (cost < best_cost) { best_cost = cost; best = IgemmCfg{bm, bn, s, per, 0, 0, 0, 0, 0, 0, 0, 0}; }
    }
  }
  return best;
}

inline void tune_of(const gca_conv_geom* g, int which, int& fbm, int& fs, int& tail, int& box) {
  if (which == 0) { fbm = g->tune_fwd_bm; fs = g->tune_fwd_splits; tail = g->tune_fwd_tail; box = g->tune_fwd_box; }
  else { fbm = g->tune_dgrad_bm; fs = g->tune_dgrad_splits; tail = g->tune_dgrad_tail; box = g->tune_dgrad_box; }
}

// ---- LDS-halo kernels (conv3d_halo.hip): geometry of a box tiling, eligibility, heuristic choice -------------------------
// GCA_HALO=0 keeps the un-tuned heuristic on the gather kernels (A/B runs; forced tune codes are still honoured)
inline bool halo_heuristic_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("GCA_HALO"); on = (e && e[0] == '0') ? 0 : 1; }
  return on != 0;
}
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ---- pointwise fp16 GEMM kernel (conv3d_pw.hip): 1x1x1, unit stride, no padding, fp16 storage, positions % 8 == 0 ---------
// GCA_PW=0 keeps un-tuned pointwise convs on the gather kernels (the forced tune code 8192 is still honoured)
inline bool pw_heuristic_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("GCA_PW"); on = (e && e[0] == '0') ? 0 : 1; }
  return on != 0;
}
inline bool pw_geometry(const gca_conv_geom* g, int which, const IgemmParams& ip, PwParams& pw) {
  if (!g->act_f16 || g->kd != 1 || g->kh != 1 || g->kw != 1 || g->sd != 1 || g->sh != 1 || g->sw != 1 || g->pd || g->ph || g->pw)
    return false;
  const long long SP = (long long)g->D * g->H * g->W;
  if (SP % 8 != 0 || SP < 128 || g->x_batch_stride % 8 != 0) return false;
  const int Kc = which == 0 ? g->C : g->K, DK = which == 0 ? g->K : g->C;
  pw.DK = DK; pw.Kc = Kc; pw.Kpad = (int)gca_round_up(Kc, 32);
  pw.SP = (int)SP; pw.N = g->N;
  pw.tilesM = cdiv(DK, 128); pw.tiles_sp = cdiv((int)SP, 128);
  pw.P = g->N * pw.tiles_sp;
  pw.accumulate = ip.accumulate;
  pw.src_nstride = (unsigned)ip.src_nstride;
  pw.src_bytes = ip.src_bytes; pw.dst_bytes = ip.dst_bytes;
  const long long pb = (long long)pack_rows(DK) * pw.Kpad * 2;
  pw.pack_bytes = pb > 0xfffff000LL ? 0xfffff000u : (unsigned)pb;
  return true;
}

// LDS layout of a stem halo row (conv3d_stem.hip): [phase][part][copy][cd dwords].  (Measured and not kept: cd = 16 (mod 32)
// with a row pitch of bw/2 (mod 32) dwords per output row, which makes every 32-lane read group conflict-free on paper --
// the 2.7x larger rows cost more in occupancy and box shape than the conflicts do: 0.64 vs 0.66 ms on the R(2+1)D-18 stem
// in bf16x6, 2.6 vs 1.6 ms on the 3D-ResNet stem in fp16.)
inline int stem_copy_dwords(int bw) { return (bw + 4) / 2; }
inline int stem_row_bytes(int bw, int sh, int np) { (void)sh; return 4 * np * stem_copy_dwords(bw) * 4; }
// ---- stem kernels (conv3d_stem.hip): forward class, <= 4 input channels, stride 2 and <= 8 taps along W ---------------------
// GCA_STEM=0 keeps un-tuned stems on the gather kernels (A/B runs; the forced tune code 4096 is still honoured)
inline bool stem_heuristic_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("GCA_STEM"); on = (e && e[0] == '0') ? 0 : 1; }
  return on != 0;
}
// Fills sp (tile = 64 channels x 256 positions, or 128 x 128 for more than 64 output channels; box = the tiling with the least staging work per output that fits two workgroups per CU, or one if
// none does); false when the conv cannot run there.
inline bool stem_geometry(const gca_conv_geom* g, const IgemmParams& p, int math, StemParams& sp) {
  if (math < 1 || math > 3 || g->C > 4 || g->sw != 2 || g->kw > 8 || g->x_batch_stride % 4 != 0) return false;
  if ((long long)g->N * g->OD * g->OH * g->OW < 256 * 32) return false;          // not worth a halo
  const int np = math == 3 ? 1 : (math == 2 ? 3 : 2);
  const int nrows = g->C * g->kd * g->kh, nsteps = cdiv(nrows, 2);
  // LDS tables: row offsets of the reduction rows (2 per step) + input-row origins of the halo (<= C*hd*hh, sized per box below)
  double best = 1e300;
  int bb[3] = {0, 0, 0};
  const int wmr = g->K > 64 ? 4 : 2, npos = 512 / wmr;         // wave rows; positions per tile (4 waves of 32 rows x 128 columns)
  for (int pass = 0; pass < 2 && bb[0] == 0; ++pass) {
    const size_t cap = pass == 0 ? (size_t)(76 << 10) : (size_t)(156 << 10);
    for (int bd = 1; bd <= npos; bd *= 2)
      for (int bh = 1; bd * bh <= npos; bh *= 2) {
        const int bw = npos / (bd * bh);
        if (bw < 4 || bw > 64) continue;
        if (bd > 2 * g->OD || bh > 2 * g->OH || bw > 2 * g->OW) continue;
        const int hd = (bd - 1) * g->sd + g->kd, hh = (bh - 1) * g->sh + g->kh, wp = bw + 4;
        const size_t lds = gca_round_up((2 * nsteps + g->C * hd * hh) * 4, 16) + (size_t)g->C * hd * hh * stem_row_bytes(bw, g->sh, np);
        if (lds > cap) continue;
        const double cover = (double)cdiv(g->OD, bd) * bd / g->OD * cdiv(g->OH, bh) * bh / g->OH * cdiv(g->OW, bw) * bw / g->OW;
        const double staging = (double)g->C * hd * hh * 2 * wp / (double)npos;    // staged elements per output position
        const double cost = cover * (1.0 + staging / (double)(nsteps * 8)) * (bw >= 8 ? 1.0 : 1.15);
        if (cost < best) { best = cost; bb[0] = bd; bb[1] = bh; bb[2] = bw; }
      }
  }
  if (bb[0] == 0) return false;
  sp.g = p;
  sp.wm = wmr;
  sp.bd = bb[0]; sp.bh = bb[1]; sp.bw = bb[2]; sp.lbh = ilog2(bb[1]); sp.lbw = ilog2(bb[2]);
  sp.nbd = cdiv(g->OD, sp.bd); sp.nbh = cdiv(g->OH, sp.bh); sp.nbw = cdiv(g->OW, sp.bw);
  sp.hd = (sp.bd - 1) * g->sd + g->kd; sp.hh = (sp.bh - 1) * g->sh + g->kh; sp.wp = sp.bw + 4;
  sp.cd = stem_copy_dwords(sp.bw); sp.hrowb = stem_row_bytes(sp.bw, g->sh, np);
  sp.sd = g->sd; sp.sh = g->sh; sp.pd = g->pd; sp.ph = g->ph; sp.pw = g->pw;
  sp.kd = g->kd; sp.kh = g->kh; sp.kw = g->kw;
  sp.nrows = nrows; sp.nsteps = nsteps;
  sp.Mrows = pack_rows(g->K);
  sp.rowoff_bytes = (int)gca_round_up((2 * nsteps + g->C * sp.hd * sp.hh) * 4, 16);
  sp.cs_bytes = (unsigned)((long long)g->D * g->H * g->W * (g->act_f16 ? 2 : 4));
  const long long pb = (long long)nsteps * sp.Mrows * (math == 3 ? 32 : (math == 2 ? 96 : 64));
  sp.pack_bytes = pb > 0xfffff000LL ? 0xfffff000u : (unsigned)pb;
  sp.m_w2 = gca_make_magic((unsigned)sp.wp); sp.m_hdh = gca_make_magic((unsigned)(sp.hd * sp.hh));
  sp.m_hh = gca_make_magic((unsigned)sp.hh);
  return true;
}

// halo geometry of class c tiled by boxes (bd, bh, bw); false when the class / box cannot run on the halo kernels
inline bool halo_geometry(const ClassInfo& c, const IgemmParams& p, int bd, int bh, int bw, int math, HaloParams& hp) {
  hp.in_scale = nullptr; hp.in_shift = nullptr;
  if (c.ntaps < 1 || c.ntaps > 64 || c.srcC < 8) return false;
  if (!is_pow2(bd) || !is_pow2(bh) || !is_pow2(bw)) return false;
  const int bn = bd * bh * bw;
  if (bn != 128 && bn != 256) return false;
  const int cn[3] = {c.na, c.nb, c.nc}, b[3] = {bd, bh, bw};
  int h[3], dmin[3];
  for (int d = 0; d < 3; ++d) {
    const int lo = c.dls[d] > 0 ? c.dl0[d] : c.dl0[d] + c.dls[d] * (cn[d] - 1);
    const int hi = c.dls[d] > 0 ? c.dl0[d] + c.dls[d] * (cn[d] - 1) : c.dl0[d];
    dmin[d] = lo;
    h[d] = (b[d] - 1) * c.m[d] + (hi - lo) + 1;
    if (lo < -127 || hi > 127) return false;
  }
  const long long P = (long long)h[0] * h[1] * h[2];
  if (P > halo_max_positions()) return false;
  hp.g = p;
  hp.bd = bd; hp.bh = bh; hp.bw = bw; hp.lbh = ilog2(bh); hp.lbw = ilog2(bw);
  hp.hd = h[0]; hp.hh = h[1]; hp.hw = h[2];
  hp.h0d = c.o[0] + dmin[0]; hp.h0h = c.o[1] + dmin[1]; hp.h0w = c.o[2] + dmin[2];
  hp.dmin_d = dmin[0]; hp.dmin_h = dmin[1]; hp.dmin_w = dmin[2];
  hp.nbd = cdiv(c.q[0], bd); hp.nbh = cdiv(c.q[1], bh); hp.nbw = cdiv(c.q[2], bw);
  hp.P = (int)P;
  hp.nchunks = cdiv(c.srcC, 16);
  hp.chunks_per_split = hp.nchunks;
  hp.Mrows = pack_rows(c.M);
  const long long cs = (long long)p.SD * p.SH * p.SW * (math == 3 ? 2 : 4);
  const long long pk = (long long)hp.nchunks * c.ntaps * hp.Mrows * halo_row_bytes(math);
  if (cs > 0x7fffffffLL || pk > 0xfffff000LL) return false;
  hp.cs_bytes = (unsigned)cs;
  hp.pack_bytes = (unsigned)pk;
  return true;
}

// heuristic box for `bn` columns: least (padding of the iteration grid) x (halo positions per output), W runs >= 8 preferred
inline bool halo_pick_box(const ClassInfo& c, const IgemmParams& p, int bn, int math, int& bd, int& bh, int& bw, double& cost_out) {
  double best = 1e300;
  for (int d = 1; d <= bn; d <<= 1)
    for (int h = 1; d * h <= bn; h <<= 1) {
      const int w = bn / (d * h);
      if (d > 2 * c.q[0] || h > 2 * c.q[1] || w > 2 * c.q[2]) continue;      // at most one doubling past the grid
      HaloParams hp;
      if (!halo_geometry(c, p, d, h, w, math, hp)) continue;
      const double cover = (double)cdiv(c.q[0], d) * d * cdiv(c.q[1], h) * h * (double)cdiv(c.q[2], w) * w /
                           ((double)c.q[0] * c.q[1] * c.q[2]);
      const double cost = cover * (1.0 + 0.08 * hp.P / bn) * (w >= 8 ? 1.0 : (w >= 4 ? 1.1 : 1.3));
      if (cost < best) { best = cost; bd = d; bh = h; bw = w; }
    }
  cost_out = best;
  return best < 1e299;
}

// tune code of a pass: rows (32..160) | 1024 = 256-column float4 variant | 2048 = LDS-halo kernel (box in tune_*_box:
// bd | bh << 8 | bw << 16; bd*bh*bw = 128 or 256)
inline IgemmCfg cfg_for(const gca_conv_geom* g, int which, const ClassInfo& c, const IgemmParams& p, size_t nclasses) {
  int fbm, fs, tail, box;
  tune_of(g, which, fbm, fs, tail, box);
  const int math = resolve_math(which == 0 ? g->tune_fwd_math : g->tune_dgrad_math, g->act_f16);
  // ---- stem kernel (forward, C <= 4, stride-2 W windows): forced by tune code 4096, or by the heuristic when un-tuned
  if (which == 0 && ((fbm & 4096) || (fbm == 0 && stem_heuristic_on()))) {
    StemParams sp;
    if (stem_geometry(g, p, math, sp))
      return IgemmCfg{32 * sp.wm, 512 / sp.wm, 1, sp.nsteps, 0, 0, math, 2, sp.bd, sp.bh, sp.bw, math == 3};
  }
  if (fbm & 4096) { fbm = 0; fs = 0; tail = 0; }           // not runnable as asked
  // ---- pointwise fp16 GEMM kernel: forced by tune code 8192, or by the heuristic when un-tuned
  if ((fbm & 8192) || (fbm == 0 && pw_heuristic_on())) {
    PwParams pw;
    if (pw_geometry(g, which, p, pw)) return IgemmCfg{128, 128, 1, pw.Kpad / 32, 0, 0, 3, 3, 0, 0, 0, 1};
  }
  if (fbm & 8192) { fbm = 0; fs = 0; tail = 0; }
  // ---- LDS-halo kernel: forced by the tune code, or by the heuristic for multi-tap classes with enough channels and tiles
  {
    int bd = box & 255, bh = (box >> 8) & 255, bw = (box >> 16) & 255, rows = fbm & 1023;
    bool use = false;
    HaloParams hp;
    if ((fbm & 2048) && !(fbm & (4096 | 8192))) {
      use = bd > 0 && halo_geometry(c, p, bd, bh, bw, math, hp);
    } else if (fbm == 0 && halo_heuristic_on() && (c.ntaps >= 3 || math == 3) && c.srcC >= 32 && p.Ntot >= 128LL * 192) {
      double cost = 0, cost2 = 0;
      int d2, h2, w2;
      const bool wide = p.DK <= 96 && halo_pick_box(c, p, 256, math, d2, h2, w2, cost2);
      use = halo_pick_box(c, p, 128, math, bd, bh, bw, cost) && cost < 1.6;
      if (wide && cost2 < 1.6 && (!use || cost2 <= cost * 1.05)) { bd = d2; bh = h2; bw = w2; use = true; }
      if (use) use = halo_geometry(c, p, bd, bh, bw, math, hp);
      rows = 0;
    }
    if (use) {
      const int bn = bd * bh * bw, tmax = bn == 256 ? 3 : 5;
      if (rows == 0) {                                   // fewest padded rows, taller tile on a tie
        int bestpad = 1 << 30;
        for (int tm = 1; tm <= tmax; ++tm) {
          const int pad = cdiv(p.DK, 32 * tm) * 32 * tm;
          if (pad <= bestpad) { bestpad = pad; rows = 32 * tm; }
        }
      }
      if (rows >= 32 && rows <= 32 * (bn == 256 ? 4 : 5) && rows % 32 == 0 &&
          halo_lds_bytes(rows, math, hp.P) <= (size_t)(160 << 10) - 1024) {
        IgemmCfg cf{rows, bn, 1, hp.nchunks, 0, 0, math, 1, bd, bh, bw, math == 3};
        const long long tiles = (long long)cdiv(p.DK, rows) * g->N * hp.nbd * hp.nbh * hp.nbw;
        int sp = 1;
        if (fs > 0) sp = fs;
        else if (tiles < 2 * NUM_CU && hp.nchunks >= 8 && (long long)p.DK * p.Ntot <= (1LL << 20)) {
          long long want = gca_ceil_div(2 * NUM_CU, tiles);
          if (want > hp.nchunks / 4) want = hp.nchunks / 4;
          if (want > 16) want = 16;
          if (want > 1) sp = (int)want;
        }
        if (sp > hp.nchunks) sp = hp.nchunks;
        if (sp < 1) sp = 1;
        cf.kt_per_split = cdiv(hp.nchunks, sp);
        cf.splits = cdiv(hp.nchunks, cf.kt_per_split);
        return cf;
      }
    }
    if (fbm & 2048) { fbm = 0; fs = 0; tail = 0; }       // not runnable as asked: fall back to the gather kernels' heuristic
  }
  IgemmCfg cf = choose_cfg(p.DK, p.Ntot, p.Kpad / BK, c.vec, fbm, fs);
  cf.math = math;
  cf.h = math == 3;
  if (cf.h) tail = 0;                                    // (the two-phase launch is built for fp32 storage only)
  // Two-phase launch (measured configurations only; single-class passes, 128-column tiles, no split-K): the first
  // `main_cols` column tiles run with the tall tile, the rest with a short one -- the last, partly filled wave of tall
  // workgroups (up to a quarter of the launch time on the layer-1 shapes) becomes a full wave of short ones.
  // tail code = short tile rows/32 | main column tiles << 8.
  if (tail > 0 && nclasses == 1 && cf.splits == 1 && cf.bn == 128) {
    const int tb = (tail & 255) * 32, mc = tail >> 8;
    const long long tilesN = gca_ceil_div(p.Ntot, 128);
    if (tb >= 32 && tb < cf.bm && mc > 0 && mc < tilesN) { cf.tail_bm = tb; cf.main_cols = mc; }
  }
  return cf;
}

inline long long halo_tiles_n(const gca_conv_geom* g, const ClassInfo& c, const IgemmCfg& cf) {
  return (long long)g->N * cdiv(c.q[0], cf.bd) * cdiv(c.q[1], cf.bh) * cdiv(c.q[2], cf.bw);
}

template <int TM, int BN, int FAST, bool VEC>
void launch_one(int math, dim3 grid, hipStream_t st, const float* src, const float* apack, const int2* table, const float* bias,
                float* dst, float* psum, float* psq, float* slab, const IgemmParams& p, bool h = false) {
  if (h)
    hipLaunchKernelGGL((conv_igemm_kernel<TM, BN, FAST, VEC, 3, true>), grid, dim3(256), 0, st, src, apack, table, bias,
                       dst, psum, psq, slab, p);
  else if (math == 1)
    hipLaunchKernelGGL((conv_igemm_kernel<TM, BN, FAST, VEC, 1>), grid, dim3(256), 0, st, src, apack, table, bias,
                       dst, psum, psq, slab, p);
  else if (math == 2)
    hipLaunchKernelGGL((conv_igemm_kernel<TM, BN, FAST, VEC, 2>), grid, dim3(256), 0, st, src, apack, table, bias,
                       dst, psum, psq, slab, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<TM, BN, FAST, VEC, 0>), grid, dim3(256), 0, st, src, apack, table, bias,
                       dst, psum, psq, slab, p);
}

template <int TM>
void launch_tm(const IgemmCfg& c, int fast, dim3 grid, hipStream_t st, const float* src, const float* apack,
               const int2* table, const float* bias, float* dst, float* psum, float* psq, float* slab,
               const IgemmParams& p) {
  if (c.bn == 256) launch_one<TM, 256, 1, true>(c.math, grid, st, src, apack, table, bias, dst, psum, psq, slab, p, c.h);
  else if (fast == 1) launch_one<TM, 128, 1, false>(c.math, grid, st, src, apack, table, bias, dst, psum, psq, slab, p, c.h);
  else if (fast == 2) launch_one<TM, 128, 2, false>(c.math, grid, st, src, apack, table, bias, dst, psum, psq, slab, p, c.h);
  else launch_one<TM, 128, 0, false>(c.math, grid, st, src, apack, table, bias, dst, psum, psq, slab, p, c.h);
}

inline int stat_parts(const IgemmCfg& c, long long Ntot, long long halo_tiles = 0) {
  if (c.splits > 1) return (int)gca_ceil_div(Ntot, FINISH_CHUNK);
  if (c.halo) return (int)halo_tiles;
  return (int)gca_ceil_div(Ntot, c.bn);
}

int launch_tiles(int bm, const IgemmCfg& c, int fast, dim3 grid, hipStream_t st, const float* src, const float* apack,
                 const int2* table, const float* bias, float* dst, float* ps, float* pq, float* slab, const IgemmParams& p) {
  switch (bm / 32) {
    case 1: launch_tm<1>(c, fast, grid, st, src, apack, table, bias, dst, ps, pq, slab, p); break;
    case 2: launch_tm<2>(c, fast, grid, st, src, apack, table, bias, dst, ps, pq, slab, p); break;
    case 3: launch_tm<3>(c, fast, grid, st, src, apack, table, bias, dst, ps, pq, slab, p); break;
    case 4: launch_tm<4>(c, fast, grid, st, src, apack, table, bias, dst, ps, pq, slab, p); break;
    default: launch_tm<5>(c, fast, grid, st, src, apack, table, bias, dst, ps, pq, slab, p); break;
  }
  return gca_launch_status();
}

int run_class(const IgemmCfg& c, int fast, const float* src, const float* apack, const int2* table, const float* bias,
              float* dst, float* psum, float* psq, float* slab, IgemmParams p, hipStream_t st) {
  const int tilesN = (int)gca_ceil_div(p.Ntot, c.bn);
  p.splits = c.splits; p.kt_per_split = c.kt_per_split;
  p.P = stat_parts(c, p.Ntot);
  if (c.splits > 1 && !slab) return GCA_EINVAL;
  float* ps = c.splits > 1 ? nullptr : psum;
  float* pq = c.splits > 1 ? nullptr : psq;
  if (c.tail_bm > 0) {                       // two-phase: tall tiles over [0, main_cols), short tiles over the rest
    if ((fast == 1 || fast == 2) && c.tail_bm <= 64) {          // one launch (conv_igemm_2phase_kernel)
      p.tilesM = (int)gca_ceil_div(p.DK, c.bm); p.tilesN = c.main_cols; p.tileN_off = 0;
      const int tmB = (int)gca_ceil_div(p.DK, c.tail_bm), tnB = tilesN - c.main_cols;
      const long long nA = (long long)p.tilesM * p.tilesN, nB = (long long)tmB * tnB;
      if (nA + nB > 0x7fffffffLL) return GCA_EINVAL;
      bool done = true;
#define GCA_2PL(A, B, F, M)                                                                                                  \
  hipLaunchKernelGGL((conv_igemm_2phase_kernel<A, B, F, M>), dim3((unsigned)(nA + nB)), dim3(256), 0, st, src, apack, table,  \
                     bias, dst, ps, pq, p, (int)nA, tmB, tnB)
#define GCA_2P(A, B)                                                                                                          \
  if (c.bm == 32 * A && c.tail_bm == 32 * B) {                                                                                \
    if (c.math == 1) { if (fast == 1) GCA_2PL(A, B, 1, 1); else GCA_2PL(A, B, 2, 1); }                                        \
    else if (c.math == 2) { if (fast == 1) GCA_2PL(A, B, 1, 2); else GCA_2PL(A, B, 2, 2); }                                   \
    else { if (fast == 1) GCA_2PL(A, B, 1, 0); else GCA_2PL(A, B, 2, 0); }                                                    \
  } else
      GCA_2P(2, 1) GCA_2P(3, 1) GCA_2P(3, 2) GCA_2P(4, 1) GCA_2P(4, 2) GCA_2P(5, 1) GCA_2P(5, 2) { done = false; }
#undef GCA_2P
#undef GCA_2PL
      if (done) return gca_launch_status();
    }
    p.tilesM = (int)gca_ceil_div(p.DK, c.bm); p.tilesN = c.main_cols; p.tileN_off = 0;
    int rc = launch_tiles(c.bm, c, fast, dim3((unsigned)((long long)p.tilesM * p.tilesN)), st, src, apack, table, bias, dst, ps, pq,
                          slab, p);
    if (rc) return rc;
    p.tilesM = (int)gca_ceil_div(p.DK, c.tail_bm); p.tilesN = tilesN - c.main_cols; p.tileN_off = c.main_cols;
    return launch_tiles(c.tail_bm, c, fast, dim3((unsigned)((long long)p.tilesM * p.tilesN)), st, src, apack, table, bias, dst, ps,
                        pq, slab, p);
  }
  p.tilesM = (int)gca_ceil_div(p.DK, c.bm);
  p.tilesN = tilesN;
  p.tileN_off = 0;
  const long long nblk = (long long)p.tilesM * p.tilesN * c.splits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return GCA_EINVAL;
  int rc = launch_tiles(c.bm, c, fast, dim3((unsigned)nblk), st, src, apack, table, bias, dst, ps, pq, slab, p);
  if (rc || c.splits == 1) return rc;
  if (t_leave_slabs) { t_left_splits = c.splits; return gca_launch_status(); }      // gca_conv_fwd_slabs: the consumer folds them
  if (c.h)
    hipLaunchKernelGGL(conv_splitk_finish_kernel<_Float16>, dim3((unsigned)p.DK, (unsigned)p.P), dim3(256), 0, st, slab, c.splits,
                       bias, reinterpret_cast<_Float16*>(dst), psum, psq, p);
  else
    hipLaunchKernelGGL(conv_splitk_finish_kernel<float>, dim3((unsigned)p.DK, (unsigned)p.P), dim3(256), 0, st, slab, c.splits,
                       bias, dst, psum, psq, p);
  return gca_launch_status();
}

// One class on the LDS-halo kernels.  `tapdelta` = the 64-entry tap-delta table that follows the class's gather rows.
int run_class_halo(const gca_conv_geom* g, const ClassInfo& c, const IgemmCfg& cf, const float* src, const float* apack,
                   const int2* table, const float* bias, float* dst, float* psum, float* psq, float* slab, IgemmParams p,
                   hipStream_t st, const float* in_scale = nullptr, const float* in_shift = nullptr) {
  HaloParams hp;
  if (!halo_geometry(c, p, cf.bd, cf.bh, cf.bw, cf.math, hp)) return GCA_EINVAL;
  hp.in_scale = in_scale; hp.in_shift = in_shift;
  if (cf.splits > 1 && !slab) return GCA_EINVAL;
  hp.g.splits = cf.splits; hp.g.kt_per_split = cf.kt_per_split;
  hp.chunks_per_split = cf.kt_per_split;
  hp.g.tilesM = cdiv(p.DK, cf.bm);
  const long long tn = halo_tiles_n(g, c, cf);
  if (tn > 0x7fffffffLL) return GCA_EINVAL;
  hp.g.tilesN = (int)tn; hp.g.tileN_off = 0;
  hp.g.P = stat_parts(cf, p.Ntot, tn);
  HaloCfg hc{cf.bm, cf.bn, cf.bd, cf.bh, cf.bw, cf.splits, cf.kt_per_split, cf.math};
  int rc = halo_launch(hc, hp, src, reinterpret_cast<const unsigned char*>(apack), reinterpret_cast<const int*>(table + p.Kpad),
                       bias, dst, cf.splits > 1 ? nullptr : psum, cf.splits > 1 ? nullptr : psq, slab, st);
  if (rc || cf.splits == 1) return rc;
  if (t_leave_slabs) { t_left_splits = cf.splits; return gca_launch_status(); }     // gca_conv_fwd_slabs: the consumer folds them
  p.splits = cf.splits; p.P = hp.g.P;
  if (cf.h)
    hipLaunchKernelGGL(conv_splitk_finish_kernel<_Float16>, dim3((unsigned)p.DK, (unsigned)p.P), dim3(256), 0, st, slab, cf.splits,
                       bias, reinterpret_cast<_Float16*>(dst), psum, psq, p);
  else
    hipLaunchKernelGGL(conv_splitk_finish_kernel<float>, dim3((unsigned)p.DK, (unsigned)p.P), dim3(256), 0, st, slab, cf.splits,
                       bias, dst, psum, psq, p);
  return gca_launch_status();
}

// A pointwise class on the fp16 GEMM kernel.
int run_class_pw(const gca_conv_geom* g, int which, const float* src, const float* apack, const float* bias, float* dst, float* psum,
                 float* psq, const IgemmParams& p, hipStream_t st) {
  PwParams pw;
  if (!pw_geometry(g, which, p, pw)) return GCA_EINVAL;
  return pw_launch(pw, src, reinterpret_cast<const unsigned char*>(apack), bias, dst, psum, psq, st);
}

// The forward class on the stem kernel.
int run_class_stem(const gca_conv_geom* g, const IgemmCfg& cf, const float* src, const float* apack, const float* bias, float* dst,
                   float* psum, float* psq, const IgemmParams& p, hipStream_t st) {
  StemParams sp;
  if (!stem_geometry(g, p, cf.math, sp) || sp.bd != cf.bd || sp.bh != cf.bh || sp.bw != cf.bw) return GCA_EINVAL;
  sp.g.splits = 1; sp.g.kt_per_split = sp.nsteps;
  sp.g.tilesM = cdiv(p.DK, 32 * sp.wm);
  const long long tn = (long long)g->N * sp.nbd * sp.nbh * sp.nbw;
  if (tn > 0x7fffffffLL) return GCA_EINVAL;
  sp.g.tilesN = (int)tn; sp.g.tileN_off = 0;
  sp.g.P = (int)tn;
  return stem_launch(cf.math, sp, src, reinterpret_cast<const unsigned char*>(apack), bias, dst, psum, psq, st);
}

int64_t ws_bytes_for(const gca_conv_geom* g, int which) {
  std::vector<ClassInfo> cls;
  build_classes(g, which, cls);
  int64_t need = 0;
  for (const ClassInfo& c : cls) {
    if (c.ntaps == 0) continue;
    IgemmParams p{};
    class_params(g, which, c, p);
    const IgemmCfg cf = cfg_for(g, which, c, p, cls.size());
    if (cf.splits > 1) {
      const int64_t b = (int64_t)cf.splits * p.DK * p.Ntot * (int64_t)sizeof(float);
      if (b > need) need = b;
    }
  }
  return need;
}

int g_conv_math = -1;       // -1: not read yet (GCA_CONV_MATH)

}  // namespace

namespace gca_conv {
int conv_math() {
  if (g_conv_math < 0) {
    const char* e = getenv("GCA_CONV_MATH");
    g_conv_math = !e ? 2 : (!strcmp(e, "bf16x3") || !strcmp(e, "1")) ? 1 : (!strcmp(e, "f32") || !strcmp(e, "0")) ? 0 : 2;
  }
  return g_conv_math;
}
}  // namespace gca_conv

extern "C" {

int gca_version(void) { return 15; }

int gca_set_conv_math(int mode) {
  if (mode < 0 || mode > 2) return GCA_EINVAL;
  g_conv_math = mode;
  return GCA_OK;
}
int gca_get_conv_math(void) { return conv_math(); }

int64_t gca_conv_pack_elems(const gca_conv_geom* g, int which) {
  if (!geom_ok(g) || (which != 0 && which != 1)) return GCA_EINVAL;
  std::vector<ClassInfo> cls;
  build_classes(g, which, cls);
  int64_t n = 0;
  for (const ClassInfo& c : cls) n += pack_reserve(c);
  return n > 0 ? n : 64;
}

static void pack_params_of(const gca_conv_geom* g, int which, const ClassInfo& c, size_t nclasses, PackParams& p) {
  const int T = taps(g);
  {
    IgemmParams ip{};
    class_params(g, which, c, ip);
    const IgemmCfg cf = cfg_for(g, which, c, ip, nclasses);
    p.fmt = cf.halo; p.math = cf.math; p.SC = c.srcC; p.nsteps = cf.halo == 2 ? cf.kt_per_split : cdiv(c.srcC, 16) * c.ntaps;
  }
  const bool pwfmt = p.fmt == 3;
  p.Kred = (int)c.Kred; p.M = c.M; p.Kpad = pwfmt ? (int)gca_round_up(c.srcC, 32) : (int)c.Kpad; p.Mrows = pack_rows(c.M);
  p.ntaps = p.fmt == 2 ? g->kd * g->kh : c.ntaps; p.nb = c.nb; p.nc = c.nc;       // (stem pack: reduction rows per channel)
  p.k0d = c.k0[0]; p.k0h = c.k0[1]; p.k0w = c.k0[2]; p.sd = c.ks[0]; p.sh = c.ks[1]; p.sw = c.ks[2];
  p.KH = g->kh; p.KW = g->kw; p.T = T;
  if (which == 0) { p.s_ch = T; p.s_m = (long long)g->C * T; }       // W[m=ko][ch=c][tap]
  else { p.s_ch = (long long)g->C * T; p.s_m = T; }                  // W[ch=ko][m=c][tap]
}

int64_t gca_conv_pack_jobs_host(const gca_conv_geom* g, int which, const float* w, float* packed, void* jobs_out) {
  if (!geom_ok(g) || (which != 0 && which != 1)) return GCA_EINVAL;
  std::vector<ClassInfo> cls;
  build_classes(g, which, cls);
  int64_t n = 0;
  for (const ClassInfo& c : cls) {
    if (c.ntaps == 0) continue;
    if (jobs_out) {
      if (!w || !packed) return GCA_EINVAL;
      PackJob j{};
      pack_params_of(g, which, c, cls.size(), j.p);
      j.w = w; j.packed = packed + c.pack_off;
      j.first_block = 0;
      j.nblocks = (int)gca_ceil_div(pack_items(j.p), PACK_CHUNK);
      unsigned char* dst = reinterpret_cast<unsigned char*>(jobs_out) + (size_t)n * GCA_PACK_JOB_BYTES;
      memset(dst, 0, GCA_PACK_JOB_BYTES);
      memcpy(dst, &j, sizeof(j));
    }
    ++n;
  }
  return n;
}

int64_t gca_conv_pack_jobs_finalize_host(void* jobs, int64_t njobs) {
  if (!jobs || njobs <= 0) return GCA_EINVAL;
  int64_t first = 0;
  for (int64_t i = 0; i < njobs; ++i) {
    PackJob j;
    unsigned char* rec = reinterpret_cast<unsigned char*>(jobs) + (size_t)i * GCA_PACK_JOB_BYTES;
    memcpy(&j, rec, sizeof(j));
    if (j.nblocks <= 0 || first + j.nblocks > 0x7fffffffLL) return GCA_EINVAL;
    j.first_block = (int)first;
    first += j.nblocks;
    memcpy(rec, &j, sizeof(j));
  }
  return first;
}

int gca_conv_pack_batched(const void* jobs_dev, int64_t njobs, int64_t total_blocks, void* stream) {
  if (!jobs_dev || njobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) return GCA_EINVAL;
  hipLaunchKernelGGL(conv_pack_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const unsigned char*>(jobs_dev), (int)njobs);
  return gca_launch_status();
}

int gca_conv_pack(const gca_conv_geom* g, int which, const float* w, float* packed, void* stream) {
  if (!geom_ok(g) || (which != 0 && which != 1) || !w || !packed) return GCA_EINVAL;
  std::vector<ClassInfo> cls;
  build_classes(g, which, cls);
  for (const ClassInfo& c : cls) {
    if (c.ntaps == 0) continue;
    PackParams p{};
    pack_params_of(g, which, c, cls.size(), p);
    long long blocks = gca_ceil_div(pack_items(p), 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                       packed + c.pack_off, p);
  }
  return gca_launch_status();
}

int64_t gca_conv_table_rows(const gca_conv_geom* g, int which) {
  if (!geom_ok(g) || which < 0 || which > 2) return GCA_EINVAL;
  if (which == 2) return gca_round_up((int64_t)g->C * taps(g) + TABLE_PAD_W, 64);   // any wgrad tile (<= 192 wide) stays inside
  std::vector<ClassInfo> cls;
  build_classes(g, which, cls);
  int64_t n = 0;
  for (const ClassInfo& c : cls) if (c.ntaps) n += c.Kpad + 32;
  return n > 0 ? n : 32;
}

int gca_conv_table_build_host(const gca_conv_geom* g, int which, int32_t* t) {
  if (!geom_ok(g) || which < 0 || which > 2 || !t) return GCA_EINVAL;
  const int64_t HW = (int64_t)g->H * g->W, DHW = HW * g->D;
  if (which == 2) {                       // wgrad: one row per (c, tap) with the forward deltas
    const int T = taps(g);
    const int64_t rows = gca_conv_table_rows(g, 2), kred = (int64_t)g->C * T;
    for (int64_t k = 0; k < rows; ++k) {
      int32_t off = 0, pk = pack_row_meta(0, 0, 0, 0, 63);
      if (k < kred) {
        const int ch = (int)(k / T), tap = (int)(k % T);
        const int a = tap / (g->kh * g->kw), r = tap % (g->kh * g->kw), b = r / g->kw, c = r % g->kw;
        off = (int32_t)(uint32_t)((uint64_t)((ch * DHW + a * HW + (int64_t)b * g->W + c) * 4) & 0xffffffffu);   // BYTES
        pk = pack_row_meta(a, b, c, 1, tap < 63 ? tap : 63);
      }
      t[2 * k] = off; t[2 * k + 1] = pk;
    }
    return GCA_OK;
  }
  std::vector<ClassInfo> cls;
  build_classes(g, which, cls);
  const int64_t SHW = which == 0 ? HW : (int64_t)g->OH * g->OW;
  const int64_t SW = which == 0 ? g->W : g->OW;
  const int64_t SDHW = which == 0 ? DHW : SHW * g->OD;
  for (const ClassInfo& c : cls) {
    if (c.ntaps == 0) continue;
    int32_t* rows = t + 2 * c.table_off;
    for (int64_t k = 0; k < c.Kpad; ++k) {
      int32_t off = 0, pk = pack_row_meta(0, 0, 0, 0, 63);
      if (k < c.Kred) {
        const int ch = (int)(k / c.ntaps), tl = (int)(k % c.ntaps);
        const int a = tl / (c.nb * c.nc), r = tl % (c.nb * c.nc), b = r / c.nc, cc = r % c.nc;
        const int dd = c.dl0[0] + c.dls[0] * a, dh = c.dl0[1] + c.dls[1] * b, dw = c.dl0[2] + c.dls[2] * cc;
        off = (int32_t)(uint32_t)((uint64_t)((int64_t)(ch * SDHW + dd * SHW + dh * SW + dw) * 4) & 0xffffffffu);   // BYTES
        pk = pack_row_meta(dd, dh, dw, 1, tl < 63 ? tl : 63);
      }
      rows[2 * k] = off; rows[2 * k + 1] = pk;
    }
    int32_t* tt = rows + 2 * c.Kpad;       // 64-int tap-delta table for the validity mask
    for (int tl = 0; tl < 64; ++tl) {
      int32_t pk = 0;
      if (tl < c.ntaps && c.ntaps <= FAST_MAX_TAPS) {
        const int a = tl / (c.nb * c.nc), r = tl % (c.nb * c.nc), b = r / c.nc, cc = r % c.nc;
        pk = pack_row_meta(c.dl0[0] + c.dls[0] * a, c.dl0[1] + c.dls[1] * b, c.dl0[2] + c.dls[2] * cc, 1, 0);
      }
      tt[tl] = pk;
    }
  }
  return GCA_OK;
}

int64_t gca_conv_pack_layout(const gca_conv_geom* g, int which) {
  if (!geom_ok(g) || which < 0 || which > 1) return GCA_EINVAL;
  std::vector<ClassInfo> cls;
  build_classes(g, which, cls);
  int64_t sig = 0;
  for (const ClassInfo& c : cls) {
    int code = 0;                                  // 0: k-major fp32 rows; 4 + arithmetic: LDS-halo layout; 1..3: stem layout
    if (c.ntaps) {
      IgemmParams p{};
      class_params(g, which, c, p);
      const IgemmCfg cf = cfg_for(g, which, c, p, cls.size());
      if (cf.halo == 3) code = 8;                   // pointwise fp16 GEMM operand
      else if (cf.halo == 2) code = cf.math;        // stem layout (arithmetic 1..3)
      else if (cf.halo) code = 4 + cf.math;
    }
    sig = sig * 16 + code;
    if (sig > (1LL << 56)) sig %= 1000000007LL;    // (more than 14 classes: a hash is enough)
  }
  return sig;
}

int gca_conv_kernel_cfg(const gca_conv_geom* g, int which, int32_t* out4) {
  if (!geom_ok(g) || which < 0 || which > 1 || !out4) return GCA_EINVAL;
  std::vector<ClassInfo> cls;
  build_classes(g, which, cls);
  for (const ClassInfo& c : cls) {
    if (c.ntaps == 0) continue;
    IgemmParams p{};
    class_params(g, which, c, p);
    const IgemmCfg cf = cfg_for(g, which, c, p, cls.size());
    out4[0] = cf.bm; out4[1] = cf.bn; out4[2] = cf.splits; out4[3] = (int)cls.size() | (fast_of(c.ntaps) << 8) | (c.vec << 10) | (cf.math << 12) | ((cf.halo == 1) << 14) | (cf.h << 15) | ((cf.halo == 2) << 16) | ((cf.halo == 3) << 17);
    return GCA_OK;
  }
  return GCA_EINVAL;
}

int64_t gca_conv_fwd_stat_parts(const gca_conv_geom* g) {
  if (!geom_ok(g)) return GCA_EINVAL;
  std::vector<ClassInfo> cls;
  build_classes(g, 0, cls);
  IgemmParams p{};
  class_params(g, 0, cls[0], p);
  const IgemmCfg cf = cfg_for(g, 0, cls[0], p, 1);
  if (cf.halo == 3) return (int64_t)g->N * cdiv((int)((long long)g->OD * g->OH * g->OW), 128);
  return stat_parts(cf, p.Ntot, cf.halo ? halo_tiles_n(g, cls[0], cf) : 0);
}

int64_t gca_conv_fwd_ws_bytes(const gca_conv_geom* g) { return geom_ok(g) ? ws_bytes_for(g, 0) : GCA_EINVAL; }
int64_t gca_conv_dgrad_ws_bytes(const gca_conv_geom* g) { return geom_ok(g) ? ws_bytes_for(g, 1) : GCA_EINVAL; }

int gca_conv_fwd(const gca_conv_geom* g, const void* x_, const float* wpack, const int32_t* table,
                 const float* bias, void* y_, float* stat_sum, float* stat_sq, void* ws, void* stream) {
  // (activations are fp32, or fp16 when g->act_f16: the kernels below take them as opaque buffers and index in bytes)
  const float* x = reinterpret_cast<const float*>(x_);
  float* y = reinterpret_cast<float*>(y_);
  if (!geom_ok(g) || !x || !wpack || !table || !y) return GCA_EINVAL;
  if ((stat_sum == nullptr) != (stat_sq == nullptr)) return GCA_EINVAL;
  std::vector<ClassInfo> cls;
  build_classes(g, 0, cls);
  const ClassInfo& c = cls[0];
  IgemmParams p{};
  class_params(g, 0, c, p);
  p.accumulate = 0;
  const IgemmCfg cf = cfg_for(g, 0, c, p, 1);
  if (cf.halo == 3) return run_class_pw(g, 0, x, wpack, bias, y, stat_sum, stat_sq, p, (hipStream_t)stream);
  if (cf.halo == 2) return run_class_stem(g, cf, x, wpack, bias, y, stat_sum, stat_sq, p, (hipStream_t)stream);
  if (cf.halo)
    return run_class_halo(g, c, cf, x, wpack, reinterpret_cast<const int2*>(table), bias, y, stat_sum, stat_sq,
                          reinterpret_cast<float*>(ws), p, (hipStream_t)stream);
  return run_class(cf, fast_of(c.ntaps), x, wpack, reinterpret_cast<const int2*>(table), bias,
                   y, stat_sum, stat_sq, reinterpret_cast<float*>(ws), p, (hipStream_t)stream);
}

int gca_conv_fwd_slabs(const gca_conv_geom* g, const void* x, const float* wpack, const int32_t* table, void* ws,
                       int32_t* out_splits, void* stream) {
  if (!out_splits || !ws || !geom_ok(g) || g->act_f16) return GCA_EINVAL;
  t_leave_slabs = true; t_left_splits = 0;
  // (y is never written in this mode; the pointer only has to be non-null for the argument check)
  const int rc = gca_conv_fwd(g, x, wpack, table, nullptr, ws, nullptr, nullptr, ws, stream);
  t_leave_slabs = false;
  if (rc) return rc;
  if (t_left_splits < 2) return GCA_EINVAL;             // this launch shape does not split: the caller must use gca_conv_fwd
  *out_splits = t_left_splits;
  return GCA_OK;
}

int gca_conv_xf_ok(const gca_conv_geom* g) {
  if (!geom_ok(g) || g->act_f16) return 0;
  std::vector<ClassInfo> cls;
  build_classes(g, 0, cls);
  IgemmParams p{};
  class_params(g, 0, cls[0], p);
  const IgemmCfg cf = cfg_for(g, 0, cls[0], p, 1);
  if (cf.halo != 1 || cf.math == 3) return 0;
  const int wm = resolve_math(g->tune_wgrad_math, 0);
  return g->tune_wgrad_tile >= 11 && g->tune_wgrad_tile <= 12 && wgrad_ts_ok(g, g->tune_wgrad_tile, wm) ? 1 : 0;
}

int gca_conv_fwd_xf(const gca_conv_geom* g, const void* x_, const float* in_scale, const float* in_shift, const float* wpack,
                    const int32_t* table, const float* bias, void* y_, float* stat_sum, float* stat_sq, void* ws, void* stream) {
  const float* x = reinterpret_cast<const float*>(x_);
  float* y = reinterpret_cast<float*>(y_);
  if (!geom_ok(g) || g->act_f16 || !x || !in_scale || !in_shift || !wpack || !table || !y) return GCA_EINVAL;
  if ((stat_sum == nullptr) != (stat_sq == nullptr)) return GCA_EINVAL;
  std::vector<ClassInfo> cls;
  build_classes(g, 0, cls);
  const ClassInfo& c = cls[0];
  IgemmParams p{};
  class_params(g, 0, c, p);
  p.accumulate = 0;
  const IgemmCfg cf = cfg_for(g, 0, c, p, 1);
  if (cf.halo != 1) return GCA_EINVAL;                 // only the LDS-halo kernels stage their input through registers
  return run_class_halo(g, c, cf, x, wpack, reinterpret_cast<const int2*>(table), bias, y, stat_sum, stat_sq,
                        reinterpret_cast<float*>(ws), p, (hipStream_t)stream, in_scale, in_shift);
}

int gca_conv_dgrad(const gca_conv_geom* g, const void* dy_, const float* wpack, const int32_t* table,
                   void* dx_, int accumulate, void* ws, void* stream) {
  const float* dy = reinterpret_cast<const float*>(dy_);
  float* dx = reinterpret_cast<float*>(dx_);
  if (!geom_ok(g) || !dy || !wpack || !table || !dx) return GCA_EINVAL;
  if (g->x_batch_stride != 0 && g->x_batch_stride != (long long)g->C * g->D * g->H * g->W) return GCA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  std::vector<ClassInfo> cls;
  build_classes(g, 1, cls);
  bool any_empty = false;
  for (const ClassInfo& c : cls) any_empty |= c.ntaps == 0;
  if (any_empty && !accumulate) {          // positions no tap reaches (e.g. 1x1x1 stride 2) get an exact zero
    const long long n = (long long)g->N * g->C * g->D * g->H * g->W;
    const long long n32 = g->act_f16 ? n / 2 : n;                 // dx holds n elements of 2 (fp16 storage) or 4 bytes
    long long blocks = gca_ceil_div(n32 + 1, 1024);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<unsigned*>(dx), n32,
                       g->act_f16 ? (int)(n & 1) : 0);
  }
  const int2* tab = reinterpret_cast<const int2*>(table);
  for (const ClassInfo& c : cls) {
    if (c.ntaps == 0) continue;
    IgemmParams p{};
    class_params(g, 1, c, p);
    p.accumulate = accumulate ? 1 : 0;
    const IgemmCfg cf = cfg_for(g, 1, c, p, cls.size());
    int rc = cf.halo == 3 ? run_class_pw(g, 1, dy, wpack + c.pack_off, nullptr, dx, nullptr, nullptr, p, st)
           : cf.halo ? run_class_halo(g, c, cf, dy, wpack + c.pack_off, tab + c.table_off, nullptr, dx, nullptr, nullptr,
                                      reinterpret_cast<float*>(ws), p, st)
                     : run_class(cf, fast_of(c.ntaps), dy, wpack + c.pack_off,
                                 tab + c.table_off, nullptr, dx, nullptr, nullptr, reinterpret_cast<float*>(ws), p, st);
    if (rc) return rc;
  }
  return GCA_OK;
}

}  // extern "C"
