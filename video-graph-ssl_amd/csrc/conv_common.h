// Shared by conv3d.hip (forward / dgrad implicit GEMM) and conv3d_wgrad.hip.
#pragma once
#include "gca_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace gca_conv {

// Conv arithmetic (include/gca_hip.h, gca_set_conv_math): 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = bf16x3, 2 = bf16x6:
// fp32 operands split in the kernel into 2 / 3 bf16 parts, products on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
// Defined in conv3d.hip (gca_set_conv_math / GCA_CONV_MATH).
int conv_math();
constexpr int math_parts(int math) { return math == 2 ? 3 : (math == 3 ? 1 : 2); }   // 16-bit parts per operand element (3 = fp16 storage: the element itself)
// arithmetic of one pass: the tune_*_math override (1 + m) when m is at least as accurate as the mode in force
inline int math_rank(int m) { return m == 0 ? 2 : m == 2 ? 1 : 0; }     // f32 > bf16x6 > bf16x3
// act_f16: the activations of this convolution are fp16 in HBM (gca_conv_geom.act_f16) -> arithmetic 3: v_mfma_f32_32x32x16_f16
// on the stored halves and on the weights rounded to fp16 (RNE) where they become an MFMA operand; nothing to pin
inline int resolve_math(int tune_math, int act_f16 = 0) {
  const int mode = conv_math();
  if (act_f16) return 3;
  if (tune_math >= 1 && tune_math <= 3 && math_rank(tune_math - 1) >= math_rank(mode)) return tune_math - 1;
  return mode;
}

// (x0, x1) -> packed bf16 pairs hi = bf16(x) and lo = bf16(x - hi)  (round to nearest even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ void split_bf16x2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
// three parts: x = hi + mid + lo up to 2^-27 |x| (both residuals are exact in fp32)
__device__ __forceinline__ void split_bf16x3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const f32x2 r2 = {r.x - __uint_as_float(mid << 16), r.y - __uint_as_float(mid & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

constexpr int BK = 16;          // GEMM-K tile of the forward/dgrad kernels
constexpr int WBK = 32;         // GEMM-K (spatial) tile of the wgrad kernel
constexpr int MPAD = 64;        // packed weights: M padded to this
constexpr int TABLE_PAD_W = 192;   // widest wgrad tile: the table holds this many invalid rows past C*taps
constexpr int FAST_MAX_TAPS = 62;
constexpr int NUM_CU = 256;

// row table entry: x = offset of the row's (channel, tap) inside one image of the gathered tensor (BYTES for the
// forward/dgrad tables, elements for the wgrad table); y = tap6 | valid<<6 | dd<<8 | dh<<16 | dw<<24 (signed
// bytes).  The tap id sits in the low bits so that v_bfe_i32(mask, y, 1) tests its validity bit directly.
__device__ __forceinline__ void decode_row(int2 e, int& off, int& dd, int& dh, int& dw, int& valid) {
  off = e.x;
  dd = (e.y << 16) >> 24;
  dh = (e.y << 8) >> 24;
  dw = e.y >> 24;
  valid = (e.y >> 6) & 1;
}
inline int pack_row_meta(int dd, int dh, int dw, int valid, int tap6) {
  return (tap6 & 63) | ((valid & 1) << 6) | ((dd & 0xff) << 8) | ((dh & 0xff) << 16) | (int)((unsigned)(dw & 0xff) << 24);
}

// Sum over the 32 lanes of each wave half with DPP adds (no LDS traffic).  The total lands in lanes 16..31 of
// the lower half and 48..63 of the upper half (row_bcast15 feeds rows 1 and 3).
__device__ __forceinline__ float half_wave_sum_hi(float v) {
#define GCA_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xf, false))
  v += GCA_DPP(v, 0xB1, 0xf);     // quad_perm [1,0,3,2]
  v += GCA_DPP(v, 0x4E, 0xf);     // quad_perm [2,3,0,1]
  v += GCA_DPP(v, 0x141, 0xf);    // row_half_mirror
  v += GCA_DPP(v, 0x140, 0xf);    // row_mirror   -> every lane of a 16-lane row holds the row total
  v += GCA_DPP(v, 0x142, 0xa);    // row_bcast15 into rows 1 and 3
#undef GCA_DPP
  return v;
}

inline bool geom_ok(const gca_conv_geom* g) {
  if (!g) return false;
  if (g->N <= 0 || g->C <= 0 || g->D <= 0 || g->H <= 0 || g->W <= 0 || g->K <= 0) return false;
  if (g->kd <= 0 || g->kh <= 0 || g->kw <= 0 || g->sd <= 0 || g->sh <= 0 || g->sw <= 0) return false;
  if (g->pd < 0 || g->ph < 0 || g->pw < 0) return false;
  if (g->kd > 127 || g->kh > 127 || g->kw > 127) return false;
  const int od = (g->D + 2 * g->pd - g->kd) / g->sd + 1;
  const int oh = (g->H + 2 * g->ph - g->kh) / g->sh + 1;
  const int ow = (g->W + 2 * g->pw - g->kw) / g->sw + 1;
  if (od != g->OD || oh != g->OH || ow != g->OW || od <= 0 || oh <= 0 || ow <= 0) return false;
  // the kernels address both tensors with 32-bit byte offsets from their base: < 2^30 elements (4 GiB) each
  const long long cdhw = (long long)g->C * g->D * g->H * g->W;
  if (g->x_batch_stride != 0 && g->x_batch_stride < cdhw) return false;
  const long long in_elems = (long long)g->N * (g->x_batch_stride ? g->x_batch_stride : cdhw);
  const long long out_elems = (long long)g->N * g->K * od * oh * ow;
  if (in_elems >= (1LL << 30) || out_elems >= (1LL << 30)) return false;
  for (int v : {g->tune_fwd_bm, g->tune_dgrad_bm}) if (v != 0 && !((v & 1023) % 32 == 0 && (v & 1023) >= 32 && (v & 1023) <= 160 && ((v >> 10) <= 2 || (v >> 10) == 4 || (v >> 10) == 8))) return false;   // | 1024 float4 gathers, | 2048 LDS halo, | 4096 stem, | 8192 pointwise fp16 GEMM
  for (int v : {g->tune_fwd_box, g->tune_dgrad_box}) if (v < 0 || v > 0xffffff) return false;
  if (g->act_f16 != 0 && g->act_f16 != 1) return false;
  for (int v : {g->tune_fwd_splits, g->tune_dgrad_splits, g->tune_wgrad_splits}) if (v < 0 || v > 1024) return false;
  if (g->tune_wgrad_tile < 0 || g->tune_wgrad_tile > 14) return false;     // 1..10 tile shapes of conv_wgrad_kernel, 11 / 12 the streaming temporal kernel, 13 the streaming (1,3,3) kernel, 14 the stem kernel
  for (int v : {g->tune_fwd_math, g->tune_dgrad_math, g->tune_wgrad_math}) if (v < 0 || v > 3) return false;
  for (int v : {g->tune_fwd_tail, g->tune_dgrad_tail}) if (v < 0 || (v != 0 && ((v & 255) < 1 || (v & 255) > 4))) return false;
  return true;
}

inline int taps(const gca_conv_geom* g) { return g->kd * g->kh * g->kw; }

// conv3d_wgrad_ts.hip: streaming weight-gradient kernel of the unit-stride temporal convolutions (tune_wgrad_tile 11 / 12)
bool wgrad_ts_ok(const gca_conv_geom* g, int tile, int math);
int wgrad_ts_splits(const gca_conv_geom* g, int tile, int want);
int wgrad_ts_launch(const gca_conv_geom* g, int tile, int math, int splits, const float* x, const float* dy, float* slab,
                    hipStream_t st, const float* in_scale = nullptr, const float* in_shift = nullptr);
// conv3d_wgrad_stem.hip: weight gradient of the <= 4-channel, stride-2 stem convolutions (tune_wgrad_tile 14)
bool wgrad_stem_ok(const gca_conv_geom* g, int math);
int wgrad_stem_splits(const gca_conv_geom* g, int want);
int wgrad_stem_launch(const gca_conv_geom* g, int math, int splits, const void* x, const void* dy, float* slab, hipStream_t st);
inline bool unit_stride(const gca_conv_geom* g) { return g->sd == 1 && g->sh == 1 && g->sw == 1; }


}  // namespace gca_conv
