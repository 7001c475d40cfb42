// Shared helpers for the gfx950 kernels of libgca_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gca_hip.h"

#define GCA_WAVE 64

static inline int gca_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GCA_OK : GCA_ELAUNCH;
}

static inline int64_t gca_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t gca_round_up(int64_t a, int64_t b) { return gca_ceil_div(a, b) * b; }

// Division of a 31-bit unsigned by a runtime constant without the ~40-instruction integer divide:
// n / d == (umulhi(n, m) + ((n - umulhi(n, m)) >> 1)) >> (s - 1)   (Granlund-Montgomery round-up method).
struct gca_magic { unsigned m; int s; unsigned d; };
static inline gca_magic gca_make_magic(unsigned d) {
  gca_magic g{0u, 0, d};
  if (d <= 1) return g;
  int s = 0;
  while ((1ull << s) < d) ++s;
  g.s = s;
  g.m = (unsigned)((((1ull << s) - d) << 32) / d + 1);
  return g;
}
__device__ __forceinline__ unsigned gca_fdiv(unsigned n, gca_magic g) {
  if (g.d <= 1) return n;
  const unsigned t = __umulhi(n, g.m);
  return (t + ((n - t) >> 1)) >> (g.s - 1);
}

// XCD-aware block remap (8 XCDs, blocks are dealt round-robin): logical ids that are
// consecutive end up on the same XCD, so tiles sharing an operand panel share an L2.
// Bijective for any grid size.  Speed only, never correctness.
__device__ __forceinline__ int gca_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float gca_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double gca_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float gca_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float gca_block_sum256(float v, float* sh /* >= 4 floats */) {
  v = gca_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ double gca_block_sum256_d(double v, double* sh) {
  v = gca_wave_sum_d(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- activation storage: fp32, or IEEE fp16 (the fp16-storage path: 2 bytes per activation element in HBM, every kernel
// widens on load and rounds once on store; arithmetic, statistics and parameters stay fp32 / fp64)
typedef _Float16 gca_half;
template <typename T> struct gca_act;
template <> struct gca_act<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct gca_act<gca_half> {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ float ld(const gca_half* p) { return (float)*p; }
  static __device__ __forceinline__ void st(gca_half* p, float v) { *p = (gca_half)v; }
  static __device__ __forceinline__ float4 ld4(const gca_half* p) {
    const h4 v = *reinterpret_cast<const h4*>(p);
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
  }
  static __device__ __forceinline__ void st4(gca_half* p, float4 v) {
    h4 o; o.x = (gca_half)v.x; o.y = (gca_half)v.y; o.z = (gca_half)v.z; o.w = (gca_half)v.w;
    *reinterpret_cast<h4*>(p) = o;
  }
};
