#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k3(const float* src, float* out, unsigned bytes) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(threadIdx.x * 16), 0, 0));
  out[threadIdx.x * 4 + 0] = f.x * 2.f; out[threadIdx.x * 4 + 1] = f.y * 3.f; out[threadIdx.x * 4 + 2] = f.z; out[threadIdx.x * 4 + 3] = f.w + 1.f;
}
__global__ void k4(const float* src, float* out, unsigned bytes) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(threadIdx.x * 16), 0, 0);
  out[threadIdx.x * 4 + 0] = __uint_as_float(v[0]); out[threadIdx.x * 4 + 1] = __uint_as_float(v[1]);
  out[threadIdx.x * 4 + 2] = __uint_as_float(v[2]); out[threadIdx.x * 4 + 3] = __uint_as_float(v[3]);
}
