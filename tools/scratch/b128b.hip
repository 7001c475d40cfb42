#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k1(const float* src, float4* out, unsigned bytes) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(threadIdx.x * 16), 0, 0);
  out[threadIdx.x] = make_float4(__builtin_bit_cast(float, v.x), __builtin_bit_cast(float, v.y), __builtin_bit_cast(float, v.z), __builtin_bit_cast(float, v.w));
}
__global__ void k2(const float* src, u32x4* out, unsigned bytes) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(threadIdx.x * 16), 0, 0);
}
