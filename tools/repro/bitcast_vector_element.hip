// Reproducer for the hipcc (ROCm 7.2, gfx950) miscompile worked around in csrc/conv3d*.hip:
// __builtin_bit_cast applied to an ELEMENT of the vector returned by raw_buffer_load_b128 narrows the load to
// one dword and replicates it (expected output 1 2 3 4 5 6 7 8, observed 1 1 1 1 5 5 5 5).  Bit-casting the whole
// vector, or __uint_as_float on the elements, is fine.   hipcc --offload-arch=gfx950 -O3 bitcast_vector_element.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void bad(const float* p, float* o, int n) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, n * 4, 0x00020000);
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(threadIdx.x * 16), 0, 0);
  o[threadIdx.x * 4 + 0] = __builtin_bit_cast(float, v[0]); o[threadIdx.x * 4 + 1] = __builtin_bit_cast(float, v[1]);
  o[threadIdx.x * 4 + 2] = __builtin_bit_cast(float, v[2]); o[threadIdx.x * 4 + 3] = __builtin_bit_cast(float, v[3]);
}
__global__ void good(const float* p, float* o, int n) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, n * 4, 0x00020000);
  const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(threadIdx.x * 16), 0, 0));
  o[threadIdx.x * 4 + 0] = f.x; o[threadIdx.x * 4 + 1] = f.y; o[threadIdx.x * 4 + 2] = f.z; o[threadIdx.x * 4 + 3] = f.w;
}
int main() {
  float h[8], *d, *o;
  for (int i = 0; i < 8; ++i) h[i] = i + 1;
  hipMalloc(&d, 32); hipMalloc(&o, 32); hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
  for (int k = 0; k < 2; ++k) {
    if (k == 0) hipLaunchKernelGGL(bad, dim3(1), dim3(2), 0, 0, d, o, 8); else hipLaunchKernelGGL(good, dim3(1), dim3(2), 0, 0, d, o, 8);
    hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
    printf("%s:", k ? "whole-vector bit_cast" : "element bit_cast     ");
    for (int i = 0; i < 8; ++i) printf(" %g", h[i]);
    printf("\n");
  }
  return 0;
}
