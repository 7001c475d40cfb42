#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* src, float* out, unsigned bytes) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(threadIdx.x * 16), 0, 0);
  out[threadIdx.x * 4 + 0] = __builtin_bit_cast(float, v[0]);
  out[threadIdx.x * 4 + 1] = __builtin_bit_cast(float, v[1]);
  out[threadIdx.x * 4 + 2] = __builtin_bit_cast(float, v[2]);
  out[threadIdx.x * 4 + 3] = __builtin_bit_cast(float, v[3]);
  const unsigned s = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(threadIdx.x * 4) | (threadIdx.x == 5 ? -1 : 0), 0, 0);
  out[64 + threadIdx.x] = __builtin_bit_cast(float, s);
}
int main() {
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = i + 1;
  float *d, *o; hipMalloc(&d, 256); hipMalloc(&o, 512); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o, 256u);
  float r[128]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 32; ++i) printf("%g ", r[i]); printf("\n");
  for (int i = 0; i < 8; ++i) printf("%g ", r[64 + i]); printf("\n");
  return 0;
}
