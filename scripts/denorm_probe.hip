// Does v_mfma_f32_16x16x32_f16 / _bf16 keep denormal inputs?  (The certificate's error norms are measured on the stored
// 16-bit values; if the matrix pipe flushed fp16 denormals to zero the image would have to be stored denormal-free.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float a_val, float* out) {
  f16x8 a, b;
  bf16x8 c, d;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)1.0f; c[i] = (__bf16)1e-39f; d[i] = (__bf16)1.0f; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, d, acc2, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; out[2] = acc2[0]; out[3] = (float)c[0]; }
}
int main() {
  float* d; float h[4];
  hipMalloc((void**)&d, 16);
  const float vals[3] = {3e-5f, 1e-6f, 6e-8f};           // all below the smallest normal fp16 (6.1e-5)
  for (float v : vals) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, v, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("fp16 input %.3e (stored %.6e): MFMA sum of 32 products with 1.0 = %.6e  (expected %.6e)\n", v, h[1], h[0], 32.0 * h[1]);
  }
  printf("bf16 input 1e-39 (stored %.6e): MFMA sum = %.6e (expected %.6e)\n", h[3], h[2], 32.0 * h[3]);
  return 0;
}
