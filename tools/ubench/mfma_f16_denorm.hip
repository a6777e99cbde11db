// Do the fp16 MFMAs of gfx950 honour subnormal inputs?  A = 2^-20 (fp16 subnormal), B = 2^10.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float av, float bv) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
  a[0] = (_Float16)av;            // k = 0 (lanes 0-31) and k = 8 (lanes 32-63)
  b[0] = (_Float16)bv;
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
  float *d, h[2];
  hipMalloc(&d, 8);
  const float cases[][2] = {{9.5367431640625e-07f, 1024.f}, {6.0e-8f, 16384.f}, {3.0517578125e-05f, 1.0f}, {1.0f, 9.5367431640625e-07f}};
  for (auto &c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a=%.6e (as fp16 %.6e) b=%.6e -> mfma sum over 2 k-slots = %.9e (expected %.9e)\n", c[0], h[1], c[1], h[0], 2.0 * h[1] * c[1]);
  }
  return 0;
}
