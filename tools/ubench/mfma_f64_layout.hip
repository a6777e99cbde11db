// Operand layouts of v_mfma_f64_16x16x4_f64 on gfx950, found empirically:
// A[m][k] = 1000 m + k (lane l supplies one element), B[k][n] = (k == K0) at n: D = A[:, K0] broadcast etc.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double *out, int test) {
  const int l = threadIdx.x;
  double a, b;
  if (test == 0) {            // hypothesis: A lane l = A[m = l%16][k = l/16]; B lane l = B[k = l/16][n = l%16]
    a = 100.0 * (l % 16) + (l / 16);        // A[m][k] = 100 m + k
    b = (l / 16 == 2) ? 1.0 : 0.0;          // B[k][n] = (k == 2): D[m][n] = A[m][2] = 100 m + 2
  } else {
    a = (l / 16 == 1) ? 1.0 : 0.0;          // A[m][k] = (k == 1): D[m][n] = B[1][n]
    b = 100.0 * (l / 16) + (l % 16);        // B[k][n] = 100 k + n -> D[m][n] = 100 + n
  }
  f64x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main() {
  double *d, h[256];
  hipMalloc(&d, sizeof(h));
  for (int t = 0; t < 2; ++t) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("test %d (expect D[m][n] = %s)\n", t, t == 0 ? "100 m + 2" : "100 + n");
    for (int l : {0, 1, 15, 16, 17, 33, 63})
      printf("  lane %2d: %7.1f %7.1f %7.1f %7.1f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
