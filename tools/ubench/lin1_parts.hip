// Additive decomposition of gemm_f16x2_kernel (head.hip, lin1): the same kernel built with parts switched
// off (TT_LIN1_SKIP), random operands, timed with HIP events; M = images, N = 1000, K = 16384.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DTT_LIN1_SKIP=<mask> -o lin1_parts_<mask> lin1_parts.hip
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#include "../../scale_imagenet_amd/csrc/head.hip"

namespace ttnet {
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
int ensure_dynamic_lds(const void *kernel, size_t bytes) {
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -3;
}
}  // namespace ttnet

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 256, N = 1000, K = 16384;
  const size_t ae = ttnet::frag_elems((M + 255) / 256 * 256, K), be = ttnet::frag_elems(1024, K);
  uint16_t *A, *B;
  float *part;
  const int splits = ttnet::gemm_f16x2_splits(M, N, K / 16);
  hipMalloc(&A, ae * 2); hipMalloc(&B, be * 2); hipMalloc(&part, (size_t)splits * M * N * 4);
  std::vector<uint16_t> h(std::max(ae, be));
  uint64_t s = 88172645463325252ull;
  for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint16_t)(0x3000 + (s & 0x7FF)) | (uint16_t)((s >> 20) & 0x8000); }
  hipMemcpy(A, h.data(), ae * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), be * 2, hipMemcpyHostToDevice);
  // a 512 MiB buffer written between launches: the operands come from HBM, as in the forward
  char *flush; hipMalloc(&flush, 512u << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float tot = 0; const int reps = 20;
  for (int i = 0; i < reps + 2; ++i) {
    hipMemsetAsync(flush, i, 512u << 20, 0);
    hipEventRecord(e0, 0);
    ttnet::launch_gemm_f16x2(A, B, part, M, N, K, splits, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (i >= 2) tot += ms;
  }
  printf("skip=%d M=%d splits=%d: %.2f us per launch (cold operands)\n", TT_LIN1_SKIP, M, splits, 1e3 * tot / reps);
#ifdef TT_LIN1_STAMP
  {
    static unsigned long long h_st[512][2];
    (void)hipMemcpyFromSymbol(h_st, HIP_SYMBOL(ttnet::g_lin1_stamps), sizeof(h_st));
    std::vector<double> cyc, ns;
    for (int b = 0; b < 256; ++b) { cyc.push_back((double)h_st[b][0]); ns.push_back((double)h_st[b][1]); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ns.begin(), ns.end());
    printf("  main loop, median workgroup: %.0f shader cycles, %.0f ns -> %.2f GHz; %.0f cycles per k-step (64 k-steps)\n", cyc[128], ns[128],
           cyc[128] / ns[128], cyc[128] / 64.0);
  }
#endif
  return 0;
}
