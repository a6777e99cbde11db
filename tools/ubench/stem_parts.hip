// Additive decomposition of stem_pc_kernel (stem.hip): the same kernel built with parts switched off
// (TT_STEM_SKIP) and, with -DTT_STEM_STAMP, s_memtime stamps of every period of both roles; random
// input, timed with HIP events.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DTT_STEM_SKIP=<mask> [-DTT_STEM_STAMP] -o stem_parts_<mask> stem_parts.hip
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "../../scale_imagenet_amd/csrc/stem.hip"

namespace ttnet {
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
float weight_prescale(const float *w, size_t n) {
  float amax = 0.f;
  for (size_t i = 0; i < n; ++i) amax = fmaxf(amax, fabsf(w[i]));
  int e;
  frexpf(amax, &e);
  return ldexpf(1.0f, 14 - e);
}
int ensure_dynamic_lds(const void *kernel, size_t bytes) {
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -3;
}
}  // namespace ttnet

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;

  const size_t xe = (size_t)n * 3 * 224 * 224;
  float *x;
  uint64_t *rp;
  uint16_t *wf;
  float *init;
  uint32_t *flag;
  hipMalloc(&x, xe * 4); hipMalloc(&rp, (size_t)n * 64 * 56 * 8); hipMalloc(&wf, ttnet::stem_split_weights_elems() * 2);
  hipMalloc(&init, 256); hipMalloc(&flag, 4);
  hipMemset(flag, 0, 4);
  std::vector<float> h(xe);
  uint64_t s = 88172645463325252ull;
  for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = ((float)(s & 0xFFFF) / 65536.f - 0.5f) * 4.f; }
  if (argc > 3 && atoi(argv[3]) == 1) std::fill(h.begin(), h.end(), 0.f);      // zero input: what the clock does without operand toggling
  hipMemcpy(x, h.data(), xe * 4, hipMemcpyHostToDevice);
  std::vector<float> w(64 * 147);
  std::vector<double> sc(64, 1.0), sh(64, 0.01);
  for (auto &v : w) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = ((float)(s & 0xFFFF) / 65536.f - 0.5f) * 0.16f; }
  std::vector<uint16_t> wfh(ttnet::stem_split_weights_elems());
  std::vector<float> inith(64);
  ttnet::stem_split_weights(w.data(), sc.data(), sh.data(), 64, wfh.data(), inith.data());
  hipMemcpy(wf, wfh.data(), wfh.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(init, inith.data(), 256, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) ttnet::launch_stem(x, false, nullptr, wf, init, rp, nullptr, n, 64, flag, 0);
  hipDeviceSynchronize();
  const int reps = 50;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) ttnet::launch_stem(x, false, nullptr, wf, init, rp, nullptr, n, 64, flag, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("skip=%d n=%d: %.2f us per launch\n", TT_STEM_SKIP, n, 1e3 * ms / reps);
#ifdef TT_STEM_STAMP
  static unsigned long long st[256][2][16];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(ttnet::g_stem_stamps), sizeof(st));
  for (int b : {0, 1, 100, 255}) {
    for (int role = 0; role < 2; ++role) {
      printf("block %3d %s:", b, role ? "prod" : "cons");
      for (int j = 1; j < 12; ++j) printf(" %6lld", (long long)(st[b][role][j] - st[b][role][0]));
      printf("\n");
    }
  }
#endif
  return 0;
}
