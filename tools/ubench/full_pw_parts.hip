// Additive decomposition of full_pw_fast_kernel (gate_full.hip): the same kernel built with parts switched off
// (TT_FULLPW_SKIP), random bits and weights, timed with HIP events.  Geometry of the full model's first block
// (Block_conv3 at 56x56: 2 groups of 30 -> 240 -> 30), 512 images.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DTT_FULLPW_SKIP=<mask> -o full_pw_parts_<mask> full_pw_parts.hip
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../scale_imagenet_amd/csrc/gate_full.hip"

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
  const int n = argc > 1 ? atoi(argv[1]) : 512, H = 56, W = 56, C = 60, G = 2;
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  std::vector<uint64_t> x((size_t)n * C * H);
  for (auto &v : x) v = rnd() & ((1ull << W) - 1ull);
  std::vector<float> w1((size_t)G * 240 * 30), w2((size_t)G * 30 * 240);
  for (auto &v : w1) v = ((float)(rnd() & 0xFFFF) / 65536.f - 0.5f) * 0.36f;
  for (auto &v : w2) v = ((float)(rnd() & 0xFFFF) / 65536.f - 0.5f) * 0.13f;
  std::vector<double> s1(G * 240), t1(G * 240), s2(G * 30), t2(G * 30);
  for (auto &v : s1) v = 0.8 + (double)(rnd() & 0xFF) / 512.0;
  for (auto &v : t1) v = ((double)(rnd() & 0xFF) / 256.0 - 0.5) * 0.4;
  for (auto &v : s2) v = 0.8 + (double)(rnd() & 0xFF) / 512.0;
  for (auto &v : t2) v = ((double)(rnd() & 0xFF) / 256.0 - 0.5) * 0.4;
  uint64_t *dx, *dout;
  float *dw1, *dw2;
  double *ds1, *dt1, *ds2, *dt2;
  uint32_t *fix;
  hipMalloc(&dx, x.size() * 8); hipMalloc(&dout, x.size() * 8);
  hipMalloc(&dw1, w1.size() * 4); hipMalloc(&dw2, w2.size() * 4);
  hipMalloc(&ds1, s1.size() * 8); hipMalloc(&dt1, t1.size() * 8); hipMalloc(&ds2, s2.size() * 8); hipMalloc(&dt2, t2.size() * 8);
  hipMalloc(&fix, (64 + (size_t)G * n * H * W) * 4);
  hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(ds1, s1.data(), s1.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dt1, t1.data(), t1.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(ds2, s2.data(), s2.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dt2, t2.data(), t2.size() * 8, hipMemcpyHostToDevice);
  hipMemset(fix, 0, 256);
  ttnet::FullPwArgs a{};
  a.n = n; a.H = H; a.W = W; a.groups = G; a.cin = 30; a.mid = 240; a.cout = 30; a.Cout = 60; a.Csrc = C; a.interleaved = 0;
  a.src[0] = dx; a.w1 = dw1; a.w2 = dw2; a.s1 = ds1; a.t1 = dt1; a.s2 = ds2; a.t2 = dt2; a.out_rp = dout; a.out_float = nullptr;
  a.fix_count = fix; a.fix_list = fix + 64; a.range_flag = nullptr;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) ttnet::launch_full_pw(a, 0);
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) ttnet::launch_full_pw(a, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  uint32_t cnt[64];
  hipMemcpy(cnt, fix, 256, hipMemcpyDeviceToHost);
  printf("skip=%d n=%d: %.1f us per block (fast + float64 pass), listed %u + %u of %d pixels per group\n", TT_FULLPW_SKIP, n, 1e3 * ms / reps,
         cnt[0], cnt[1], n * H * W);
#ifdef TT_FULLPW_STAMP
  unsigned long long st[8];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(ttnet::g_pw_stamps), sizeof(st));
  const double runs = reps + 3;
  printf("wave 0 of block 0, us per launch: gather %.1f, layer 1 + GELU %.1f, layer 2 %.1f, epilogue %.1f, (loop head %.1f)\n", st[0] / 1e3 / runs, st[1] / 1e3 / runs,
         st[2] / 1e3 / runs, st[3] / 1e3 / runs, st[5] / 1e3 / runs);
#endif
  return 0;
}
