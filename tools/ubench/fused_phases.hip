// Additive decomposition of gate_block_kernel (gate_fused.hip): the same kernel built with parts
// switched off (TT_FUSED_SKIP), random tables and rows, timed with HIP events.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DTT_FUSED_SKIP=<mask> -o fused_phases_<mask> fused_phases.hip
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../scale_imagenet_amd/csrc/gate_fused.hip"

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
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  struct Geo { int C, H, Ho, off, last; };
  const Geo geos[3] = {{64, 56, 29, 1, 0}, {128, 29, 15, 0, 0}, {256, 15, 8, 0, 1}};
  for (const Geo &g : geos) {
    const size_t xin = (size_t)n * g.C * g.H * 8, yout = (size_t)n * 2 * g.C * g.Ho * 8, idxb = (size_t)n * (g.C / 8) * g.Ho * g.Ho * 4;
    void *x, *y, *c3, *dw, *cf;
    uint32_t *idx;
    hipMalloc(&x, xin); hipMalloc(&y, yout); hipMalloc((void **)&idx, idxb);
    hipMalloc(&c3, (size_t)(g.C / 8) * 65536); hipMalloc(&dw, (size_t)g.C * 16384); hipMalloc(&cf, (size_t)(g.C / 4) * 65536);
    std::vector<uint32_t> r(xin / 4);
    uint64_t s = 88172645463325252ull;
    auto fill = [&](void *dst, size_t bytes, uint32_t mask) {
      r.resize(bytes / 4);
      for (auto &v : r) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s & mask; }
      hipMemcpy(dst, r.data(), bytes, hipMemcpyHostToDevice);
    };
    fill(x, xin, g.H > 32 ? 0xFFFFFFFFu : (g.H > 16 ? 0x1FFFFFFFu : 0x7FFF7FFFu));
    fill(c3, (size_t)(g.C / 8) * 65536, ~0u); fill(dw, (size_t)g.C * 16384, ~0u); fill(cf, (size_t)(g.C / 4) * 65536, ~0u);
    ttnet::FusedBlockArgs f{};
    f.n = n; f.C = g.C; f.H = g.H; f.Ho = g.Ho; f.off34 = g.off; f.last = g.last;
    f.x = x; f.img_c3 = c3; f.img_dw = dw; f.t_cf = (const uint8_t *)cf; f.y = y; f.idx = g.last ? idx : nullptr;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) ttnet::launch_gate_block(f, 0);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) ttnet::launch_gate_block(f, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("skip=%d n=%d C=%d %dx%d: %.2f us per launch\n", TT_FUSED_SKIP, n, g.C, g.H, g.H, 1e3 * ms / reps);
    hipFree(x); hipFree(y); hipFree(idx); hipFree(c3); hipFree(dw); hipFree(cf);
  }
  return 0;
}
