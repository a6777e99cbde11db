// Standalone check + timing of the sign-only stem EXPERIMENT (stem_signonly_experiment.hip), no Python: the listed (fast) path against the EXACT chain
// bit for bit, both against a float64 evaluation on the host (first images), with the threshold scaled up so that
// lists overflow and several diagonal tiles run; then event timings of each variant.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o stem_check stem_check.hip
//   ./stem_check [images=256] [checked_images=4] [u8=0]
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "stem_signonly_experiment.hip"

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
int launch_rp_to_cp(const uint64_t *, uint16_t *, int, int, int, int, hipStream_t) { return -1; }     // (gate.hip; never reached: cp = nullptr)
int ensure_dynamic_lds(const void *kernel, size_t bytes) {
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -3;
}
}  // namespace ttnet

static uint64_t rng_state = 88172645463325252ull;
static inline uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 20); }

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  const int nchk = std::min(n, argc > 2 ? atoi(argv[2]) : 4);
  const size_t xe = (size_t)n * 3 * 224 * 224;
  const size_t rows = (size_t)n * 64 * 56;
  float *x; uint64_t *rp; uint16_t *wf; float *init; uint32_t *flag, *stats;
  hipMalloc(&x, xe * 4); hipMalloc(&rp, rows * 8); hipMalloc(&wf, ttnet::stem_split_weights_elems() * 2);
  hipMalloc(&init, 256); hipMalloc(&flag, 4); hipMalloc(&stats, 16);
  hipMemset(flag, 0, 4);
  // images: normalised uint8 noise with block structure (like the synthetic generator's)
  std::vector<float> h(xe);
  const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) {
      std::vector<uint8_t> coarse(14 * 14), mid(56 * 56);
      for (auto &v : coarse) v = rnd() & 255;
      for (auto &v : mid) v = rnd() & 255;
      for (int y = 0; y < 224; ++y)
        for (int xx = 0; xx < 224; ++xx) {
          const int u = ((int)(rnd() & 255) + mid[(y / 4) * 56 + xx / 4] + coarse[(y / 16) * 14 + xx / 16] + 128) >> 2;
          h[(((size_t)i * 3 + c) * 224 + y) * 224 + xx] = ((float)u / 255.0f - mean[c]) / sd[c];
        }
    }
  // a few forced near-zero / tiny values so that fp16 subnormals of the split occur
  for (int k = 0; k < 1000; ++k) h[rnd() % xe] = ((rnd() & 1) ? 1.f : -1.f) * ldexpf(1.f, -(int)(rnd() % 30));
  hipMemcpy(x, h.data(), xe * 4, hipMemcpyHostToDevice);
  std::vector<float> w(64 * 147);
  std::vector<double> sc(64), sh(64);
  for (auto &v : w) v = ((float)(rnd() & 0xFFFF) / 65536.f - 0.5f) * 0.29f;
  for (int c = 0; c < 64; ++c) { sc[c] = (0.5 + (rnd() & 1023) / 1024.0) / 0.45; sh[c] = ((int)(rnd() & 1023) - 512) / 2048.0; }
  sc[7] = 1e-3; sh[9] = 37.5; sc[11] = 55.0;          // odd channels: tiny filter, large shift, large filter
  std::vector<uint16_t> wfh(ttnet::stem_split_weights_elems());
  std::vector<float> inith(64);
  if (!ttnet::stem_split_weights(w.data(), sc.data(), sh.data(), 64, wfh.data(), inith.data())) { printf("split failed\n"); return 1; }
  printf("shift constant c = %g, thr_abs = %g\n", inith[0], inith[1]);
  hipMemcpy(wf, wfh.data(), wfh.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(init, inith.data(), 256, hipMemcpyHostToDevice);

  auto run = [&](const char *exact, const char *tau, std::vector<uint64_t> &out, uint32_t st[4]) {
    if (exact) setenv("TTNET_STEM_EXACT", exact, 1); else unsetenv("TTNET_STEM_EXACT");
    if (tau) setenv("TTNET_STEM_TAU_SCALE", tau, 1); else unsetenv("TTNET_STEM_TAU_SCALE");
    hipMemset(rp, 0xAB, rows * 8);
    hipMemset(stats, 0, 16);
    const int r = ttnet::launch_stem(x, false, nullptr, wf, init, rp, nullptr, n, 64, flag, stats, 0);
    if (r != 0 || hipDeviceSynchronize() != hipSuccess) { printf("launch failed %d %s\n", r, hipGetErrorString(hipGetLastError())); exit(1); }
    out.resize(rows);
    hipMemcpy(out.data(), rp, rows * 8, hipMemcpyDeviceToHost);
    hipMemcpy(st, stats, 16, hipMemcpyDeviceToHost);
  };
  std::vector<uint64_t> ex, fa, f3, f30, fbig;
  uint32_t st[4];
  run("1", nullptr, ex, st);
  auto cmp = [&](const char *name, const std::vector<uint64_t> &a) {
    size_t bad = 0;
    for (size_t i = 0; i < rows; ++i) bad += __builtin_popcountll(a[i] ^ ex[i]);
    printf("%-14s listed %u (%.3f %%), diagonal tiles %u, list flushes %u: %zu bits differ from EXACT\n", name, st[0],
           100.0 * st[0] / ((double)n * 64 * 3136), st[1], st[2], bad);
    return bad;
  };
  size_t bad = 0;
  run(nullptr, nullptr, fa, st); bad += cmp("fast", fa);
  run(nullptr, "3", f3, st); bad += cmp("fast tau x3", f3);
  run(nullptr, "30", f30, st); bad += cmp("fast tau x30", f30);
  run(nullptr, "1e9", fbig, st); bad += cmp("fast tau x1e9", fbig);
  // float64 reference of the first images
  size_t wrong = 0, ties = 0;
  double minabs = 1e30;
  for (int i = 0; i < nchk; ++i) {
    std::vector<float> pool((size_t)3 * 112 * 112);
    for (int c = 0; c < 3; ++c)
      for (int y = 0; y < 112; ++y)
        for (int xx = 0; xx < 112; ++xx) {
          const float *p = &h[(((size_t)i * 3 + c) * 224 + 2 * y) * 224 + 2 * xx];
          pool[((size_t)c * 112 + y) * 112 + xx] = (((p[0] + p[1]) + p[224]) + p[225]) * 0.25f;
        }
    for (int ch = 0; ch < 64; ++ch)
      for (int oy = 0; oy < 56; ++oy) {
        const uint64_t word = ex[((size_t)i * 64 + ch) * 56 + oy];
        for (int ox = 0; ox < 56; ++ox) {
          double acc = 0.0;
          for (int c = 0; c < 3; ++c)
            for (int kh = 0; kh < 7; ++kh) {
              const int iy = 2 * oy - 3 + kh;
              if (iy < 0 || iy >= 112) continue;
              for (int kw = 0; kw < 7; ++kw) {
                const int ix = 2 * ox - 3 + kw;
                if (ix < 0 || ix >= 112) continue;
                acc += (double)w[(size_t)ch * 147 + (c * 7 + kh) * 7 + kw] * (double)pool[((size_t)c * 112 + iy) * 112 + ix];
              }
            }
          const double pre = acc * sc[ch] + sh[ch];
          const int bit = (word >> ox) & 1;
          if (std::fabs(pre) < 1e-5 * std::max(1.0, sc[ch])) { ties++; continue; }
          minabs = std::min(minabs, std::fabs(pre));
          if (bit != (pre >= 0)) { if (wrong < 10) printf("  wrong: img %d ch %d (%d,%d) pre %.3e bit %d\n", i, ch, oy, ox, pre, bit); wrong++; }
        }
      }
  }
  size_t model_listed = 0, model_total = 0;
  for (int mi = 0; mi < std::min(n, 8); ++mi) {  // host model of the listing: first pass on the fp16 head terms, threshold = |x1 window| + thr_abs
    std::vector<float> x1((size_t)3 * 112 * 112);
    for (int c = 0; c < 3; ++c)
      for (int y = 0; y < 112; ++y)
        for (int xx = 0; xx < 112; ++xx) {
          const float *pz = &h[(((size_t)mi * 3 + c) * 224 + 2 * y) * 224 + 2 * xx];
          const float v = (((pz[0] + pz[1]) + pz[224]) + pz[225]) * (0.25f * 16.0f);
          x1[((size_t)c * 112 + y) * 112 + xx] = (float)(_Float16)v;
        }
    const uint16_t *img = wfh.data();
    size_t listed = 0, total = 0;
    for (int ch = 0; ch < 64; ++ch)
      for (int oy = 0; oy < 56; ++oy)
        for (int ox = 0; ox < 56; ++ox) {
          double acc = 0.0, n2 = 0.0;
          for (int c = 0; c < 3; ++c)
            for (int kh = 0; kh < 7; ++kh) {
              const int iy = 2 * oy - 3 + kh;
              for (int slot = 0; slot < 8; ++slot) {
                const int ix = 2 * ox - 4 + slot;
                const double xv = (iy < 0 || iy >= 112 || ix < 0 || ix >= 112) ? 0.0 : (double)x1[((size_t)c * 112 + iy) * 112 + ix];
                const uint16_t wb = img[(size_t)ch * 184 + (c * 7 + kh) * 8 + slot];
                _Float16 wh; memcpy(&wh, &wb, 2);
                acc += (double)(float)wh * xv;
                n2 += xv * xv;
              }
            }
          double shv = 0.0;
          for (int slot = 0; slot < 3; ++slot) { const uint16_t wb = img[(size_t)ch * 184 + 21 * 8 + slot]; _Float16 wh; memcpy(&wh, &wb, 2); shv += (double)(float)wh * inith[0]; }
          acc += shv;
          total++;
          if (std::fabs(acc) < std::sqrt(n2) + inith[1]) listed++;
        }
    model_listed += listed; model_total += total;
  }
  printf("host model, first %d images: %.3f %% of the outputs inside the threshold\n", std::min(n, 8), 100.0 * model_listed / model_total);
  printf("EXACT vs float64 on %d images: %zu wrong bits, %zu near ties skipped, smallest |pre| checked %.2e\n", nchk, wrong, ties, minabs);
  // timings
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time_it = [&](const char *exact, const char *tau, const char *name) {
    if (exact) setenv("TTNET_STEM_EXACT", exact, 1); else unsetenv("TTNET_STEM_EXACT");
    if (tau) setenv("TTNET_STEM_TAU_SCALE", tau, 1); else unsetenv("TTNET_STEM_TAU_SCALE");
    for (int i = 0; i < 5; ++i) ttnet::launch_stem(x, false, nullptr, wf, init, rp, nullptr, n, 64, flag, nullptr, 0);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) ttnet::launch_stem(x, false, nullptr, wf, init, rp, nullptr, n, 64, flag, nullptr, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s n=%d: %.2f us per launch\n", name, n, 1e3 * ms / reps);
  };
  time_it(nullptr, nullptr, "fast");
  time_it("1", nullptr, "exact");
  time_it(nullptr, "3", "fast tau x3");
  time_it(nullptr, nullptr, "fast");
  uint32_t fl = 0;
  hipMemcpy(&fl, flag, 4, hipMemcpyDeviceToHost);
  printf("range flag %u; %s\n", fl, (bad == 0 && wrong == 0) ? "OK" : "FAILED");
  return (bad == 0 && wrong == 0) ? 0 : 1;
}
