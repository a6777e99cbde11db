// Truth-table builder: enumerates every input pattern of one Block_TT on the GPU.
//
// Replaces the reference's offline enumerator Block_TT.get_TT_block_all_filter
// (models/TT_FHE_SMALL.py:322-342) and, at inference, the float pipeline
// conv1 -> bn1 -> gelu -> conv2 -> bn2 -> (x >= 0)  of Block_TT.forward (:307-320):
// since both convolutions share `groups`, one output bit is a function of the
// n = cin_g*kh*kw input bits of its group, so the block IS this table.
//
// Arithmetic: float64 with an exact erf, from the float32 parameters.  That is the
// mathematically defined value of the reference's function; the reference's own float32
// evaluation (oneDNN) can differ from it only where |pre-activation| is below its rounding
// noise.  Entries with |pre| < 1e-5 are counted as near ties and reported.
//
// One thread per table entry; the group's weights live in LDS as doubles.
// Roofline: fp64 VALU (78 TFLOP/s); one-off work per checkpoint, not on the forward path.

#include "ttnet_common.h"

namespace ttnet {

namespace {

constexpr int kThreads = 256;

__device__ inline double gelu_exact(double x) { return 0.5 * x * (1.0 + erf(x * 0.70710678118654752440)); }

// LDS: w1 [mid_g][n] in internal bit order, w2t [mid_g][cout_g], s1,t1 [mid_g], s2,t2 [cout_g]
template <int MAXC>
__global__ __launch_bounds__(kThreads) void lut_build_kernel(LutBuildArgs a) {
  extern __shared__ __align__(16) double lds[];
  const int n = a.n, mid_g = a.mid_g, cout_g = a.cout_g;
  const int g = blockIdx.y;
  double *w1 = lds;
  double *w2t = w1 + mid_g * n;
  double *s1 = w2t + mid_g * cout_g;
  double *t1 = s1 + mid_g;
  for (int i = threadIdx.x; i < mid_g * n; i += kThreads) {
    int m = i / n, p = i % n;
    w1[i] = (double)a.w1[((size_t)g * mid_g + m) * n + a.perm[p]];
  }
  for (int i = threadIdx.x; i < mid_g * cout_g; i += kThreads) {
    int m = i / cout_g, o = i % cout_g;
    w2t[i] = (double)a.w2[((size_t)g * cout_g + o) * mid_g + m];
  }
  for (int i = threadIdx.x; i < mid_g; i += kThreads) {
    s1[i] = a.s1[(size_t)g * mid_g + i];
    t1[i] = a.t1[(size_t)g * mid_g + i];
  }
  __syncthreads();

  const unsigned idx = blockIdx.x * kThreads + threadIdx.x;  // grid.x * 256 == 2^n (or one block if 2^n < 256)
  const unsigned entries = 1u << n;
  const bool live = idx < entries;
  double acc[MAXC];
#pragma unroll
  for (int o = 0; o < MAXC; ++o) acc[o] = 0.0;
  if (live) {
    for (int m = 0; m < mid_g; ++m) {
      double s = 0.0;
      const double *wr = w1 + m * n;
      for (int p = 0; p < n; ++p) s += ((idx >> p) & 1u) ? wr[p] : 0.0;
      const double h = gelu_exact(s * s1[m] + t1[m]);
      const double *w2r = w2t + m * cout_g;
#pragma unroll
      for (int o = 0; o < MAXC; ++o)
        if (o < cout_g) acc[o] = fma(h, w2r[o], acc[o]);
    }
  }
  unsigned bits = 0, ties = 0;
  float outf[MAXC];
#pragma unroll
  for (int o = 0; o < MAXC; ++o) {
    outf[o] = 0.f;
    if (o < cout_g) {
      const double pre = acc[o] * a.s2[(size_t)g * cout_g + o] + a.t2[(size_t)g * cout_g + o];
      if (live && fabs(pre) < 1e-5) ++ties;
      bits |= (pre >= 0.0 ? 1u : 0u) << o;
      outf[o] = (float)(pre > 0.0 ? pre : 0.0);
    }
  }
  if (ties) atomicAdd(a.near_ties, ties);

  if (a.last) {
    if (live) {
      float *dst = (float *)a.table + ((size_t)g * entries + idx) * cout_g;
#pragma unroll
      for (int o = 0; o < MAXC; ++o)
        if (o < cout_g) dst[o] = outf[o];
    }
  } else if (cout_g == 1) {
    // 1-bit entries: dword w = idx>>5, bit idx&31; the dwords of 16 consecutive groups
    // (channels) are striped: [g/16][w][g%16], so that a workgroup's 16 tables interleave in
    // LDS banks (gate.hip)
    const unsigned long long m = __ballot(live && (bits & 1u));
    const int lane = threadIdx.x & 63;
    const size_t words = entries >= 32 ? entries / 32 : 1;
    if (lane == 0 && idx < entries) {
      unsigned *t32 = (unsigned *)a.table + ((size_t)(g >> 4) * words * 16 + (g & 15));
      t32[(size_t)(idx >> 5) * 16] = (unsigned)m;
      if (entries >= 64) t32[(size_t)((idx >> 5) + 1) * 16] = (unsigned)(m >> 32);
    }
  } else if (live) {
    if (cout_g <= 8)
      ((uint8_t *)a.table)[(size_t)g * entries + idx] = (uint8_t)bits;
    else
      ((uint16_t *)a.table)[(size_t)g * entries + idx] = (uint16_t)bits;
  }
}

}  // namespace

int launch_lut_build(const LutBuildArgs &a, hipStream_t s) {
  if (a.n < 1 || a.n > 20 || a.cout_g < 1 || a.cout_g > 16) {
    set_error("lut_build: unsupported geometry n=%d cout_g=%d", a.n, a.cout_g);
    return TTNET_E_UNSUPPORTED;
  }
  const unsigned entries = 1u << a.n;
  if (a.cout_g == 1 && a.groups % 16) {
    set_error("lut_build: depthwise tables are striped by 16 channels; groups=%d", a.groups);
    return TTNET_E_UNSUPPORTED;
  }
  dim3 grid((entries + kThreads - 1) / kThreads, a.groups);
  size_t lds = sizeof(double) * ((size_t)a.mid_g * a.n + (size_t)a.mid_g * a.cout_g + 2 * (size_t)a.mid_g);
  if (lds > (size_t)kMaxLds) {
    set_error("lut_build: group weights (%zu B) exceed LDS", lds);
    return TTNET_E_UNSUPPORTED;
  }
  auto k = lut_build_kernel<16>;
  TT_TRY(ensure_dynamic_lds((const void *)k, lds));
  hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, s, a);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
