// Gate path of the full variant (fan-in 30): TT_vf_19lv3_imgnet,
// models/TT_general_imagenet_v2.py:21-136 -- (6,5)/(5,6) depthwise windows, 30-channel
// grouped 1x1 blocks.  A flat truth table would need 2^30 entries per output bit, so the
// blocks are evaluated directly:  pad -> conv1 -> bn1 -> gelu -> conv2 -> bn2 -> (x >= 0)
// (Block_TT.forward, models/TT_FHE_SMALL.py:307-320) on bit inputs, in float64 with an exact
// erf -- the same arithmetic that defines the truth tables of the other variants
// (lut_build.hip), so "the bit" means the same thing for every variant and is independent of
// summation order (float32 re-ordering was measured to flip outputs).
//
// Everything stays on row-packed planes (uint64 per image row).  Bound: the float64 matrix pipe
// (0.72 GFLOP per image in the two grouped convolutions of the 1x1 blocks: 4.7 ms per 512 images at
// the 78.6 TFLOP/s FP64 peak) plus the GELU (a branch-free table erf, see gelu_exact).

#include <cstdlib>

#include "ttnet_common.h"

namespace ttnet {

namespace {

// erf in float64 without branches: libm's erf picks one of five range-dependent algorithms per lane, so
// a wave executes all of them (108 vector instructions per call, and the full model calls it 7.6 M times
// per image: it was most of the variant's time).  Here: 48 intervals of width 1/8 on [0, 6), the degree-8
// Chebyshev-node interpolant per interval (tools/gen_erf_table.py, mpmath at 60 digits; max abs error
// 3.3e-16 against mpmath's erf, i.e. as exact as libm's), evaluated with 8 FMAs from a 3.4 KiB LDS copy
// of the table.  erf(x) rounds to 1.0 for x >= 6.
constexpr int kErfN = 48, kErfC = 9;
__device__ const double kErfTable[kErfN][kErfC] = {
#include "erf_table.inc"
};
__device__ inline void erf_table_to_lds(double *dst) {
  for (int i = threadIdx.x; i < kErfN * kErfC; i += blockDim.x) dst[i] = (&kErfTable[0][0])[i];
}
__device__ inline double gelu_exact(double z, const double *tab) {
  const double x = z * 0.70710678118654752440, ax = __builtin_fabs(x);
  int i = (int)(ax * 8.0);
  i = i < kErfN - 1 ? i : kErfN - 1;
  const double t = ax - ((double)i + 0.5) * 0.125;
  const double *c = tab + i * kErfC;
  double p = c[8];
#pragma unroll
  for (int k = 7; k >= 0; --k) p = fma(p, t, c[k]);
  p = ax >= 6.0 ? 1.0 : p;
  return 0.5 * z * (1.0 + __builtin_copysign(p, x));
}

// ---- depthwise Block_TT: one input channel -> 8 mid -> 1 output ---------------------------------
// grid (C, n-chunks); thread = (image, output row); weights of the channel in LDS.
__global__ __launch_bounds__(256) void full_dw_kernel(FullDwArgs a) {
  __shared__ double w1[8 * 36], s1[8], t1[8], w2[8], s2, t2;
  __shared__ double erf_tab[kErfN * kErfC];
  erf_table_to_lds(erf_tab);
  const int c = blockIdx.x, nk = a.kh * a.kw;
  for (int i = threadIdx.x; i < 8 * nk; i += blockDim.x) w1[i] = (double)a.w1[(size_t)c * 8 * nk + i];
  if (threadIdx.x < 8) {
    s1[threadIdx.x] = a.s1[c * 8 + threadIdx.x];
    t1[threadIdx.x] = a.t1[c * 8 + threadIdx.x];
    w2[threadIdx.x] = (double)a.w2[c * 8 + threadIdx.x];
  }
  if (threadIdx.x == 0) {
    s2 = a.s2[c];
    t2 = a.t2[c];
  }
  __syncthreads();
  const int rows = a.n * a.ho;
  for (int t = blockIdx.y * blockDim.x + threadIdx.x; t < rows; t += gridDim.y * blockDim.x) {
    const int n = t / a.ho, oy = t % a.ho;
    uint64_t r[6];
    for (int kh = 0; kh < a.kh; ++kh) {
      const int iy = oy * a.stride - a.pad + kh;
      r[kh] = (iy >= 0 && iy < a.H) ? a.x_rp[((size_t)n * a.C + c) * a.H + iy] << a.pad : 0ull;   // bit 0 = column -pad
    }
    uint64_t out = 0;
    for (int ox = 0; ox < a.wo; ++ox) {
      double acc = 0.0;
      for (int m = 0; m < 8; ++m) {
        double s = 0.0;
        for (int kh = 0; kh < a.kh; ++kh) {
          const uint32_t bits = (uint32_t)(r[kh] >> (ox * a.stride));
          for (int kw = 0; kw < a.kw; ++kw) s += ((bits >> kw) & 1u) ? w1[m * nk + kh * a.kw + kw] : 0.0;
        }
        acc = fma(gelu_exact(s * s1[m] + t1[m], erf_tab), w2[m], acc);
      }
      const double pre = acc * s2 + t2;
      out |= (uint64_t)(pre >= 0.0) << (ox + a.pad_l);
    }
    a.out[((size_t)n * a.C + c) * a.Ho + oy + a.pad_t] = out;
  }
}

// The same block with per-row partial sums from tables: the pre-activation of mid unit m is a sum
// over the window rows of (sum over kw of w[m][kh][kw] x[kh][kw]), and a row has only KW <= 6 bits, so
// that inner sum is a 2^KW-entry float64 table per (m, kh) (at most 20 KiB per channel, built in
// LDS by the workgroup): KH lookups and adds per mid unit instead of KH*KW conditional adds.  (The
// taps of a row are summed kw-ascending, the rows kh-ascending: another association than the
// oracle's single running sum, a 1e-16 relative effect.)
template <int KH, int KW>
__global__ __launch_bounds__(256) void full_dw_tab_kernel(FullDwArgs a) {
  __shared__ double tab[8][KH][1 << KW];
  __shared__ double s1[8], t1[8], w2[8], s2, t2;
  __shared__ double erf_tab[kErfN * kErfC];
  erf_table_to_lds(erf_tab);
  const int c = blockIdx.x;
  constexpr int nk = KH * KW;
  for (int i = threadIdx.x; i < 8 * KH * (1 << KW); i += blockDim.x) {
    const int bits = i & ((1 << KW) - 1), kh = (i >> KW) % KH, m = i / (KH << KW);
    double sum = 0.0;
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) sum += ((bits >> kw) & 1) ? (double)a.w1[(size_t)c * 8 * nk + m * nk + kh * KW + kw] : 0.0;
    tab[m][kh][bits] = sum;
  }
  if (threadIdx.x < 8) {
    s1[threadIdx.x] = a.s1[c * 8 + threadIdx.x];
    t1[threadIdx.x] = a.t1[c * 8 + threadIdx.x];
    w2[threadIdx.x] = (double)a.w2[c * 8 + threadIdx.x];
  }
  if (threadIdx.x == 0) {
    s2 = a.s2[c];
    t2 = a.t2[c];
  }
  __syncthreads();
  const int rows = a.n * a.ho;
  for (int t = blockIdx.y * blockDim.x + threadIdx.x; t < rows; t += gridDim.y * blockDim.x) {
    const int n = t / a.ho, oy = t % a.ho;
    uint64_t r[KH];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      const int iy = oy * a.stride - a.pad + kh;
      r[kh] = (iy >= 0 && iy < a.H) ? a.x_rp[((size_t)n * a.C + c) * a.H + iy] << a.pad : 0ull;   // bit 0 = column -pad
    }
    uint64_t out = 0;
    for (int ox = 0; ox < a.wo; ++ox) {
      uint32_t idx[KH];
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) idx[kh] = (uint32_t)(r[kh] >> (ox * a.stride)) & ((1u << KW) - 1u);
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        double sm = tab[m][0][idx[0]];
#pragma unroll
        for (int kh = 1; kh < KH; ++kh) sm += tab[m][kh][idx[kh]];
        acc = fma(gelu_exact(sm * s1[m] + t1[m], erf_tab), w2[m], acc);
      }
      const double pre = acc * s2 + t2;
      out |= (uint64_t)(pre >= 0.0) << (ox + a.pad_l);
    }
    a.out[((size_t)n * a.C + c) * a.Ho + oy + a.pad_t] = out;
  }
}

// ---- grouped 1x1 Block_TT with `cin` inputs per group ------------------------------------------
// Input bit (group g, j): channel J = cin*g + j of either a plane tensor (conv3) or of the
// interleaved concat of four branch tensors (convf: channel J -> branch J%4, channel J/4;
// models/TT_general_imagenet_v2.py:131-135).  One wave = one image row; lanes = columns.
// Output: bits (ballot -> row words) or, for the last block, relu'd float32.
// Generic version (any group shape that fits; the full model's 30 -> 240 -> 30 / 15 groups take the
// matrix-instruction kernel below).  1024 threads share one copy of the group's weights in LDS (119 KiB for 30 -> 240 -> 30): sixteen
// waves per CU hide the latency of the broadcast LDS reads that one wave per SIMD exposed (3x).
// (Compile-time sizes with full unrolling were tried and were slower: more registers, same reads.)
__global__ __launch_bounds__(1024) void full_pw_kernel(FullPwArgs a) {
  extern __shared__ __align__(16) double lds[];
  __shared__ double erf_tab[kErfN * kErfC];
  erf_table_to_lds(erf_tab);
  const int g = blockIdx.x, cin = a.cin, mid = a.mid, cout = a.cout;
  double *w1 = lds;                      // [mid][cin]
  double *w2 = w1 + mid * cin;           // [cout][mid]
  double *s1 = w2 + cout * mid, *t1 = s1 + mid, *s2 = t1 + mid, *t2 = s2 + cout;
  for (int i = threadIdx.x; i < mid * cin; i += blockDim.x) w1[i] = (double)a.w1[(size_t)g * mid * cin + i];
  for (int i = threadIdx.x; i < cout * mid; i += blockDim.x) w2[i] = (double)a.w2[(size_t)g * cout * mid + i];
  for (int i = threadIdx.x; i < mid; i += blockDim.x) {
    s1[i] = a.s1[g * mid + i];
    t1[i] = a.t1[g * mid + i];
  }
  for (int i = threadIdx.x; i < cout; i += blockDim.x) {
    s2[i] = a.s2[g * cout + i];
    t2[i] = a.t2[g * cout + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  // A wave covers as many whole image rows as fit its 64 lanes (one row of 56 pixels, two of 29,
  // four of 16, seven of 9): lane = (row r of the bundle, column x).  (One row per wave left 55-85 % of
  // the lanes idle in the later blocks.)
  const int rpw = 64 / a.W, bundles = (a.H + rpw - 1) / rpw;
  const int r = lane / a.W, x = lane - r * a.W;
  const int tasks = a.n * bundles;
  for (int t = blockIdx.y * nwaves + wave; t < tasks; t += gridDim.y * nwaves) {
    const int n = t / bundles, y = (t % bundles) * rpw + r;
    const bool live = r < rpw && y < a.H;
    // this pixel's `cin` input bits
    uint32_t in = 0;
    if (live)
      for (int j = 0; j < cin; ++j) {
        const int J = cin * g + j;
        uint64_t row;
        if (a.interleaved) row = a.src[J & 3][((size_t)n * a.Csrc + (J >> 2)) * a.H + y];
        else row = a.src[0][((size_t)n * a.Csrc + J) * a.H + y];
        in |= (uint32_t)((row >> x) & 1ull) << j;
      }
    double acc[30];
#pragma unroll
    for (int o = 0; o < 30; ++o) acc[o] = 0.0;
    for (int m = 0; m < mid; ++m) {
      double s = 0.0;
      const double *wr = w1 + m * cin;
      for (int j = 0; j < cin; ++j) s += ((in >> j) & 1u) ? wr[j] : 0.0;
      const double h = gelu_exact(s * s1[m] + t1[m], erf_tab);
#pragma unroll
      for (int o = 0; o < 30; ++o)
        if (o < cout) acc[o] = fma(h, w2[o * mid + m], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < 30; ++o) {
      if (o >= cout) break;
      const double pre = acc[o] * s2[o] + t2[o];
      if (a.out_float) {
        if (live) a.out_float[(((size_t)n * a.Cout + g * cout + o) * a.H + y) * a.W + x] = (float)(pre > 0.0 ? pre : 0.0);
      } else {
        const uint64_t m64 = __ballot(live && pre >= 0.0);
        if (live && x == 0)                               // the first lane of every row writes its row word
          a.out_rp[((size_t)n * a.Cout + g * cout + o) * a.H + y] = (m64 >> (r * a.W)) & ((1ull << a.W) - 1ull);
      }
    }
  }
}

// The same block on the float64 matrix instruction v_mfma_f64_16x16x4_f64 (operand layouts probed in
// tools/ubench/mfma_f64_layout.hip: A lane l = A[l%16][l/16], B lane l = B[l/16][l%16], D register i of
// lane l = D[4i + l/16][l%16]).  Both layers are small GEMMs over the 64 pixels of a task:
//   layer 1   H[240 x 64] = W1[240 x 30] * X[30 x 64]      (X = the input bits as 0.0 / 1.0)
//   layer 2   O[cout x 64] = W2[cout x 240] * gelu(bn1(H))
// walked in 15 tiles of 16 hidden units: the D registers of a layer-1 tile are, after BN + GELU,
// directly the B operands of four layer-2 k-steps (register i = hidden rows 4i .. 4i+3 of the tile).
// Weights sit in LDS in fragment order and are read once per tile and wave (240 LDS reads per task
// instead of 14,400 broadcast reads); no operand is broadcast lane by lane.  Sums are formed in a
// different order than the oracle's (blocks of four products): a 1e-16 relative effect, i.e. only
// an exact tie could turn.
typedef double f64x4 __attribute__((ext_vector_type(4)));

// FIX: the tasks are not image rows but the pixels full_pw_fast_kernel listed (64 per wave, any images, any
// positions); their bits are patched into the row words with atomics.
template <int OT, bool FIX>      // 16-row output tiles: 2 (cout = 30) or 1 (cout = 15)
__global__ __launch_bounds__(512) void full_pw_mfma_kernel(FullPwArgs a) {
  extern __shared__ __align__(16) double lds[];
  __shared__ double erf_tab[kErfN * kErfC];
  erf_table_to_lds(erf_tab);
  constexpr int CIN = 30, MT = 15, KS1 = 8;              // 240 hidden units, K = 30 padded to 32
  constexpr int NTT = FIX ? 1 : 4;                       // pixel tiles of 16 per task (FIX: one, so that a short list still spreads over the chip)
  if constexpr (FIX) {                                   // nothing listed for this workgroup: skip the weight staging too
    const uint32_t capq = (uint32_t)a.n * (uint32_t)a.H * (uint32_t)a.W, cnt = min(a.fix_count[blockIdx.x], capq);
    if (blockIdx.y * (blockDim.x >> 6) >= (cnt + 15u) / 16u) return;
  }
  double *w1f = lds;                                     // [MT][KS1][64]
  double *w2f = w1f + MT * KS1 * 64;                     // [MT][4][OT][64]
  double *s1 = w2f + MT * 4 * OT * 64, *t1 = s1 + 16 * MT;
  const int g = blockIdx.x, cout = a.cout, mid = 16 * MT;
  for (int i = threadIdx.x; i < MT * KS1 * 64; i += blockDim.x) {
    const int l = i & 63, ks = (i >> 6) % KS1, mt = i / (64 * KS1);
    const int m = 16 * mt + (l & 15), k = 4 * ks + (l >> 4);
    w1f[i] = k < CIN ? (double)a.w1[((size_t)g * mid + m) * CIN + k] : 0.0;
  }
  for (int i = threadIdx.x; i < MT * 4 * OT * 64; i += blockDim.x) {
    const int l = i & 63, ot = (i >> 6) % OT, ii = (i / (64 * OT)) & 3, mt = i / (64 * OT * 4);
    const int o = 16 * ot + (l & 15), hid = 16 * mt + 4 * ii + (l >> 4);
    w2f[i] = o < cout ? (double)a.w2[((size_t)g * cout + o) * mid + hid] : 0.0;
  }
  for (int i = threadIdx.x; i < mid; i += blockDim.x) {
    s1[i] = a.s1[g * mid + i];
    t1[i] = a.t1[g * mid + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  const int lg = lane >> 4, ln = lane & 15;
  const int rpw = 64 / a.W, bundles = (a.H + rpw - 1) / rpw;
  const uint32_t cap = (uint32_t)a.n * (uint32_t)a.H * (uint32_t)a.W;
  const uint32_t listed = FIX ? min(a.fix_count[g], cap) : 0u;
  const int tasks = FIX ? (int)((listed + 15u) / 16u) : a.n * bundles;
  for (int t = blockIdx.y * nwaves + wave; t < tasks; t += gridDim.y * nwaves) {
    int n, y0, y, x, r;
    bool live;
    uint32_t pid = 0;
    if constexpr (FIX) {
      live = (uint32_t)(16 * t + ln) < listed;           // lane l and its three lane-group twins: listed pixel 16 t + l%16
      pid = live ? a.fix_list[(size_t)g * cap + 16 * t + ln] : 0u;
      x = (int)(pid % (uint32_t)a.W);
      y = (int)((pid / (uint32_t)a.W) % (uint32_t)a.H);
      n = (int)(pid / (uint32_t)(a.W * a.H));
      y0 = y;
      r = 0;
    } else {
      r = lane / a.W;
      x = lane - r * a.W;
      n = t / bundles;
      y0 = (t % bundles) * rpw;
      y = y0 + r;
      live = r < rpw && y < a.H;
    }
    uint32_t in = 0;                                     // this lane's pixel: its 30 input bits
    if (live)
#pragma unroll
      for (int j = 0; j < CIN; ++j) {
        const int J = CIN * g + j;
        const uint64_t row = a.interleaved ? a.src[J & 3][((size_t)n * a.Csrc + (J >> 2)) * a.H + y]
                                           : a.src[0][((size_t)n * a.Csrc + J) * a.H + y];
        in |= (uint32_t)((row >> x) & 1ull) << j;
      }
    // B fragments of layer 1: lane l of (k-step ks, pixel tile nt) = bit 4ks + l/16 of pixel 16nt + l%16
    double xf[KS1][NTT];
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
      const uint32_t inp = (FIX ? in : (uint32_t)__shfl((int)in, 16 * nt + ln)) >> lg;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) xf[ks][nt] = ((inp >> (4 * ks)) & 1u) ? 1.0 : 0.0;
    }
    f64x4 acc[OT][NTT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) acc[ot][nt] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int mt = 0; mt < MT; ++mt) {
      f64x4 d[NTT];
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) d[nt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const double wa = w1f[(mt * KS1 + ks) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) d[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, xf[ks][nt], d[nt], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {                       // register i = hidden unit 16mt + 4i + l/16
        const double sc = s1[16 * mt + 4 * i + lg], sh = t1[16 * mt + 4 * i + lg];
        double h[NTT];
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) h[nt] = gelu_exact(d[nt][i] * sc + sh, erf_tab);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const double wb = w2f[((mt * 4 + i) * OT + ot) * 64 + lane];
#pragma unroll
          for (int nt = 0; nt < NTT; ++nt) acc[ot][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wb, h[nt], acc[ot][nt], 0, 0, 0);
        }
      }
    }
    // acc[ot][nt][i] = output channel 16ot + 4i + l/16 at pixel 16nt + l%16
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = 16 * ot + 4 * i + lg;               // this lane's channel of the quartet
        const bool o_ok = o < cout;
        const double sc = o_ok ? a.s2[g * cout + o] : 0.0, sh = o_ok ? a.t2[g * cout + o] : -1.0;
        if constexpr (FIX) {
          // this lane's value belongs to its own listed pixel
          if (o_ok && live) {
            unsigned long long *word = (unsigned long long *)(a.out_rp + ((size_t)n * a.Cout + g * cout + o) * a.H + y);
            if (acc[ot][0][i] * sc + sh >= 0.0) atomicOr(word, 1ull << x);
            else atomicAnd(word, ~(1ull << x));
          }
        } else if (a.out_float) {
#pragma unroll
          for (int nt = 0; nt < NTT; ++nt) {
            const int p = 16 * nt + ln, pr = p / a.W, px = p - pr * a.W, py = y0 + pr;
            const double pre = acc[ot][nt][i] * sc + sh;
            if (o_ok && pr < rpw && py < a.H)
              a.out_float[(((size_t)n * a.Cout + g * cout + o) * a.H + py) * a.W + px] = (float)(pre > 0.0 ? pre : 0.0);
          }
        } else {
          // one ballot per pixel tile: bits 16q .. 16q+15 = channel quartet member q over the tile's 16 pixels
          uint64_t bal[4] = {0, 0, 0, 0};
#pragma unroll
          for (int nt = 0; nt < NTT; ++nt) bal[nt] = __ballot(acc[ot][nt][i] * sc + sh >= 0.0);
          // lane (q = l/16, row r' = l%16 < rpw) writes the row word of channel 16ot + 4i + q, image row y0 + r'
          const uint64_t mine = ((bal[0] >> (16 * lg)) & 0xFFFFull) | (((bal[1] >> (16 * lg)) & 0xFFFFull) << 16) |
                                (((bal[2] >> (16 * lg)) & 0xFFFFull) << 32) | (((bal[3] >> (16 * lg)) & 0xFFFFull) << 48);
          if (o_ok && ln < rpw && y0 + ln < a.H)
            a.out_rp[((size_t)n * a.Cout + g * cout + o) * a.H + y0 + ln] = (mine >> (ln * a.W)) & ((1ull << a.W) - 1ull);
        }
      }
  }
}

// ---- the same block on the 16-bit matrix cores, float64 only where the sign is in doubt --------------------------
// 0.72 GFLOP per image in float64 is what bounds the variant (the kernel above: 61 k matrix-pipe cycles per
// 64-pixel task).  The bit a block emits is sign(pre); it needs float64 only where |pre| is smaller than the
// error of a cheaper evaluation.  So: evaluate every output in split-fp16 arithmetic on v_mfma_f32_16x16x32_f16
// (the scheme of stem.hip: operands prescaled by a power of two and split into two fp16 terms, exact products,
// float32 accumulation; the inputs of layer 1 are bits, exact in fp16, so it needs two products, layer 2 three)
// with a float32 GELU, compare |pre| with a bound tau on that evaluation's error, and put the (pixel, group)
// pairs that fail on a list; full_pw_mfma_kernel<OT, true> then re-evaluates the listed pixels in float64 and
// patches their bits.  Every emitted bit is therefore the float64 bit -- the path is bit-identical to the
// float64 one -- provided tau really bounds the error.  tau (per output channel, computed by the kernel from
// the weights it stages): with A_m = sum_c |w1[m][c]|, zmax_m = |s1_m| A_m + |t1_m| (no hidden unit can exceed it),
//     ez_m = |s1_m| A_m (2^-21 + 16 x 2^-24) + 3 x 2^-24 zmax_m       layer 1: operand split, accumulation, BatchNorm fma
//     eg_m = 1.13 ez_m + 1.6e-6 + 2e-7 zmax_m                          GELU: its slope, its own error (below: gelu_lin_node)
//     E    = sum_m |w2[o][m]| eg_m + 3.2e-6 sum_m |w2[o][m]| |g_m|     layer 2: both operand splits, the dropped
//                                                                      low x low product, 32 roundings of the sum
//     tau  = 2 (|s2_o| E + 2^-22 |t2_o|)                               factor 2: margin
// The second sum of E is the pixel's own: one more matrix instruction per k-step on |w2| x g (the high halves, which
// under-state either factor by at most 2^-11: the 3.2 for 3.1; g signed, since |g| <= g + 0.34 -- gelu >= -0.17 --
// and 0.35 sum |w2| goes into the constant part) accumulates it beside the outputs -- with the worst case
// |g_m| <= zmax_m in its place the list was twice as long.
// On the synthetic model about 1 in 1000 (pixel, group) pairs is listed.  The last block emits relu'd float32
// features for a float32 head (tolerance 1e-5 on the logits): it takes the fast evaluation as it stands.
//
// gelu(z) = z Phi(z) with Phi from a table in LDS: 512 nodes z_i = i / 32 on [-8, 8), at each the value, the
// slope and half the curvature of Phi (float64 -> float32), evaluated as a quadratic around the nearest node:
// |dz| <= 1/64, so the neglected cubic term is <= (1/64)^3 / 6 x max |third derivative of Phi| (0.4) = 2.5e-7,
// and with the float32 roundings |gelu_f32(z) - gelu(z)| <= 4e-7 (|z| + 0.1) (tests/test_full_fast_bounds.py
// checks the formula in float32 against float64 on a dense grid).  Nine vector instructions and one 16-byte LDS
// read; the Abramowitz-Stegun form it replaced (a reciprocal, an exponential, five fmas) cost twice that and set
// the kernel's time.  Beyond the table Phi is 0 or 1 to 1e-15: the edge nodes hold exactly that, with no slope.
constexpr int kPhiN = 512;
// (slope and curvature are stored for an offset measured in the caller's units: z x scale)
__device__ inline void phi_table_to_lds(float4 *dst, const double *erf_tab, double scale) {
  for (int i = threadIdx.x; i < kPhiN; i += blockDim.x) {
    const double z = (double)(i - kPhiN / 2) * (1.0 / 32.0);
    const double x = z * 0.70710678118654752440, ax = __builtin_fabs(x);
    int k = (int)(ax * 8.0);
    k = k < kErfN - 1 ? k : kErfN - 1;
    const double t = ax - ((double)k + 0.5) * 0.125;
    const double *c = erf_tab + k * kErfC;
    double p = c[8];
    for (int q = 7; q >= 0; --q) p = fma(p, t, c[q]);
    p = ax >= 6.0 ? 1.0 : p;
    const double Phi = 0.5 * (1.0 + __builtin_copysign(p, x));
    const double pdf = 0.39894228040143267794 * exp(-0.5 * z * z);
    const bool edge = i == 0 || i == kPhiN - 1;
    dst[i] = edge ? make_float4(i ? 1.f : 0.f, 0.f, 0.f, 0.f)
                  : make_float4((float)Phi, (float)(pdf / scale), (float)(-0.5 * z * pdf / (scale * scale)), 0.f);
  }
}
// z x SCALE in, gelu(z) x SCALE out (SCALE = ACT_PRESCALE: the prescale of layer 2's operand rides along for free).
// In two steps so that a batch of table reads can be in flight before the first is needed.
template <int SCALE>
__device__ inline int gelu_node(float zs, float &dz) {
  float r = __builtin_rintf(zs * (32.0f / (float)SCALE));               // nearest node, in units of 1/32
  r = __builtin_fminf(__builtin_fmaxf(r, -(float)(kPhiN / 2)), (float)(kPhiN / 2 - 1));
  dz = fmaf(r, -(float)SCALE / 32.0f, zs);                             // offset from the node, in units of 1 / SCALE
  return (int)r + kPhiN / 2;
}
__device__ inline float gelu_eval(float zs, float dz, const float4 &c) { return zs * fmaf(dz, fmaf(dz, c.z, c.y), c.x); }
// full_pw_fast_kernel's GELU (round 3): a table of gelu ITSELF, kGelN = 4096 nodes z_i = (i - 2048) / 256 on [-8, 8), evaluated
// as the tangent at the NEAREST node.  |u - i| <= 1/2 node = 1/512 in z, so the neglected term is <= (1/512)^2 / 2 x
// max |gelu''| (= 2 pdf(0) = 0.798) = 1.53e-6; an entry holds the tangent as a line in u itself, (intercept, slope) with
// intercept = SCALE gelu(z_i) - i x slope (from the ROUNDED slope, so that the slope's rounding is not multiplied by u), and
// g = fma(u, slope, intercept) has the roundings of the intercept and of the fma: 2^-24 (|intercept| + |g|) <= 2e-7 |z|.
//   |gelu_lin(z) - gelu(z)| <= 1.6e-6 + 2e-7 |z|      (tests/test_full_fast_bounds.py restates it in float32 on a dense grid)
// Five vector instructions and one 8-byte LDS read per hidden value where the quadratic in Phi of rounds 2 took nine and a
// 12-byte read -- the kernel is bound by vector issue -- and a bound that no longer grows 2.4e-6 per unit of zmax:
//   u = fma(d, 256 sc, 256 sh)          the BatchNorm output in node widths
//   t = med3(u + magic, lo, hi)         magic = 1.5 x 2^23: the float32 add rounds u to the nearest integer (ties to even),
//                                       which then sits in the low mantissa bits; the clamp keeps it inside the table
//   entry address = (bits of t) << 3 + constant;   g = fma(u, slope, intercept)
// Beyond the table: node 0 holds (0, 0) (|gelu(z)| < 1e-14 for z <= -8), node 4095 holds (0, SCALE / 256): gelu(z) = z
// to 1e-14 for z >= 7.99, and that IS the line.
constexpr float kNodeMagic = 12582912.0f;            // 1.5 x 2^23
constexpr int kGelN = 4096;
// (dst: LDS, or the plan's copy in global memory -- launch_full_gelu_tables -- which the kernels then only copy)
__device__ inline void gelu_table_to_lds(float2 *dst, const double *erf_tab, double scale) {
  for (int i = threadIdx.x; i < kGelN; i += blockDim.x) {
    const double z = (double)(i - kGelN / 2) * (1.0 / 256.0);
    const double x = z * 0.70710678118654752440, ax = __builtin_fabs(x);
    int k = (int)(ax * 8.0);
    k = k < kErfN - 1 ? k : kErfN - 1;
    const double t = ax - ((double)k + 0.5) * 0.125;
    const double *c = erf_tab + k * kErfC;
    double p = c[8];
    for (int q = 7; q >= 0; --q) p = fma(p, t, c[q]);
    p = ax >= 6.0 ? 1.0 : p;
    const double Phi = 0.5 * (1.0 + __builtin_copysign(p, x));
    const double pdf = 0.39894228040143267794 * exp(-0.5 * z * z);
    const float slope = (float)(scale * (Phi + z * pdf) * (1.0 / 256.0));
    float2 e = make_float2((float)(scale * z * Phi - (double)(i - kGelN / 2) * (double)slope), slope);
    if (i == 0) e = make_float2(0.f, 0.f);
    if (i == kGelN - 1) e = make_float2(0.f, (float)(scale * (1.0 / 256.0)));
    dst[i] = e;
  }
}
__device__ inline void gelu_table_copy(float2 *dst, const float *src) {
  for (int i = threadIdx.x; i < kGelN / 2; i += blockDim.x) ((float4 *)dst)[i] = ((const float4 *)src)[i];
}
__global__ void gelu_tables_kernel(float *dst) {
  __shared__ double erf_tab[kErfN * kErfC];
  erf_table_to_lds(erf_tab);
  __syncthreads();
  gelu_table_to_lds((float2 *)dst + (size_t)blockIdx.x * kGelN, erf_tab, blockIdx.x ? (double)ACT_PRESCALE : 1.0);
}
__device__ inline uint32_t gelu_lin_node(float u, uint32_t addr_k) {
  const float t = __builtin_amdgcn_fmed3f(u + kNodeMagic, kNodeMagic - (float)(kGelN / 2), kNodeMagic + (float)(kGelN / 2 - 1));
  return (__float_as_uint(t) << 3) + addr_k;
}
template <int SCALE>
__device__ inline float gelu_f32(float zs, const float4 *tab) {
  float dz;
  const int i = gelu_node<SCALE>(zs, dz);
  return gelu_eval(zs, dz, tab[i]);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Host and device: power of two that brings amax into [8192, 16384)
__device__ inline float pow2_prescale(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);
  int k = 14 - e;
  k = k > 60 ? 60 : (k < -60 ? -60 : k);
  return ldexpf(1.0f, k);
}
__device__ inline void split_halves(float v, uint16_t &h1, uint16_t &h2) {
  const _Float16 a = (_Float16)v, b = (_Float16)(v - (float)a);
  h1 = __builtin_bit_cast(uint16_t, a);
  h2 = __builtin_bit_cast(uint16_t, b);
}

constexpr int kFastMT = 15, kFastKP = 8, kFastMid = 240, kFastCin = 30;

// tools/ubench/full_pw_parts.hip builds this file with parts of full_pw_fast_kernel switched off (bit mask:
// 1 no tasks (staging only), 2 no input gather, 4 no layer 1 / GELU, 8 no layer 2, 16 no epilogue).  0 in the library.
#ifndef TT_FULLPW_SKIP
#define TT_FULLPW_SKIP 0
#endif
constexpr int kPwSkip = TT_FULLPW_SKIP;
#ifdef TT_FULLPW_STAMP
__device__ unsigned long long g_pw_stamps[8];        // ns spent by wave 0 of block (0,0): gather, layer 1 + GELU, layer 2, epilogue, tasks
#define PW_STAMP(k) do { if (stamping) { const unsigned long long now = 10ull * __builtin_amdgcn_s_memrealtime(); g_pw_stamps[k] += now - stamp_t; stamp_t = now; } } while (0)
#else
#define PW_STAMP(k) do {} while (0)
#endif
template <int OT>
constexpr size_t fast_lds_bytes() {
  return (size_t)kFastMT * 2 * 64 * 16 + (size_t)kFastKP * OT * 2 * 64 * 16 + 2 * 256 * sizeof(float) + 3 * 32 * sizeof(float) + 64 +
         (size_t)kGelN * 8 + 32 * sizeof(float) + (size_t)kFastKP * OT * 64 * 16 + 2 * 256 * sizeof(float);
}

template <int OT>      // 16-row output tiles: 2 (cout = 30) or 1 (cout = 15)
#ifndef TT_FULLPW_MINWAVES
#define TT_FULLPW_MINWAVES 2      /* one workgroup per CU; 4 = two, at most 128 registers: spills, and no faster */
#endif
__global__ __launch_bounds__(512, TT_FULLPW_MINWAVES) void full_pw_fast_kernel(FullPwArgs a) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  constexpr int CIN = kFastCin, MT = kFastMT, KP = kFastKP, MID = kFastMid;
  uint4 *w1f = (uint4 *)lds_raw;                                // [MT][plane][lane]: layer-1 A fragments
  uint4 *w2f = w1f + MT * 2 * 64;                               // [KP][OT][plane][lane]: layer-2 A fragments
  float *zmx = (float *)(w2f + KP * OT * 2 * 64), *egm = zmx + 256;   // [256] each: zmax_m; eg_m

  float *s2f = egm + 256, *t2f = s2f + 32, *tau = t2f + 32;     // [32] each (tau: the part of the bound that does not depend on the pixel)
  float *tauk = tau + 32;                                       // [32] bound per unit of the accumulated |w2| x |g|
  float *red = tauk + 32;                                       // [16] reductions
  float2 *gel = (float2 *)(red + 16);                           // [4096] GELU table: value, slope per node
  uint4 *w2a = (uint4 *)(gel + kGelN);                          // [KP][OT][lane]: |w2|, high halves, layer-2 fragment order
  float *s1n = (float *)(w2a + KP * OT * 64), *t1n = s1n + 256; // [256] each: BatchNorm scale / layer-1 prescale and shift, in node widths (gelu_lin_node)
  __shared__ double erf_tab[kErfN * kErfC];
  erf_table_to_lds(erf_tab);
  const int g = blockIdx.x, cout = a.cout;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  // ---- stage the group's weights: prescale, split, fragment order; the error bound ------------------
  {
    float m1 = 0.f, m2 = 0.f;
    for (int i = threadIdx.x; i < MID * CIN; i += blockDim.x) m1 = fmaxf(m1, fabsf(a.w1[(size_t)g * MID * CIN + i]));
    for (int i = threadIdx.x; i < cout * MID; i += blockDim.x) m2 = fmaxf(m2, fabsf(a.w2[(size_t)g * cout * MID + i]));
    for (int o = 32; o > 0; o >>= 1) {
      m1 = fmaxf(m1, __shfl_xor(m1, o));
      m2 = fmaxf(m2, __shfl_xor(m2, o));
    }
    if (lane == 0) {
      red[wave] = m1;
      red[8 + wave] = m2;
    }
    if (threadIdx.x == 0) red[15] = 0.f;
    __syncthreads();
    if (a.gel) gelu_table_copy(gel, a.gel);
    else gelu_table_to_lds(gel, erf_tab, (double)ACT_PRESCALE);
    m1 = 0.f;
    m2 = 0.f;
    for (int w = 0; w < nwaves; ++w) {
      m1 = fmaxf(m1, red[w]);
      m2 = fmaxf(m2, red[8 + w]);
    }
    const float ws1 = pow2_prescale(m1), ws2 = pow2_prescale(m2);
    uint16_t *w1h = (uint16_t *)w1f, *w2h = (uint16_t *)w2f;
    // layer 1, A[hidden 16 mt + l%16][input 8 (l/16) + j]
    for (int i = threadIdx.x; i < MT * 64 * 8; i += blockDim.x) {
      const int j = i & 7, l = (i >> 3) & 63, mt = i >> 9;
      const int m = 16 * mt + (l & 15), k = 8 * (l >> 4) + j;
      const float v = k < CIN ? a.w1[((size_t)g * MID + m) * CIN + k] * ws1 : 0.f;
      uint16_t h1, h2;
      split_halves(v, h1, h2);
      w1h[((mt * 2 + 0) * 64 + l) * 8 + j] = h1;
      w1h[((mt * 2 + 1) * 64 + l) * 8 + j] = h2;
    }
    // layer 2, A[output 16 ot + l%16][slot j of lane group q = l/16]: slots 0-3 = hidden 16 (2 kp) + 4 q + j,
    // slots 4-7 = hidden 16 (2 kp + 1) + 4 q + (j - 4): exactly the registers a lane holds after layer 1
    for (int i = threadIdx.x; i < KP * OT * 64 * 8; i += blockDim.x) {
      const int j = i & 7, l = (i >> 3) & 63, ot = (i >> 9) % OT, kp = i / (512 * OT);
      const int o = 16 * ot + (l & 15), q = l >> 4;
      const int hid = 16 * (2 * kp + (j >> 2)) + 4 * q + (j & 3);
      const float v = (o < cout && hid < MID) ? a.w2[((size_t)g * cout + o) * MID + hid] * ws2 : 0.f;
      uint16_t h1, h2;
      split_halves(v, h1, h2);
      w2h[(((kp * OT + ot) * 2 + 0) * 64 + l) * 8 + j] = h1;
      w2h[(((kp * OT + ot) * 2 + 1) * 64 + l) * 8 + j] = h2;
      ((uint16_t *)w2a)[((kp * OT + ot) * 64 + l) * 8 + j] = h1 & 0x7FFFu;
    }
    for (int m = threadIdx.x; m < 256; m += blockDim.x) {
      if (m < MID) {
        double A = 0.0;
        for (int c = 0; c < CIN; ++c) A += fabs((double)a.w1[((size_t)g * MID + m) * CIN + c]);
        const double sc = a.s1[g * MID + m], sh = a.t1[g * MID + m];
        const double zmax = fabs(sc) * A + fabs(sh);
        const double ez = fabs(sc) * A * (4.76837158203125e-7 + 16.0 * 5.9604644775390625e-8) + 3.0 * 5.9604644775390625e-8 * zmax;
        s1n[m] = (float)(sc * 256.0 / (double)ws1);
        t1n[m] = (float)(sh * 256.0);
        zmx[m] = (float)zmax;
        egm[m] = (float)(1.13 * ez + 1.6e-6 + 2e-7 * zmax);        // (gelu_lin_node)
        if (!(zmax * (double)ACT_PRESCALE < 65000.0)) red[15] = 1.0f;     // |gelu(z)| <= |z| <= zmax: only then can a split overflow
      } else {
        zmx[m] = 0.f; egm[m] = 0.f; s1n[m] = 0.f; t1n[m] = 0.f;
      }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int o = threadIdx.x;
      if (o < cout) {
        double E = 0.0, S1 = 0.0;
        for (int m = 0; m < MID; ++m) {
          const double w = fabs((double)a.w2[((size_t)g * cout + o) * MID + m]);
          E += w * (double)egm[m];
          S1 += w;
        }
        const double sc = a.s2[g * cout + o], sh = a.t2[g * cout + o];
        s2f[o] = (float)(sc / ((double)ws2 * (double)ACT_PRESCALE));
        t2f[o] = (float)sh;
        // (the pixel's own sum |w2| |g| is accumulated as sum |w2| g <= that, plus at most 0.34 sum |w2|: gelu >= -0.17)
        tau[o] = (float)(2.0 * (fabs(sc) * (E + 3.2e-6 * 0.35 * S1) + 2.384185791015625e-7 * fabs(sh))) * a.tau_scale;
        tauk[o] = (float)(2.0 * fabs(sc) * 3.2e-6 / ((double)ws2 * (double)ACT_PRESCALE)) * a.tau_scale;
      } else {
        s2f[o] = 0.f; t2f[o] = -1.0f; tau[o] = 0.f; tauk[o] = 0.f;
      }
    }
    __syncthreads();
  }
  const int lg = lane >> 4, ln = lane & 15;
  const int rpw = 64 / a.W, bundles = (a.H + rpw - 1) / rpw;
  const int r = lane / a.W, x = lane - r * a.W;
  const int tasks = a.n * bundles;
  const uint32_t cap = (uint32_t)a.n * (uint32_t)a.H * (uint32_t)a.W;
  bool out_of_range = false;
  const bool check_range = red[15] != 0.f;               // (block-uniform; false for any sane BatchNorm)
  // gelu_lin_node: byte address of table entry i = gel + 8 i, i = (bits of t) - (bits of the magic number) + 2048
  const uint32_t gel_k = (uint32_t)((uint8_t *)gel - lds_raw) + 8u * (uint32_t)(kGelN / 2) - (0x4B400000u << 3);
#ifdef TT_FULLPW_STAMP
  const bool stamping = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  unsigned long long stamp_t = 10ull * __builtin_amdgcn_s_memrealtime();
#endif
  for (int t = blockIdx.y * nwaves + wave; t < ((kPwSkip & 1) ? 0 : tasks); t += gridDim.y * nwaves) {
    PW_STAMP(5);
    // (the quotient of two wave-uniform values comes out of the vector unit: tell the compiler it is uniform)
    const int n = __builtin_amdgcn_readfirstlane(t / bundles), y0 = __builtin_amdgcn_readfirstlane((t % bundles) * rpw), y = y0 + r;
    const bool live = r < rpw && y < a.H;
    // this lane's pixel: its 30 input bits.  Thirty loads off wave-uniform bases (the lane's part of the address
    // is its row inside the bundle), issued together; written with a branch per channel they were issued and
    // waited for one by one, which was most of a task's time.
    uint32_t in = 0;
    if constexpr (kPwSkip & 2) in = (uint32_t)(lane * 0x9E3779B1u + t) & 0x3FFFFFFFu;
    else {
      const int rc = live ? r : 0;
      uint64_t rows[CIN];
      const uint64_t *s0 = a.src[0], *s1 = a.src[1], *s2 = a.src[2], *s3 = a.src[3];
      const size_t img = (size_t)n * a.Csrc * a.H + y0 + rc;     // (the only 64-bit product of the task)
#pragma unroll
      for (int j = 0; j < CIN; ++j) {
        const int J = CIN * g + j;                       // (wave-uniform)
        const int k = J & 3;
        const uint64_t *sk = k == 0 ? s0 : (k == 1 ? s1 : (k == 2 ? s2 : s3));
        rows[j] = a.interleaved ? sk[img + (uint32_t)((J >> 2) * a.H)] : s0[img + (uint32_t)(J * a.H)];
      }
#pragma unroll
      for (int j = 0; j < CIN; ++j) in |= (uint32_t)((rows[j] >> x) & 1ull) << j;
      in = live ? in : 0u;
    }
    // layer-1 B fragments: lane l of pixel tile nt = input bits 8 (l/16) .. + 7 of pixel 16 nt + l%16, as fp16 0 / 1
    uint4 xb[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const uint32_t inp = ((uint32_t)__shfl((int)in, 16 * nt + ln) >> (8 * lg)) & 0xFFu;
      uint32_t d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = ((inp >> (2 * k)) & 1u) * 0x3C00u + ((inp >> (2 * k + 1)) & 1u) * 0x3C000000u;
      xb[nt] = make_uint4(d[0], d[1], d[2], d[3]);
    }
    f32x4 acc[OT][4], accb[OT][4];                       // the outputs; sum |w2| |g| of each (scaled like the outputs)
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[ot][nt] = accb[ot][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    PW_STAMP(0);
#pragma unroll 1
    for (int kp = 0; kp < KP; ++kp) {
      uint32_t g1[4][4], g2[4][4];                     // [pixel tile][dword]: high / low halves of 16 g, slots (0,1) (2,3) (4,5) (6,7)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int mt = 2 * kp + hf;
        if ((kPwSkip & 4) == 0 && mt < MT) {
          const f16x8 wa = __builtin_bit_cast(f16x8, w1f[(mt * 2 + 0) * 64 + lane]), wb = __builtin_bit_cast(f16x8, w1f[(mt * 2 + 1) * 64 + lane]);
          const f32x4 scn = *(const f32x4 *)(s1n + 16 * mt + 4 * lg), shn = *(const f32x4 *)(t1n + 16 * mt + 4 * lg);
          // All four pixel tiles of the hidden tile at once: eight matrix instructions, then sixteen independent
          // BatchNorm / node computations, sixteen table reads in flight, sixteen evaluations.  (Tile by tile, every
          // step waited for the one before: 150 cycles per value with two waves per SIMD.)
          f32x4 d[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, __builtin_bit_cast(f16x8, xb[nt]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, __builtin_bit_cast(f16x8, xb[nt]), d[nt], 0, 0, 0);
          float u[16];
          uint32_t node[16];                            // LDS byte address of the table entry
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {               // register i = hidden unit 16 mt + 4 (l/16) + i
              u[4 * nt + i] = fmaf(d[nt][i], scn[i], shn[i]);
              node[4 * nt + i] = gelu_lin_node(u[4 * nt + i], gel_k);
            }
          float2 cf[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) cf[e] = *(const float2 *)(lds_raw + node[e]);
          float gv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) gv[e] = fmaf(u[e], cf[e].y, cf[e].x);
          if (check_range)
#pragma unroll
            for (int e = 0; e < 16; ++e) out_of_range |= split_out_of_range(gv[e]);
          // high and low halves, two values per conversion (v_cvt_pk_f16_f32)
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const f32x2 v = {gv[4 * nt + 2 * pr], gv[4 * nt + 2 * pr + 1]};
              const f16x2 hi = __builtin_convertvector(v, f16x2);
              f32x2 rest;
              // v - (float)hi, exact, in one instruction each: v_fma_mix_f32 reads the fp16 half directly (the compiler
              // converts back first: one more vector instruction per hidden value)
              const uint32_t hib = __builtin_bit_cast(uint32_t, hi);
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rest.x) : "v"(hib), "v"(v.x));
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rest.y) : "v"(hib), "v"(v.y));
              const f16x2 lo = __builtin_convertvector(rest, f16x2);
              g1[nt][2 * hf + pr] = __builtin_bit_cast(uint32_t, hi);
              g2[nt][2 * hf + pr] = __builtin_bit_cast(uint32_t, lo);
            }
        } else {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) g1[nt][2 * hf] = g1[nt][2 * hf + 1] = g2[nt][2 * hf] = g2[nt][2 * hf + 1] = 0u;
        }
      }
      PW_STAMP(1);
#pragma unroll
      for (int ot = 0; ot < ((kPwSkip & 8) ? 0 : OT); ++ot) {
        const f16x8 wa = __builtin_bit_cast(f16x8, w2f[((kp * OT + ot) * 2 + 0) * 64 + lane]);
        const f16x8 wb = __builtin_bit_cast(f16x8, w2f[((kp * OT + ot) * 2 + 1) * 64 + lane]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f16x8 b1 = __builtin_bit_cast(f16x8, make_uint4(g1[nt][0], g1[nt][1], g1[nt][2], g1[nt][3]));
          const f16x8 b2 = __builtin_bit_cast(f16x8, make_uint4(g2[nt][0], g2[nt][1], g2[nt][2], g2[nt][3]));
          f32x4 c = acc[ot][nt];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, b1, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, b2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, b1, c, 0, 0, 0);
          acc[ot][nt] = c;
        }
        if (!a.out_float) {
          const f16x8 wabs = __builtin_bit_cast(f16x8, w2a[(kp * OT + ot) * 64 + lane]);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            // |g| <= g + 0.34 (gelu >= -0.17): the signed high halves go in as they are and the constant part,
            // 0.34 sum |w2|, sits in tau[o] -- sixteen sign-clearing instructions per k-step fewer
            const f16x8 b1 = __builtin_bit_cast(f16x8, make_uint4(g1[nt][0], g1[nt][1], g1[nt][2], g1[nt][3]));
            accb[ot][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wabs, b1, accb[ot][nt], 0, 0, 0);
          }
        }
      }
    }
    PW_STAMP(2);
    if constexpr (kPwSkip & 16) {
      if (acc[0][0][0] == 123.456f) a.out_rp[t] = in;
      continue;
    }
    // acc[ot][nt][i] = output channel 16 ot + 4 (l/16) + i at pixel 16 nt + l%16
    uint32_t doubt = 0;                                  // bit nt: some output of this lane at pixel tile nt is inside its bound
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = 16 * ot + 4 * lg + i;              // this lane's channel of the quartet
        const bool o_ok = o < cout;
        const float sc = s2f[o], sh = t2f[o], tb = tau[o], tk = tauk[o];
        if (a.out_float) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int p = 16 * nt + ln, pr = p / a.W, px = p - pr * a.W, py = y0 + pr;
            const float pre = fmaf(acc[ot][nt][i], sc, sh);
            if (o_ok && pr < rpw && py < a.H)
              a.out_float[(((size_t)n * a.Cout + g * cout + o) * a.H + py) * a.W + px] = pre > 0.f ? pre : 0.f;
          }
        } else {
          uint64_t bal[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float pre = fmaf(acc[ot][nt][i], sc, sh);
            bal[nt] = __ballot(pre >= 0.f);
            doubt |= (o_ok && !(__builtin_fabsf(pre) >= fmaf(accb[ot][nt][i], tk, tb))) ? (1u << nt) : 0u;      // (a NaN is in doubt too)
          }
          // bits 16 q .. 16 q + 15 of a ballot = channel 16 ot + 4 q + i over the tile's 16 pixels; lane (q, row r' = l%16 < rpw)
          // writes the row word of that channel, image row y0 + r'
          const uint64_t mine = ((bal[0] >> (16 * lg)) & 0xFFFFull) | (((bal[1] >> (16 * lg)) & 0xFFFFull) << 16) |
                                (((bal[2] >> (16 * lg)) & 0xFFFFull) << 32) | (((bal[3] >> (16 * lg)) & 0xFFFFull) << 48);
          if (o_ok && ln < rpw && y0 + ln < a.H)
            a.out_rp[((size_t)n * a.Cout + g * cout + o) * a.H + y0 + ln] = (mine >> (ln * a.W)) & ((1ull << a.W) - 1ull);
        }
      }
    PW_STAMP(3);
    if (!a.out_float && !(kPwSkip & 32)) {
      // a pixel is listed if any of its channels (spread over the four lane groups) is in doubt
      uint32_t mine = 0;                                 // lanes 0-15: bit nt = pixel 16 nt + lane is listed
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const uint64_t b = __ballot((doubt >> nt) & 1u);
        const uint32_t any = (uint32_t)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xFFFFull);
        mine |= ((any >> ln) & 1u) << nt;
      }
      if (lg != 0) mine = 0;
      // drop the pixels beyond the bundle (idle lanes of the last rows)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int p = 16 * nt + ln, pr = p / a.W;
        if (!(pr < rpw && y0 + pr < a.H)) mine &= ~(1u << nt);
      }
      const int cnt = __popc(mine);
      int total = cnt;                                   // wave-wide sum, this lane's offset in it
      int before = 0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(total, o);
        if (lane >= o) total += v;
      }
      before = total - cnt;
      const int wave_total = __shfl(total, 63);
      if (wave_total) {
        uint32_t base = 0;
        if (lane == 0) {
          base = atomicAdd(a.fix_count + g, (uint32_t)wave_total);
          atomicAdd(a.fix_count + 62, (uint32_t)wave_total);            // running total (ttnet_plan_query "full_listed_pw")
        }
        base = (uint32_t)__shfl((int)base, 0);
        uint32_t k = base + (uint32_t)before;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          if ((mine >> nt) & 1u) {
            const int p = 16 * nt + ln, pr = p / a.W, px = p - pr * a.W;
            if (k < cap) a.fix_list[(size_t)g * cap + k] = ((uint32_t)n * (uint32_t)a.H + (uint32_t)(y0 + pr)) * (uint32_t)a.W + (uint32_t)px;
            ++k;
          }
      }
    }
  }
  if (out_of_range && a.range_flag) *a.range_flag = 1u;
}

// ---- depthwise blocks: float32 tables and GELU, float64 where the sign is in doubt ---------------------------
// The scheme of full_pw_fast_kernel for full_dw_tab_kernel: per-row partial sums from float32 tables (entries
// rounded once from float64), float32 BatchNorm / GELU / second layer, and a bound tau on that evaluation's
// error; an output with |pre| < tau goes on a list and full_dw_fix_kernel re-evaluates it in float64, in
// full_dw_tab_kernel's order of summation, so the emitted bits are those of the float64 kernel.
//     ez_m = 2 KH 2^-24 zmax_m                              the row tables hold 256 (s1 x partial sum (+ t1 in row 0)), the
//                                                           BatchNorm output in node widths of the GELU table: KH entries
//                                                           rounded once each and KH - 1 adds, each <= 2^-24 x 256 zmax_m
//                                                           (the factor 2: margin)
//     eg_m = 1.13 ez_m + 1.6e-6 + 2e-7 zmax_m               gelu_lin_node (the 1x1 kernel's tangent-line table, scale 1)
//     E    = sum_m |w2_m| eg_m + 10 x 2^-24 sum_m |w2_m| zmax_m       eight fmas
//     tau  = 2 (|s2| E + 2^-22 |t2|)
// If the list overflows (it holds 1/16 of the outputs; about 1 in 10^5 is listed) the fix kernel recomputes
// every output instead.  Round 3: the BatchNorm folded into the tables and the tangent-line GELU -- per hidden value
// KH table reads, KH - 1 adds, add / med3 / shift-add for the node, one 8-byte read and two fmas, where the quadratic in
// Phi took eighteen vector instructions.
template <int KH, int KW>
__global__ __launch_bounds__(256) void full_dw_fast_kernel(FullDwArgs a) {
  __shared__ float tab[8][KH][1 << KW];
  __shared__ float w2f[8], s2f, t2f, tauf;
  __shared__ __align__(16) float2 gel[kGelN];
  __shared__ double erf_tab[kErfN * kErfC];
  if (!a.gel) {
    erf_table_to_lds(erf_tab);
    __syncthreads();
    gelu_table_to_lds(gel, erf_tab, 1.0);
  } else gelu_table_copy(gel, a.gel);
  const int c = blockIdx.x;
  constexpr int nk = KH * KW;
  for (int i = threadIdx.x; i < 8 * KH * (1 << KW); i += blockDim.x) {
    const int bits = i & ((1 << KW) - 1), kh = (i >> KW) % KH, m = i / (KH << KW);
    double sum = 0.0;
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) sum += ((bits >> kw) & 1) ? (double)a.w1[(size_t)c * 8 * nk + m * nk + kh * KW + kw] : 0.0;
    tab[m][kh][bits] = (float)(256.0 * (a.s1[c * 8 + m] * sum + (kh == 0 ? a.t1[c * 8 + m] : 0.0)));
  }
  if (threadIdx.x == 0) {
    double E = 0.0, S2 = 0.0;
    for (int m = 0; m < 8; ++m) {
      double A = 0.0;
      for (int k = 0; k < nk; ++k) A += fabs((double)a.w1[(size_t)c * 8 * nk + m * nk + k]);
      const double sc = a.s1[c * 8 + m], sh = a.t1[c * 8 + m], w = (double)a.w2[c * 8 + m];
      const double zmax = fabs(sc) * A + fabs(sh);
      const double ez = (2.0 * KH) * 5.9604644775390625e-8 * zmax;
      E += fabs(w) * (1.13 * ez + 1.6e-6 + 2e-7 * zmax);
      S2 += fabs(w) * zmax;
      w2f[m] = (float)w;
    }
    E += 10.0 * 5.9604644775390625e-8 * S2;
    s2f = (float)a.s2[c];
    t2f = (float)a.t2[c];
    tauf = (float)(2.0 * (fabs(a.s2[c]) * E + 2.384185791015625e-7 * fabs(a.t2[c]))) * a.tau_scale;
  }
  __syncthreads();
  const int rows = a.n * a.ho;
  const uint32_t cap = a.fix_cap;
  const uint32_t gel_k = 8u * (uint32_t)(kGelN / 2) - (0x4B400000u << 3);        // byte offset of entry i inside gel = (bits of t) << 3 + this
  for (int t = blockIdx.y * blockDim.x + threadIdx.x; t < rows; t += gridDim.y * blockDim.x) {
    const int n = t / a.ho, oy = t % a.ho;
    uint64_t r[KH];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      const int iy = oy * a.stride - a.pad + kh;
      r[kh] = (iy >= 0 && iy < a.H) ? a.x_rp[((size_t)n * a.C + c) * a.H + iy] << a.pad : 0ull;   // bit 0 = column -pad
    }
    uint64_t out = 0, doubt = 0;
    for (int ox = 0; ox < a.wo; ++ox) {
      uint32_t idx[KH];
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) idx[kh] = (uint32_t)(r[kh] >> (ox * a.stride)) & ((1u << KW) - 1u);
      float u[8];
      uint32_t node[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        float sm = tab[m][0][idx[0]];
#pragma unroll
        for (int kh = 1; kh < KH; ++kh) sm += tab[m][kh][idx[kh]];
        u[m] = sm;
        node[m] = gelu_lin_node(sm, gel_k);
      }
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float2 cf = *(const float2 *)((const uint8_t *)gel + node[m]);
        acc = fmaf(fmaf(u[m], cf.y, cf.x), w2f[m], acc);
      }
      const float pre = fmaf(acc, s2f, t2f);
      out |= (uint64_t)(pre >= 0.f) << (ox + a.pad_l);
      doubt |= (uint64_t)(!(__builtin_fabsf(pre) >= tauf)) << ox;
    }
    a.out[((size_t)n * a.C + c) * a.Ho + oy + a.pad_t] = out;
    while (doubt) {                                      // (rare)
      const int ox = __builtin_ctzll(doubt);
      doubt &= doubt - 1;
      const uint32_t k = atomicAdd(a.fix_count, 1u);
      atomicAdd(a.fix_count + 63, 1u);                   // running total (ttnet_plan_query "full_listed_dw")
      if (k < cap) a.fix_list[k] = (((uint32_t)n * (uint32_t)a.C + (uint32_t)c) * (uint32_t)a.ho + (uint32_t)oy) * (uint32_t)a.wo + (uint32_t)ox;
    }
  }
}

// the listed outputs (or, after an overflow of the list, all of them) in float64.  Eight lanes per output, one per
// hidden unit (its 30 weight loads are independent and in flight together: one thread per output walked 240
// dependent loads and took 54 us for a few thousand outputs); lane 0 of the eight then forms the second layer in
// full_dw_tab_kernel's order.
__global__ __launch_bounds__(256) void full_dw_fix_kernel(FullDwArgs a) {
  __shared__ double erf_tab[kErfN * kErfC];
  const uint32_t listed = *a.fix_count;
  const bool all = listed > a.fix_cap;
  const uint32_t total = (uint32_t)a.n * (uint32_t)a.C * (uint32_t)a.ho * (uint32_t)a.wo;
  const uint32_t work = all ? total : listed;
  if (blockIdx.x * (blockDim.x / 8) >= work) return;
  erf_table_to_lds(erf_tab);
  __syncthreads();
  const int nk = a.kh * a.kw, m = threadIdx.x & 7;
  for (uint32_t t0 = blockIdx.x * (blockDim.x / 8); t0 < work; t0 += gridDim.x * (blockDim.x / 8)) {
    const uint32_t t = t0 + (threadIdx.x >> 3);
    const bool live = t < work;                         // (whole groups of eight lanes)
    const uint32_t id = live ? (all ? t : a.fix_list[t]) : 0u;
    const int ox = (int)(id % (uint32_t)a.wo), oy = (int)((id / (uint32_t)a.wo) % (uint32_t)a.ho);
    const int c = (int)((id / (uint32_t)(a.wo * a.ho)) % (uint32_t)a.C), n = (int)(id / (uint32_t)(a.wo * a.ho * a.C));
    uint32_t win[6];
#pragma unroll
    for (int kh = 0; kh < 6; ++kh) {
      const int iy = oy * a.stride - a.pad + kh;
      const uint64_t row = (kh < a.kh && iy >= 0 && iy < a.H) ? a.x_rp[((size_t)n * a.C + c) * a.H + iy] << a.pad : 0ull;
      win[kh] = (uint32_t)(row >> (ox * a.stride)) & ((1u << a.kw) - 1u);
    }
    // full_dw_tab_kernel's order: the taps of a row kw-ascending, then the rows kh-ascending
    const float *w = a.w1 + (size_t)c * 8 * nk + (size_t)m * nk;
    float wv[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) wv[k] = k < nk ? w[k] : 0.f;
    double sm = 0.0;
#pragma unroll
    for (int kh = 0; kh < 6; ++kh)
      if (kh < a.kh) {
        double part = 0.0;
#pragma unroll
        for (int kw = 0; kw < 6; ++kw)
          if (kw < a.kw) part += ((win[kh] >> kw) & 1u) ? (double)wv[kh * a.kw + kw] : 0.0;
        sm = kh == 0 ? part : sm + part;
      }
    const double gm = gelu_exact(sm * a.s1[c * 8 + m] + a.t1[c * 8 + m], erf_tab);
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double gq = __shfl(gm, (threadIdx.x & 56) + q);           // (the lane index inside the wave: 8 outputs per wave)
      acc = fma(gq, (double)a.w2[c * 8 + q], acc);
    }
    if (live && m == 0) {
      const double pre = acc * a.s2[c] + a.t2[c];
      unsigned long long *word = (unsigned long long *)(a.out + ((size_t)n * a.C + c) * a.Ho + oy + a.pad_t);
      if (pre >= 0.0) atomicOr(word, 1ull << (ox + a.pad_l));
      else atomicAnd(word, ~(1ull << (ox + a.pad_l)));
    }
  }
}

// act(AvgPool2d(2)(x) - 0.5): floor-cropped 2x2 majority on row-packed planes, placed at (pad_t, pad_l)
__global__ void rp_majority_kernel(const uint64_t *x, uint64_t *out, int n, int C, int H, int W, int Ho, int pad_t,
                                   int pad_l) {
  const int Hp = H / 2, Wp = W / 2;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * C * Hp) return;
  const int py = t % Hp;
  const size_t nc = t / Hp;
  const uint64_t a = x[nc * H + 2 * py], b = x[nc * H + 2 * py + 1];
  uint64_t r = 0;
  for (int px = 0; px < Wp; ++px) {
    const int cnt = __popcll((a >> (2 * px)) & 3ull) + __popcll((b >> (2 * px)) & 3ull);
    r |= (uint64_t)(cnt >= 2) << (px + pad_l);
  }
  out[nc * Ho + py + pad_t] = r;
}

// AvgPool2d(2) (floor) of the last block's float output + fp16 x 2 split into lin1's
// fragment order (feature channel ch, pooled pixel pp: k-step (ch/16)*PP + pp, k = ch%16)
__global__ void full_pool_split_kernel(const float *x, uint16_t *feat_frag, int n, int C, int H, int W, uint32_t *range_flag) {
  const int Hp = H / 2, Wp = W / 2, PP = Hp * Wp, KS = (C / 16) * PP;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * C * PP) return;
  const int pp = t % PP, ch = (t / PP) % C, img = t / ((size_t)PP * C);
  const int py = pp / Wp, px = pp % Wp;
  const float *p = x + (((size_t)img * C + ch) * H + 2 * py) * W + 2 * px;
  const float f = (((p[0] + p[1]) + p[W]) + p[W + 1]) * 0.25f;
  store_feature(feat_frag, img, KS, (ch / 16) * PP + pp, ch % 16, f, range_flag);
}

}  // namespace

// TTNET_FULL_EXACT=1: every output in float64 (the round-1 path; what the fast path is tested against).  Read at every
// launch, so that a test can switch it between two forwards of one process (a captured graph keeps what it captured).
static bool full_exact_only() {
  const char *e = getenv("TTNET_FULL_EXACT");
  return e && e[0] == '1';
}
// TTNET_FULL_TAU_SCALE=<f>: multiplies the error bound of the fast path (tests measure its margin with f < 1)
static float full_tau_scale() {
  const char *e = getenv("TTNET_FULL_TAU_SCALE");
  const float v = e ? (float)atof(e) : 1.0f;
  return v > 0.f ? v : 1.0f;
}

size_t full_gelu_tables_elems() { return (size_t)2 * kGelN * 2; }
int launch_full_gelu_tables(float *dst, hipStream_t s) {
  hipLaunchKernelGGL(gelu_tables_kernel, dim3(2), dim3(256), 0, s, dst);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_full_dw(const FullDwArgs &a_in, hipStream_t s) {
  FullDwArgs a = a_in;
  a.tau_scale = full_tau_scale();
  if (a.kh > 6 || a.kw > 6 || a.kh * a.kw > 36 || a.W + 2 * a.pad > 63) {
    set_error("full_dw: unsupported window %dx%d", a.kh, a.kw);
    return TTNET_E_UNSUPPORTED;
  }
  const int rows = a.n * a.ho;
  const int chunks = std::max(1, std::min((rows + 255) / 256, std::max(1, 1024 / a.C)));
  const bool w65 = a.kh == 6 && a.kw == 5, w56 = a.kh == 5 && a.kw == 6;
  if (!full_exact_only() && (w65 || w56) && a.fix_list && a.fix_count && a.fix_cap) {
    TT_HIP(hipMemsetAsync(a.fix_count, 0, sizeof(uint32_t), s));
    if (w65) hipLaunchKernelGGL((full_dw_fast_kernel<6, 5>), dim3(a.C, chunks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((full_dw_fast_kernel<5, 6>), dim3(a.C, chunks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(full_dw_fix_kernel, dim3(1024), dim3(256), 0, s, a);
    TT_HIP(hipGetLastError());
    return TTNET_OK;
  }
  if (a.kh == 6 && a.kw == 5) hipLaunchKernelGGL((full_dw_tab_kernel<6, 5>), dim3(a.C, chunks), dim3(256), 0, s, a);
  else if (a.kh == 5 && a.kw == 6) hipLaunchKernelGGL((full_dw_tab_kernel<5, 6>), dim3(a.C, chunks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(full_dw_kernel, dim3(a.C, chunks), dim3(256), 0, s, a);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_full_pw(const FullPwArgs &a_in, hipStream_t s) {
  FullPwArgs a = a_in;
  a.tau_scale = full_tau_scale();
  if (a.cin == 30 && a.mid == 240 && (a.cout == 30 || a.cout == 15) && a.W <= 64) {
    const int ot = a.cout == 30 ? 2 : 1;
    const size_t lds = sizeof(double) * ((size_t)15 * 8 * 64 + (size_t)15 * 4 * ot * 64 + 2 * 240);
    const int rpw = 64 / a.W, tasks = a.n * ((a.H + rpw - 1) / rpw);
    const int chunks = std::max(1, std::min((tasks + 7) / 8, std::max(1, 512 / a.groups)));
    const bool fast = !full_exact_only() && (a.out_float || (a.fix_list && a.fix_count)) && a.groups <= 62;
    if (fast) {
      if (!a.out_float) TT_HIP(hipMemsetAsync(a.fix_count, 0, 62 * sizeof(uint32_t), s));
      // two workgroups per CU (64 KiB of fragments each)
      const int fchunks = std::max(1, std::min((tasks + 7) / 8, std::max(1, 512 / a.groups)));
      if (ot == 2) {
        TT_TRY(ensure_dynamic_lds((const void *)full_pw_fast_kernel<2>, fast_lds_bytes<2>()));
        hipLaunchKernelGGL(full_pw_fast_kernel<2>, dim3(a.groups, fchunks), dim3(512), fast_lds_bytes<2>(), s, a);
      } else {
        TT_TRY(ensure_dynamic_lds((const void *)full_pw_fast_kernel<1>, fast_lds_bytes<1>()));
        hipLaunchKernelGGL(full_pw_fast_kernel<1>, dim3(a.groups, fchunks), dim3(512), fast_lds_bytes<1>(), s, a);
      }
      TT_HIP(hipGetLastError());
      if (a.out_float) return TTNET_OK;
      // the listed pixels in float64, 16 per wave task: one workgroup per CU (its float64 fragments fill the LDS);
      // workgroups without listed pixels leave at once
      const int xchunks = std::max(1, std::min((tasks / 4 + 7) / 8, std::max(1, 256 / a.groups)));
      if (ot == 2) {
        TT_TRY(ensure_dynamic_lds((const void *)full_pw_mfma_kernel<2, true>, lds));
        hipLaunchKernelGGL((full_pw_mfma_kernel<2, true>), dim3(a.groups, xchunks), dim3(512), lds, s, a);
      } else {
        TT_TRY(ensure_dynamic_lds((const void *)full_pw_mfma_kernel<1, true>, lds));
        hipLaunchKernelGGL((full_pw_mfma_kernel<1, true>), dim3(a.groups, xchunks), dim3(512), lds, s, a);
      }
      TT_HIP(hipGetLastError());
      return TTNET_OK;
    }
    if (ot == 2) {
      TT_TRY(ensure_dynamic_lds((const void *)full_pw_mfma_kernel<2, false>, lds));
      hipLaunchKernelGGL((full_pw_mfma_kernel<2, false>), dim3(a.groups, chunks), dim3(512), lds, s, a);
    } else {
      TT_TRY(ensure_dynamic_lds((const void *)full_pw_mfma_kernel<1, false>, lds));
      hipLaunchKernelGGL((full_pw_mfma_kernel<1, false>), dim3(a.groups, chunks), dim3(512), lds, s, a);
    }
    TT_HIP(hipGetLastError());
    return TTNET_OK;
  }
  if (a.cin > 32 || a.cout > 30 || a.W > 64) {
    set_error("full_pw: unsupported group %d -> %d", a.cin, a.cout);
    return TTNET_E_UNSUPPORTED;
  }
  const size_t lds = sizeof(double) * ((size_t)a.mid * a.cin + (size_t)a.cout * a.mid + 2 * a.mid + 2 * a.cout);
  if (lds > (size_t)kMaxLds) {
    set_error("full_pw: weights (%zu B) exceed LDS", lds);
    return TTNET_E_UNSUPPORTED;
  }
  if (lds > 64 * 1024)
    TT_TRY(ensure_dynamic_lds((const void *)full_pw_kernel, lds));
  const int rpw = 64 / a.W, rows = a.n * ((a.H + rpw - 1) / rpw);        // row bundles (tasks) of the launch
  const int chunks = std::max(1, std::min((rows + 15) / 16, std::max(1, 512 / a.groups)));
  hipLaunchKernelGGL(full_pw_kernel, dim3(a.groups, chunks), dim3(1024), lds, s, a);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_rp_majority(const uint64_t *x, uint64_t *out, int n, int C, int H, int W, int Ho, int pad_t, int pad_l,
                       hipStream_t s) {
  const size_t t = (size_t)n * C * (H / 2);
  hipLaunchKernelGGL(rp_majority_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, x, out, n, C, H, W, Ho, pad_t,
                     pad_l);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_full_pool_split(const float *x, void *feat_frag, int n, int C, int H, int W, uint32_t *range_flag, hipStream_t s) {
  const size_t t = (size_t)n * C * (H / 2) * (W / 2);
  hipLaunchKernelGGL(full_pool_split_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, x, (uint16_t *)feat_frag, n, C,
                     H, W, range_flag);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
