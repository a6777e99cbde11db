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

template <int OT>      // 16-row output tiles: 2 (cout = 30) or 1 (cout = 15)
__global__ __launch_bounds__(512) void full_pw_mfma_kernel(FullPwArgs a) {
  extern __shared__ __align__(16) double lds[];
  __shared__ double erf_tab[kErfN * kErfC];
  erf_table_to_lds(erf_tab);
  constexpr int CIN = 30, MT = 15, KS1 = 8;              // 240 hidden units, K = 30 padded to 32
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
  const int r = lane / a.W, x = lane - r * a.W;
  const int tasks = a.n * bundles;
  for (int t = blockIdx.y * nwaves + wave; t < tasks; t += gridDim.y * nwaves) {
    const int n = t / bundles, y0 = (t % bundles) * rpw, y = y0 + r;
    const bool live = r < rpw && y < a.H;
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
    double xf[KS1][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const uint32_t inp = (uint32_t)__shfl((int)in, 16 * nt + ln) >> lg;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) xf[ks][nt] = ((inp >> (4 * ks)) & 1u) ? 1.0 : 0.0;
    }
    f64x4 acc[OT][4];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[ot][nt] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int mt = 0; mt < MT; ++mt) {
      f64x4 d[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) d[nt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const double wa = w1f[(mt * KS1 + ks) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) d[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, xf[ks][nt], d[nt], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {                       // register i = hidden unit 16mt + 4i + l/16
        const double sc = s1[16 * mt + 4 * i + lg], sh = t1[16 * mt + 4 * i + lg];
        double h[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) h[nt] = gelu_exact(d[nt][i] * sc + sh, erf_tab);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const double wb = w2f[((mt * 4 + i) * OT + ot) * 64 + lane];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[ot][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wb, h[nt], acc[ot][nt], 0, 0, 0);
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
        if (a.out_float) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int p = 16 * nt + ln, pr = p / a.W, px = p - pr * a.W, py = y0 + pr;
            const double pre = acc[ot][nt][i] * sc + sh;
            if (o_ok && pr < rpw && py < a.H)
              a.out_float[(((size_t)n * a.Cout + g * cout + o) * a.H + py) * a.W + px] = (float)(pre > 0.0 ? pre : 0.0);
          }
        } else {
          // one ballot per pixel tile: bits 16q .. 16q+15 = channel quartet member q over the tile's 16 pixels
          uint64_t bal[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) bal[nt] = __ballot(acc[ot][nt][i] * sc + sh >= 0.0);
          // lane (q = l/16, row r' = l%16 < rpw) writes the row word of channel 16ot + 4i + q, image row y0 + r'
          const uint64_t mine = ((bal[0] >> (16 * lg)) & 0xFFFFull) | (((bal[1] >> (16 * lg)) & 0xFFFFull) << 16) |
                                (((bal[2] >> (16 * lg)) & 0xFFFFull) << 32) | (((bal[3] >> (16 * lg)) & 0xFFFFull) << 48);
          if (o_ok && ln < rpw && y0 + ln < a.H)
            a.out_rp[((size_t)n * a.Cout + g * cout + o) * a.H + y0 + ln] = (mine >> (ln * a.W)) & ((1ull << a.W) - 1ull);
        }
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

int launch_full_dw(const FullDwArgs &a, hipStream_t s) {
  if (a.kh > 6 || a.kw > 6 || a.kh * a.kw > 36 || a.W + 2 * a.pad > 63) {
    set_error("full_dw: unsupported window %dx%d", a.kh, a.kw);
    return TTNET_E_UNSUPPORTED;
  }
  const int rows = a.n * a.ho;
  const int chunks = std::max(1, std::min((rows + 255) / 256, std::max(1, 1024 / a.C)));
  if (a.kh == 6 && a.kw == 5) hipLaunchKernelGGL((full_dw_tab_kernel<6, 5>), dim3(a.C, chunks), dim3(256), 0, s, a);
  else if (a.kh == 5 && a.kw == 6) hipLaunchKernelGGL((full_dw_tab_kernel<5, 6>), dim3(a.C, chunks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(full_dw_kernel, dim3(a.C, chunks), dim3(256), 0, s, a);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_full_pw(const FullPwArgs &a, hipStream_t s) {
  if (a.cin == 30 && a.mid == 240 && (a.cout == 30 || a.cout == 15) && a.W <= 64) {
    const int ot = a.cout == 30 ? 2 : 1;
    const size_t lds = sizeof(double) * ((size_t)15 * 8 * 64 + (size_t)15 * 4 * ot * 64 + 2 * 240);
    const int rpw = 64 / a.W, tasks = a.n * ((a.H + rpw - 1) / rpw);
    const int chunks = std::max(1, std::min((tasks + 7) / 8, std::max(1, 512 / a.groups)));
    if (ot == 2) {
      TT_TRY(ensure_dynamic_lds((const void *)full_pw_mfma_kernel<2>, lds));
      hipLaunchKernelGGL(full_pw_mfma_kernel<2>, dim3(a.groups, chunks), dim3(512), lds, s, a);
    } else {
      TT_TRY(ensure_dynamic_lds((const void *)full_pw_mfma_kernel<1>, lds));
      hipLaunchKernelGGL(full_pw_mfma_kernel<1>, dim3(a.groups, chunks), dim3(512), lds, s, a);
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
