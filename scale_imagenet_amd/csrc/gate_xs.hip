// Gate path of the x-small variant (fan-in 4): TT_vf_19lv3_imgnet_xsmall,
// models/TT_general_imagenet_v2_xsmall.py:21-148 -- 2x2/s2 depthwise windows, 4-channel
// grouped 1x1 blocks, convf groups of one channel x four branches.
//
// Every truth table has 16 entries, so the whole block's tables are a few KiB and the evaluation
// is pure register bit arithmetic on row-packed planes (uint64 per image row, bit x = pixel x); no
// channel-packed layout is needed.  The tables are evaluated BIT-SLICED: a 4-input function is a
// 15-multiplexer tree (table bits -> x0 -> x1 -> x2 -> x3, one v_bfi per 32 pixels and
// multiplexer), so one pass yields the function at all 64 pixel positions of a row word at once
// instead of one lookup per pixel.  Stride-2 windows and the 2x2 majority are evaluated at every
// bit position on the un-decimated rows and the even positions are then squeezed together.

#include "ttnet_common.h"

namespace ttnet {

namespace {

// f(x0,x1,x2,x3) at all 64 bit positions; bit i of t = f at index i = x0 + 2 x1 + 4 x2 + 8 x3
__device__ inline uint64_t lut4_rows(uint32_t t, uint64_t x0, uint64_t x1, uint64_t x2, uint64_t x3) {
  auto mux = [](uint64_t sel, uint64_t hi, uint64_t lo) { return lo ^ ((hi ^ lo) & sel); };
  uint64_t a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t c0 = (uint32_t)((int32_t)(t << (31 - 2 * j)) >> 31);        // table bit 2j as 0 / ~0
    const uint32_t c1 = (uint32_t)((int32_t)(t << (30 - 2 * j)) >> 31);        // table bit 2j+1
    a[j] = mux(x0, ((uint64_t)c1 << 32) | c1, ((uint64_t)c0 << 32) | c0);
  }
  const uint64_t b0 = mux(x1, a[1], a[0]), b1 = mux(x1, a[3], a[2]), b2 = mux(x1, a[5], a[4]), b3 = mux(x1, a[7], a[6]);
  return mux(x3, mux(x2, b3, b2), mux(x2, b1, b0));
}

// bits 0, 2, 4, ... of x squeezed into bits 0, 1, 2, ... (32 results)
__device__ inline uint64_t even_bits(uint64_t x) {
  x &= 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  return (x | (x >> 16)) & 0x00000000FFFFFFFFull;
}

// column k of a 16-entry table of bytes (entries 0-7 in w0, 8-15 in w1): bit i = bit k of entry i
__device__ inline uint32_t table_column(uint64_t w0, uint64_t w1, int k) {
  auto gather = [](uint64_t w) {            // bit 0 of every byte -> the low 8 bits
    w &= 0x0101010101010101ull;
    w |= w >> 7;
    w |= w >> 14;
    w |= w >> 28;
    return (uint32_t)w & 0xFFu;
  };
  return gather(w0 >> k) | (gather(w1 >> k) << 8);
}

// act(AvgPool2d(2)(.) - 0.5) of rows (e, o): at least two of the four bits of every 2x2 cell
__device__ inline uint64_t majority_rows(uint64_t e, uint64_t o, int cells) {
  const uint64_t ae = e & (e >> 1), xe = e ^ (e >> 1), ao = o & (o >> 1), xo = o ^ (o >> 1);
  return even_bits(ae | ao | (xe & xo)) & ((1ull << cells) - 1ull);
}

// one thread = (image, 4-channel group, output row): Block_conv1, Block_conv2, Block_conv3 +
// majority, majority of x, all four branch rows of its 4 channels, already zero-padded.
__global__ void xs_branches_kernel(GateBlockArgs a, const uint32_t *t_dw1, const uint32_t *t_dw2, const uint8_t *t_c3,
                                   uint64_t *o1, uint64_t *o2, uint64_t *o3, uint64_t *o4) {
  const int G = a.C / 4;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)a.n * G * a.Ho) return;
  const int oy = t % a.Ho, g = (t / a.Ho) % G, n = t / ((size_t)a.Ho * G);
  const int py = oy - a.off34;                       // pooled row feeding out3/out4 (may be outside)
  const bool pooled_ok = py >= 0 && py < a.H / 2;
  const int cells = a.W / 2;
  uint64_t r3[4] = {0, 0, 0, 0}, r4[4] = {0, 0, 0, 0};
  if (pooled_ok) {
    uint64_t xe[4], xo[4];                            // rows 2py, 2py+1 of the 4 channels
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t *pl = a.x_rp + ((size_t)n * a.C + 4 * g + k) * a.H;
      xe[k] = pl[2 * py];
      xo[k] = pl[2 * py + 1];
    }
    const uint64_t w0 = *(const uint64_t *)(t_c3 + g * 16), w1 = *(const uint64_t *)(t_c3 + g * 16 + 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t tk = table_column(w0, w1, k);   // conv3 output bit k as a function of the 4 input channels
      const uint64_t ce = lut4_rows(tk, xe[0], xe[1], xe[2], xe[3]), co = lut4_rows(tk, xo[0], xo[1], xo[2], xo[3]);
      r3[k] = majority_rows(ce, co, cells) << a.off34;
      r4[k] = majority_rows(xe[k], xo[k], cells) << a.off34;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * g + k;
    // depthwise 2x2 / stride 2 / pad 1: input rows 2oy-1, 2oy; padded column index = x + 1.  The
    // window of output column ox is bits 2ox, 2ox+1 of both padded rows: index bit = kh*2 + kw.
    const uint64_t *pl = a.x_rp + ((size_t)n * a.C + c) * a.H;
    const int iy0 = 2 * oy - 1;
    const uint64_t ra = (iy0 >= 0 && iy0 < a.H) ? pl[iy0] << 1 : 0ull;
    const uint64_t rb = (iy0 + 1 < a.H) ? pl[iy0 + 1] << 1 : 0ull;
    const uint64_t keep = (1ull << a.Wo) - 1ull;
    const size_t dst = ((size_t)n * a.C + c) * a.Ho + oy;
    o1[dst] = even_bits(lut4_rows(t_dw1[c], ra, ra >> 1, rb, rb >> 1)) & keep;
    o2[dst] = even_bits(lut4_rows(t_dw2[c], ra, ra >> 1, rb, rb >> 1)) & keep;
    o3[dst] = r3[k];
    o4[dst] = r4[k];
  }
}

// convf of a binarised block: per channel c, index = (out1, out2, out3, out4) at the pixel,
// cout_g output bits -> channels cout_g*c .. of the next block.
__global__ void xs_pf_kernel(int n, int C, int Ho, int Wo, int cout_g, const uint64_t *o1, const uint64_t *o2,
                             const uint64_t *o3, const uint64_t *o4, const uint8_t *t_cf, uint64_t *out_rp) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * C * Ho) return;
  const int oy = t % Ho, c = (t / Ho) % C, img = t / ((size_t)Ho * C);
  const uint64_t a = o1[t], b = o2[t], d = o3[t], e = o4[t];
  const uint64_t w0 = *(const uint64_t *)(t_cf + c * 16), w1 = *(const uint64_t *)(t_cf + c * 16 + 8);
  const uint64_t keep = (1ull << Wo) - 1ull;
  for (int k = 0; k < cout_g; ++k)
    out_rp[((size_t)img * C * cout_g + c * cout_g + k) * Ho + oy] = lut4_rows(table_column(w0, w1, k), a, b, d, e) & keep;
}

// convf of the last block (float table [C][16][cout_g]) + AvgPool2d(2), features written as
// two fp16 planes in lin1's fragment order (feature channel ch, pooled pixel pp:
// k-step = (ch/16)*PP + pp, k = ch%16) -- the same convention as gate_last_kernel.
__global__ void xs_last_kernel(int n, int C, int Ho, int Wo, int cout_g, const uint64_t *o1, const uint64_t *o2,
                               const uint64_t *o3, const uint64_t *o4, const float *t_last, uint16_t *feat_frag,
                               uint32_t *range_flag) {
  const int Hp = Ho / 2, Wp = Wo / 2, PP = Hp * Wp;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * C * PP) return;
  const int pp = t % PP, c = (t / PP) % C, img = t / ((size_t)PP * C);
  const int py = pp / Wp, px = pp % Wp;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float v[4][8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const size_t row = ((size_t)img * C + c) * Ho + 2 * py + (d >> 1);
    const int x = 2 * px + (d & 1);
    const uint32_t idx = (uint32_t)((o1[row] >> x) & 1ull) | ((uint32_t)((o2[row] >> x) & 1ull) << 1) |
                         ((uint32_t)((o3[row] >> x) & 1ull) << 2) | ((uint32_t)((o4[row] >> x) & 1ull) << 3);
    for (int k = 0; k < cout_g; ++k) v[d][k] = t_last[((size_t)c * 16 + idx) * cout_g + k];
  }
  for (int k = 0; k < cout_g; ++k) acc[k] = (((v[0][k] + v[1][k]) + v[2][k]) + v[3][k]) * 0.25f;
  const int Cout = C * cout_g, KS = (Cout / 16) * PP;
  for (int k = 0; k < cout_g; ++k) {
    const int ch = c * cout_g + k;
    const float f = acc[k];
    store_feature(feat_frag, img, KS, (ch / 16) * PP + pp, ch % 16, f, range_flag);
  }
}

}  // namespace

int launch_xs_branches(const GateBlockArgs &a, const void *t_c3, uint64_t *const o[4], hipStream_t s) {
  if (a.kh1 != 2 || a.kw1 != 2 || a.stride != 2 || a.pad != 1 || a.W + 1 > 63) {
    set_error("xs_branches: unsupported geometry");
    return TTNET_E_UNSUPPORTED;
  }
  const size_t t = (size_t)a.n * (a.C / 4) * a.Ho;
  hipLaunchKernelGGL(xs_branches_kernel, dim3((unsigned)((t + 127) / 128)), dim3(128), 0, s, a, (const uint32_t *)a.t_dw1,
                     (const uint32_t *)a.t_dw2, (const uint8_t *)t_c3, o[0], o[1], o[2], o[3]);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_xs_pf(int n, int C, int Ho, int Wo, int cout_g, uint64_t *const o[4], const void *t_cf, uint64_t *out_rp,
                 hipStream_t s) {
  if (cout_g > 8) {
    set_error("xs_pf: cout_g=%d", cout_g);
    return TTNET_E_UNSUPPORTED;
  }
  const size_t t = (size_t)n * C * Ho;
  hipLaunchKernelGGL(xs_pf_kernel, dim3((unsigned)((t + 127) / 128)), dim3(128), 0, s, n, C, Ho, Wo, cout_g, o[0], o[1], o[2],
                     o[3], (const uint8_t *)t_cf, out_rp);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_xs_last(int n, int C, int Ho, int Wo, int cout_g, uint64_t *const o[4], const float *t_last, void *feat_frag,
                   uint32_t *range_flag, hipStream_t s) {
  if (cout_g > 8 || (C * cout_g) % 16) {
    set_error("xs_last: cout_g=%d", cout_g);
    return TTNET_E_UNSUPPORTED;
  }
  const size_t t = (size_t)n * C * (Ho / 2) * (Wo / 2);
  hipLaunchKernelGGL(xs_last_kernel, dim3((unsigned)((t + 127) / 128)), dim3(128), 0, s, n, C, Ho, Wo, cout_g, o[0], o[1], o[2],
                     o[3], t_last, (uint16_t *)feat_frag, range_flag);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
