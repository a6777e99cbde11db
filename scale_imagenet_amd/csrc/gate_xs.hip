// Gate path of the x-small variant (fan-in 4): TT_vf_19lv3_imgnet_xsmall,
// models/TT_general_imagenet_v2_xsmall.py:21-148 -- 2x2/s2 depthwise windows, 4-channel
// grouped 1x1 blocks, convf groups of one channel x four branches.
//
// Every truth table has 16 entries (a 16-bit word per output bit), so the whole block's tables
// are a few KiB and the evaluation is pure register bit arithmetic on row-packed planes
// (uint64 per image row, bit x = pixel x); no channel-packed layout is needed.  The work is
// ~40x smaller than the small model's; these kernels favour clarity over the last cycle.

#include "ttnet_common.h"

namespace ttnet {

namespace {

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
  uint64_t xe[4] = {0, 0, 0, 0}, xo[4] = {0, 0, 0, 0};   // rows 2py, 2py+1 of the 4 channels
  uint32_t c3tab[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c3tab[i] = t_c3[g * 16 + i];
  uint64_t r3[4] = {0, 0, 0, 0};
  if (pooled_ok) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t *pl = a.x_rp + ((size_t)n * a.C + 4 * g + k) * a.H;
      xe[k] = pl[2 * py];
      xo[k] = pl[2 * py + 1];
    }
    // conv3 at full resolution on both rows, then 2x2 majority
    for (int px = 0; px < a.W / 2; ++px) {
      uint32_t cnt[4] = {0, 0, 0, 0};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int x = 2 * px + (d & 1);
        uint32_t idx = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) idx |= (uint32_t)((((d >> 1) ? xo[k] : xe[k]) >> x) & 1ull) << k;
        const uint32_t out = c3tab[idx];
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt[k] += (out >> k) & 1u;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) r3[k] |= (uint64_t)(cnt[k] >= 2) << (px + a.off34);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * g + k;
    // out4 = majority of x itself
    uint64_t r4 = 0;
    if (pooled_ok)
      for (int px = 0; px < a.W / 2; ++px) {
        const uint32_t s = (uint32_t)((xe[k] >> (2 * px)) & 3ull), u = (uint32_t)((xo[k] >> (2 * px)) & 3ull);
        r4 |= (uint64_t)(__popc(s) + __popc(u) >= 2) << (px + a.off34);
      }
    // depthwise 2x2 / stride 2 / pad 1: input rows 2oy-1, 2oy; padded column index = x + 1
    const uint64_t *pl = a.x_rp + ((size_t)n * a.C + c) * a.H;
    const int iy0 = 2 * oy - 1;
    const uint64_t ra = (iy0 >= 0 && iy0 < a.H) ? pl[iy0] << 1 : 0ull;
    const uint64_t rb = (iy0 + 1 < a.H) ? pl[iy0 + 1] << 1 : 0ull;
    const uint32_t ta = t_dw1[c], tb = t_dw2[c];       // 16 entries, bit idx; idx bit = kh*2 + kw
    uint64_t r1 = 0, r2 = 0;
    for (int ox = 0; ox < a.Wo; ++ox) {
      const uint32_t idx = (uint32_t)((ra >> (2 * ox)) & 3ull) | ((uint32_t)((rb >> (2 * ox)) & 3ull) << 2);
      r1 |= (uint64_t)((ta >> idx) & 1u) << ox;
      r2 |= (uint64_t)((tb >> idx) & 1u) << ox;
    }
    const size_t dst = ((size_t)n * a.C + c) * a.Ho + oy;
    o1[dst] = r1;
    o2[dst] = r2;
    o3[dst] = r3[k];
    o4[dst] = r4;
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
  uint64_t rows[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int x = 0; x < Wo; ++x) {
    const uint32_t idx = (uint32_t)((a >> x) & 1ull) | ((uint32_t)((b >> x) & 1ull) << 1) |
                         ((uint32_t)((d >> x) & 1ull) << 2) | ((uint32_t)((e >> x) & 1ull) << 3);
    const uint32_t v = t_cf[c * 16 + idx];
#pragma unroll
    for (int k = 0; k < 8; ++k) rows[k] |= (uint64_t)((v >> k) & 1u) << x;
  }
  for (int k = 0; k < cout_g; ++k) out_rp[((size_t)img * C * cout_g + c * cout_g + k) * Ho + oy] = rows[k];
}

// convf of the last block (float table [C][16][cout_g]) + AvgPool2d(2), features written as
// two fp16 planes in lin1's fragment order (feature channel ch, pooled pixel pp:
// k-step = (ch/16)*PP + pp, k = ch%16) -- the same convention as gate_last_kernel.
__global__ void xs_last_kernel(int n, int C, int Ho, int Wo, int cout_g, const uint64_t *o1, const uint64_t *o2,
                               const uint64_t *o3, const uint64_t *o4, const float *t_last, uint16_t *feat_frag) {
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
    store_feature(feat_frag, img, KS, (ch / 16) * PP + pp, ch % 16, f);
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
                   hipStream_t s) {
  if (cout_g > 8 || (C * cout_g) % 16) {
    set_error("xs_last: cout_g=%d", cout_g);
    return TTNET_E_UNSUPPORTED;
  }
  const size_t t = (size_t)n * C * (Ho / 2) * (Wo / 2);
  hipLaunchKernelGGL(xs_last_kernel, dim3((unsigned)((t + 127) / 128)), dim3(128), 0, s, n, C, Ho, Wo, cout_g, o[0], o[1], o[2],
                     o[3], t_last, (uint16_t *)feat_frag);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
