// CIFAR "vAlexnet" variant: TT_FHE_XSMALL_vAlexnet (models/TT_FHE_XSMALL_vAlexnet.py:434-676).
//
//   features[0:5]  Conv2d(3,64,3,pad 1)+bias -> ReLU -> BatchNorm2d -> MaxPool2d(3) -> (x >= 0)
//   features[5]    one stride-1 block: Block_conv1 (3x2 window, pad 1), Block_conv2 (2x3),
//                  Block_conv3 (1x1, 8 channels per group), out4 = x; branch zero padding of the
//                  W = 10 rule (:544-550) to 11x11; plain concat (no interleave, no convf)
//   features[6:8]  Flatten -> lin1 (30976 -> 100) -> BatchNorm1d -> lin2 (100 -> 10)
//
// 32x32 inputs and 64- / 256-entry truth tables: a small network; the kernels keep every
// activation on row-packed planes and favour clarity.  The head reuses head.hip (the 0/1
// features are exact in the split-operand format).

#include "ttnet_common.h"

namespace ttnet {

namespace {

// stem: one workgroup per image, thread = (channel, pooled row); the zero-padded image lives in LDS
// (all 64 channel lanes of a wave read the same address: a broadcast), the 27 weights of the
// channel in registers.  Per pooled pixel the 5x5x3 input patch is read once for its nine
// convolution outputs.  Taps are accumulated in (c, kh, kw) order; a tap in the zero border adds
// w * 0, which leaves the sum unchanged.
__global__ __launch_bounds__(640) void va_stem_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, uint64_t *__restrict__ rp, int n) {
  __shared__ float img_s[3][34][34];
  const int img = blockIdx.x;
  for (int i = threadIdx.x; i < 3 * 34 * 34; i += blockDim.x) {
    const int c = i / (34 * 34), r = (i / 34) % 34, q = i % 34;
    const bool in = r >= 1 && r <= 32 && q >= 1 && q <= 32;
    img_s[c][r][q] = in ? x[((size_t)img * 3 + c) * 1024 + (r - 1) * 32 + (q - 1)] : 0.f;
  }
  __syncthreads();
  const int ch = threadIdx.x & 63, py = threadIdx.x >> 6;            // 10 waves: one pooled row each
  float wr[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) wr[i] = w[ch * 27 + i];
  const float b = bias[ch], sc = scale[ch], sh = shift[ch];
  uint64_t out = 0;
  for (int px = 0; px < 10; ++px) {
    float acc[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) acc[dy][dx] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float patch[5][5];
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int q = 0; q < 5; ++q) patch[r][q] = img_s[c][3 * py + r][3 * px + q];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
              acc[dy][dx] = fmaf(patch[dy + kh][dx + kw], wr[(c * 3 + kh) * 3 + kw], acc[dy][dx]);
    }
    float best = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) best = fmaxf(best, fmaf(fmaxf(acc[dy][dx] + b, 0.f), sc, sh));   // ReLU, eval BatchNorm, MaxPool2d(3)
    out |= (uint64_t)(best >= 0.f) << px;
  }
  if (img < n) rp[((size_t)img * 64 + ch) * 10 + py] = out;
}

// Block_conv1 / Block_conv2 / out4: thread = (image, channel, output row of the 11x11 plane).
// Tables: 64 entries, 1 bit, striped dword layout of lut_build.hip ([c/16][2][16]).
__global__ void va_dw_kernel(const uint64_t *__restrict__ x_rp, const uint32_t *__restrict__ t1,
                             const uint32_t *__restrict__ t2, uint64_t *__restrict__ y, int n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * 64 * 11) return;
  const int oy = t % 11, c = (t / 11) % 64, img = t / (11 * 64);
  const uint64_t *pl = x_rp + ((size_t)img * 64 + c) * 10;
  auto row = [&](int iy) -> uint64_t { return (iy >= 0 && iy < 10) ? pl[iy] << 1 : 0ull; };   // bit 0 = column -1
  const size_t tb = (size_t)(c >> 4) * 2 * 16 + (c & 15);
  const uint64_t T1 = t1[tb] | ((uint64_t)t1[tb + 16] << 32), T2 = t2[tb] | ((uint64_t)t2[tb + 16] << 32);
  uint64_t r1 = 0, r2 = 0;
  if (oy < 10) {      // conv1: 3x2 window, rows oy-1..oy+1, 11 output columns; bottom row 10 is zero padding
    const uint64_t a = row(oy - 1), b = row(oy), d = row(oy + 1);
    for (int ox = 0; ox < 11; ++ox) {
      const uint32_t idx = (uint32_t)((a >> ox) & 3) | ((uint32_t)((b >> ox) & 3) << 2) | ((uint32_t)((d >> ox) & 3) << 4);
      r1 |= ((T1 >> idx) & 1ull) << ox;
    }
  }
  {                   // conv2: 2x3 window, rows oy-1..oy, 10 output columns; right column 10 is zero padding
    const uint64_t a = row(oy - 1), b = row(oy);
    for (int ox = 0; ox < 10; ++ox) {
      const uint32_t idx = (uint32_t)((a >> ox) & 7) | ((uint32_t)((b >> ox) & 7) << 3);
      r2 |= ((T2 >> idx) & 1ull) << ox;
    }
  }
  y[((size_t)img * 256 + c) * 11 + oy] = r1;
  y[((size_t)img * 256 + 64 + c) * 11 + oy] = r2;
  y[((size_t)img * 256 + 192 + c) * 11 + oy] = oy < 10 ? pl[oy] : 0ull;       // out4 = x, padded right / bottom
}

// Block_conv3: thread = (image, group of 8 channels, row); table uint8 [8][256]
__global__ void va_c3_kernel(const uint64_t *__restrict__ x_rp, const uint8_t *__restrict__ t3, uint64_t *__restrict__ y,
                             int n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * 8 * 11) return;
  const int oy = t % 11, g = (t / 11) % 8, img = t / 88;
  uint64_t rows[8], out[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    rows[k] = oy < 10 ? x_rp[((size_t)img * 64 + 8 * g + k) * 10 + oy] : 0ull;
    out[k] = 0;
  }
  if (oy < 10)
    for (int xx = 0; xx < 10; ++xx) {
      uint32_t idx = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) idx |= (uint32_t)((rows[k] >> xx) & 1ull) << k;
      const uint32_t v = t3[g * 256 + idx];
#pragma unroll
      for (int k = 0; k < 8; ++k) out[k] |= (uint64_t)((v >> k) & 1u) << xx;
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) y[((size_t)img * 256 + 128 + 8 * g + k) * 11 + oy] = out[k];
}

// Flatten (C-major over [256][11][11]) into lin1's fragment-ordered operand.  The features are
// bits: 0.0 / 1.0 is exact in the split format with a zero low term, so only plane 0 is written
// (plane 1 stays zero from allocation).  Thread = (image, k-step): its 16 features are two aligned
// 16-byte runs of plane 0 (lane halves k < 8 and k >= 8 of the fragment).
__global__ void va_feat_kernel(const uint64_t *__restrict__ y, uint16_t *__restrict__ feat_frag, int n) {
  constexpr int KS = 256 * 121 / 16;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * KS) return;
  const int ks = t % KS, img = t / KS;
  constexpr uint32_t ONE = 0x4C00u;                    // fp16(1.0 * ACT_PRESCALE)
  static_assert(ACT_PRESCALE == 16.0f, "ONE encodes the activation prescale");
  uint32_t half[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int f = 16 * ks + kk, ch = f / 121, rem = f - 121 * ch, oy = rem / 11, ox = rem - 11 * oy;
    const uint32_t bit = (uint32_t)(y[((size_t)img * 256 + ch) * 11 + oy] >> ox) & 1u;
    half[kk >> 3][(kk & 7) >> 1] |= (bit ? ONE : 0u) << (16 * (kk & 1));
  }
  uint4 *dst = (uint4 *)feat_frag + ((((size_t)(img >> 5) * KS + ks) * SPLIT_PLANES + 0) * 64 + (img & 31));
  dst[0] = make_uint4(half[0][0], half[0][1], half[0][2], half[0][3]);
  dst[32] = make_uint4(half[1][0], half[1][1], half[1][2], half[1][3]);
}

__global__ void va_frag_to_flat_kernel(const uint16_t *__restrict__ af, float *__restrict__ out, int n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int K = 256 * 121, KS = K / 16;
  if (t >= (size_t)n * K) return;
  const int f = t % K, img = t / K;
  out[t] = load_feature(af, img, KS, f >> 4, f & 15);
}

}  // namespace

int launch_va_stem(const float *x, const float *w, const float *bias, const float *scale, const float *shift,
                   uint64_t *rp, int n, hipStream_t s) {
  hipLaunchKernelGGL(va_stem_kernel, dim3(n), dim3(640), 0, s, x, w, bias, scale, shift, rp, n);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_va_block(const uint64_t *x_rp, const void *t1, const void *t2, const void *t3, uint64_t *y, int n, hipStream_t s) {
  size_t t = (size_t)n * 64 * 11;
  hipLaunchKernelGGL(va_dw_kernel, dim3((unsigned)((t + 127) / 128)), dim3(128), 0, s, x_rp, (const uint32_t *)t1,
                     (const uint32_t *)t2, y, n);
  t = (size_t)n * 88;
  hipLaunchKernelGGL(va_c3_kernel, dim3((unsigned)((t + 127) / 128)), dim3(128), 0, s, x_rp, (const uint8_t *)t3, y, n);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_va_feat(const uint64_t *y, void *feat_frag, int n, hipStream_t s) {
  const size_t t = (size_t)n * (256 * 121 / 16);
  hipLaunchKernelGGL(va_feat_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, y, (uint16_t *)feat_frag, n);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_va_frag_to_flat(const void *af, float *out, int n, hipStream_t s) {
  const size_t t = (size_t)n * 256 * 121;
  hipLaunchKernelGGL(va_frag_to_flat_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, (const uint16_t *)af, out, n);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
