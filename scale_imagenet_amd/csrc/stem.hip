// Float stem: AvgPool2d(2) -> Conv2d(3,p,7,stride 2,pad 3,no bias) -> BatchNorm2d -> (x >= 0)
// (models/TT_general_imagenet_v2_small.py:168-169, :183-184; binarisation netbin.py:193),
// emitting the packed bits in both layouts the gate path reads (include/ttnet.h).
//
// fp32 throughout: bf16 operands flip ~0.07 % of the stem bits, so the
// contraction stays in exact fp32 FMA.  One workgroup = one image x ROWS output rows; the
// pooled input tile lives in LDS; a wave owns 16 output channels and its 64 lanes own the
// output columns, so the 16 weights of a tap are wave-uniform (scalar loads) and the
// binarised outputs pack with one ballot (row layout) / one shift-or chain (channel layout).
//
// Bound: fp32 VALU/MFMA, 29.5 MMAC per image (157 TFLOP/s peak).

#include "ttnet_common.h"

namespace ttnet {

namespace {

constexpr int ROWS = 4;                    // output rows per workgroup
constexpr int PR = 2 * ROWS + 5;           // pooled rows needed
constexpr int PW = 112 + 6;                // pooled row with 3 columns of zero padding each side
constexpr int PWS = PW + 2;                // LDS row stride (keeps rows 8-byte aligned)

__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                  const float *__restrict__ scale, const float *__restrict__ shift,
                                                  uint64_t *__restrict__ rp, uint16_t *__restrict__ cp, int p) {
  __shared__ float tile[3][PR][PWS];
  const int n = blockIdx.y, oy0 = blockIdx.x * ROWS;
  const int H = 224, W = 224;
  // pooled row r of the tile is pooled image row 2*oy0 - 3 + r; one wave per (c, r) row,
  // lanes along the row (no integer division in the address math)
  {
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    for (int cr = wave_; cr < 3 * PR; cr += 4) {
      const int c = cr / PR, r = cr - c * PR;      // wave-uniform
      const int iy = 2 * oy0 - 3 + r;
      const bool row_ok = iy >= 0 && iy < 112;
      const float *src_row = x + (((size_t)n * 3 + c) * H + 2 * (row_ok ? iy : 0)) * W;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int px = lane_ + 64 * k;
        if (px < PW) {
          const int ix = px - 3;
          float v = 0.f;
          if (row_ok && ix >= 0 && ix < 112) {
            const float2 a = *(const float2 *)(src_row + 2 * ix), b = *(const float2 *)(src_row + W + 2 * ix);
            v = (((a.x + a.y) + b.x) + b.y) * 0.25f;
          }
          tile[c][r][px] = v;
        }
      }
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 16-channel group of this wave
  const int ox = lane < 56 ? lane : 55;
  const int ngroups = p / 16;
  for (int grp = cg; grp < ngroups; grp += 4) {
    float acc[ROWS][16];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
    for (int c = 0; c < 3; ++c) {
      for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
          const float *wp = wt + (size_t)((c * 7 + kh) * 7 + kw) * p + grp * 16;   // wave-uniform
          float v[ROWS];
#pragma unroll
          for (int r = 0; r < ROWS; ++r) v[r] = tile[c][2 * r + kh][2 * ox + kw];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float w = wp[j];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[r][j] = fmaf(v[r], w, acc[r][j]);
          }
        }
      }
    }
    // BN + sign; the ballot of a (row, channel) is its row word; lane r*16+j keeps it
    uint32_t keep_lo = 0, keep_hi = 0;
    const uint64_t live = (1ull << 56) - 1ull;
    static_for<0, ROWS>([&](auto rr) {
      constexpr int r = decltype(rr)::value;
      uint32_t word = 0;
      static_for<0, 16>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        const int ch = grp * 16 + j;
        const float pre = fmaf(acc[r][j], scale[ch], shift[ch]);
        const uint64_t m = __ballot(pre >= 0.0f) & live;
        writelane64<r * 16 + j>(keep_lo, keep_hi, m);
        word |= (pre >= 0.0f) ? (1u << j) : 0u;
      });
      if (lane < 56) cp[(((size_t)n * ngroups + grp) * 56 + oy0 + r) * 56 + lane] = (uint16_t)word;
    });
    const uint64_t keep = ((uint64_t)keep_hi << 32) | keep_lo;
    // lane L holds the row word of channel grp*16 + (L&15), output row oy0 + (L>>4)
    rp[((size_t)n * p + grp * 16 + (lane & 15)) * 56 + oy0 + (lane >> 4)] = keep;
  }
}

}  // namespace

int launch_stem(const float *x, const float *wt, const float *scale, const float *shift, uint64_t *rp,
                uint16_t *cp, int n, int p, hipStream_t s) {
  if (p % 16) {
    set_error("stem: p=%d must be a multiple of 16", p);
    return TTNET_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL(stem_kernel, dim3(56 / ROWS, n), dim3(256), 0, s, x, wt, scale, shift, rp, cp, p);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
