// Float stem: AvgPool2d(2) -> Conv2d(3,p,7,stride 2,pad 3,no bias) -> BatchNorm2d -> (x >= 0)
// (models/TT_general_imagenet_v2_small.py:168-169, :183-184; binarisation netbin.py:193),
// emitting the packed bits in both layouts the gate path reads (include/ttnet.h).
//
// Arithmetic.  Plain bf16 or fp16 operands flip ~0.07 % of the stem bits; the exact
// f32 MFMA runs at 1/16 of the 16-bit MFMA rate, and on gfx950 VALU work does not hide under
// another wave's MFMAs (tools/ubench/mfma_valu.hip: the two add up on a SIMD), so both the
// matrix and the vector instruction counts matter.  Every f32 operand is split into two fp16
// terms, v = h1 + h2 + O(2^-22 |v|), after an exact power-of-two prescale that keeps the low
// terms out of the fp16 subnormal range (x * 16, w * 2^k with max|w| 2^k in [8192, 16384); the
// product of the two scales is divided out of the folded BatchNorm scale, exactly), and the
// three products of weight >= 2^-11 are kept:  w2x1 + w1x2 + w1x1, each an MFMA with exact
// fp16 x fp16 products and f32 accumulation.  Measured against the float64 oracle on the
// synthetic model the pre-activation error is <= 1e-6 (the reference's own float32
// conv + BatchNorm deviates 4.9e-6 from float64), ten times below the near-tie band
// (|pre| < 1e-5) inside which the output bits are allowed to differ; the bits are
// oracle-checked, exact except at near ties.  Input range: |x| < 4094 (fp16 overflow of 16 x).
// (Round-1 history: three bf16 terms / six products, 7e-8, cost twice the MFMAs and a third
// LDS plane: 107 us at B = 256.)
//
// Shape.  Implicit GEMM  D[channel][pixel] = W[channel][k] * patch[k][pixel]  with
// k = ((c*7 + kh)*8 + kw), kw padded 7 -> 8 with a zero weight so that one 8-element B-fragment
// is 8 consecutive pooled pixels of one tile row.  v_mfma_f32_32x32x16_f16: M = 32 channels,
// N = 32 output pixels, K = 16 = two (c,kh) rows.  One item = one image x 8 output rows
// (448 pixels = 14 N-tiles); the pooled tile lives in LDS as two fp16 planes; the weights are
// pre-split and pre-swizzled into fragment order at finalize and stream from L2 (45 KB,
// shared by every workgroup).
//
// Bound: 16-bit MFMA (2.5 PFLOP/s dense) at 3 MFMA flops per algorithmic flop (3.6 with the
// kw / row padding) plus the VALU work of the split and the epilogue; 29.5 MMAC/image.

#include <math.h>
#include <string.h>

#include <cmath>

#include "ttnet_common.h"

namespace ttnet {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SR = 8;                  // output rows per workgroup
constexpr int TR = 2 * SR + 5;         // pooled rows in the tile
constexpr int TW = 120;                // tile row pitch in elements (118 used)
constexpr int KSTEPS = 11;             // 22 (c,kh) rows (21 + one zero row), two per MFMA
constexpr int NT = SR * 56 / 32;       // 14 N-tiles of 32 pixels
constexpr int PLANE = 3 * TR * TW;     // elements per fp16 plane

constexpr int NPL = SPLIT_PLANES;      // fp16 planes per operand
constexpr float X_PRESCALE = ACT_PRESCALE;

constexpr int CONS_WAVES = 8, PROD_WAVES = 4, STEM_THREADS = 64 * (CONS_WAVES + PROD_WAVES);
constexpr int UNITS = NT * 2;          // (N-tile, M-tile) pairs of one item: 28
constexpr int UPW = (UNITS + CONS_WAVES - 1) / CONS_WAVES;   // units per consumer wave: 4 (waves 4-7: 3)

// Persistent producer / consumer kernel.  One workgroup per CU walks items (image, block of SR
// output rows).  Producer waves stream the raw float32 rows from HBM, pool them and write the
// two fp16 planes of the NEXT item's tile into the other half of an LDS double buffer;
// consumer waves run the MFMAs and the BN/sign/pack epilogue of the CURRENT item.  One
// workgroup barrier per item.  (With build -> MFMA -> epilogue serial inside a workgroup the
// kernel idled the matrix pipe and HBM alternately: 116 us at B = 256 against a 43 us MFMA
// floor.)  Consumer wave w owns units w, w+8, ...: all of one M-tile (u & 1 = w & 1), so a wave
// needs only that M-tile's weight fragments; waves w and w+4 share a SIMD and carry 4 + 3 units.
__global__ __launch_bounds__(STEM_THREADS) void stem_pc_kernel(const float *__restrict__ x, const uint4 *__restrict__ wfrag,
                                                               const float *__restrict__ scale,
                                                               const float *__restrict__ shift, uint64_t *__restrict__ rp,
                                                               uint16_t *__restrict__ cp, int p, int n_images) {
  extern __shared__ __align__(16) uint8_t smem[];
  uint16_t *tiles = (uint16_t *)smem;                               // [2][NPL * PLANE] fp16
  uint32_t(*stage)[64][NT + 2] = (uint32_t(*)[64][NT + 2])(smem + 2 * NPL * PLANE * 2);   // [2][64][NT+2]
  float *s_scale = (float *)(smem + 2 * NPL * PLANE * 2 + 2 * 64 * (NT + 2) * 4), *s_shift = s_scale + 64;
  const int H = 224, W = 224;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool producer = wave >= CONS_WAVES;
  const int items = n_images * (56 / SR);
  const int my_items = (items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (threadIdx.x < 64) {                               // channels >= p: zero weights, never stored
    s_scale[threadIdx.x] = (int)threadIdx.x < p ? scale[threadIdx.x] : 0.f;
    s_shift[threadIdx.x] = (int)threadIdx.x < p ? shift[threadIdx.x] : 0.f;
  }
  if (threadIdx.x < 128) {
    stage[threadIdx.x >> 6][threadIdx.x & 63][NT] = 0;
    stage[threadIdx.x >> 6][threadIdx.x & 63][NT + 1] = 0;
  }

  // ---- producer side -----------------------------------------------------------------------
  auto build_tile = [&](int item, uint16_t *tile) {
    const int n = item / (56 / SR), oy0 = (item % (56 / SR)) * SR;
    const int pw = wave - CONS_WAVES;                   // 0..3
    // pooled tile row r = pooled image row 2*oy0 - 3 + r; a wave owns (c, r) rows pw, pw+4, ... and
    // walks them in batches of B rows with every global load of the batch in flight
    constexpr int B = 8, ROWS_PER_WAVE = (3 * TR + PROD_WAVES - 1) / PROD_WAVES;
    for (int b0 = 0; b0 < ROWS_PER_WAVE; b0 += B) {
      float2 ra[B][2], rb[B][2];
#pragma unroll
      for (int bi = 0; bi < B; ++bi) {
        const int cr = pw + PROD_WAVES * (b0 + bi);
        const int c = cr / TR, r = cr - c * TR;
        const int iy = 2 * oy0 - 3 + r;
        const bool row_ok = cr < 3 * TR && iy >= 0 && iy < 112;
        const float *src_row = x + (((size_t)n * 3 + (row_ok ? c : 0)) * H + 2 * (row_ok ? iy : 0)) * W;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int ix = lane + 64 * k - 3;
          const bool ok = row_ok && ix >= 0 && ix < 112;
          ra[bi][k] = ok ? *(const float2 *)(src_row + 2 * ix) : make_float2(0.f, 0.f);
          rb[bi][k] = ok ? *(const float2 *)(src_row + W + 2 * ix) : make_float2(0.f, 0.f);
        }
      }
#pragma unroll
      for (int bi = 0; bi < B; ++bi) {
        const int cr = pw + PROD_WAVES * (b0 + bi);
        if (cr < 3 * TR) {
          const int c = cr / TR, r = cr - c * TR;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int px = lane + 64 * k;
            if (px < TW) {
              // pooled value exactly as the reference forms it (x 0.25), then the exact x16 prescale
              const float v = ((((ra[bi][k].x + ra[bi][k].y) + rb[bi][k].x) + rb[bi][k].y) * 0.25f) * X_PRESCALE;
              const _Float16 h1 = (_Float16)v;
              const _Float16 h2 = (_Float16)(v - (float)h1);
              const int e = (c * TR + r) * TW + px;
              tile[e] = __builtin_bit_cast(uint16_t, h1);
              tile[PLANE + e] = __builtin_bit_cast(uint16_t, h2);
            }
          }
        }
      }
    }
  };
  // row words of a finished item from the ballots staged by the consumers
  auto emit_rows = [&](int item, const uint32_t (*st)[NT + 2]) {
    const int n = item / (56 / SR), oy0 = (item % (56 / SR)) * SR;
    for (int idx = threadIdx.x - 64 * CONS_WAVES; idx < 64 * SR; idx += 64 * PROD_WAVES) {
      const int ch = idx & 63, row = idx >> 6;
      const int b0 = 56 * row, w0 = b0 >> 5, sft = b0 & 31;
      const uint64_t lo = st[ch][w0] | ((uint64_t)st[ch][w0 + 1] << 32);
      const uint64_t hi = st[ch][w0 + 2];
      uint64_t v = lo >> sft;
      if (sft) v |= hi << (64 - sft);
      if (ch < p) rp[((size_t)n * p + ch) * 56 + oy0 + row] = v & ((1ull << 56) - 1ull);
    }
  };

  // ---- consumer side -----------------------------------------------------------------------
  const int h = lane >> 5, col = lane & 31;
  const int m = wave & 1;                                // this wave's M-tile (consumers only)
  auto compute_item = [&](int item, const uint16_t *tile, uint32_t (*st)[NT + 2]) {
    const int n = item / (56 / SR), oy0 = (item % (56 / SR)) * SR;
    int pixoff[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const int u = wave + CONS_WAVES * i, t = u >> 1;
      const int pp = 32 * (u < UNITS ? t : 0) + col;
      const int oyl = pp / 56, ox = pp - 56 * oyl;
      pixoff[i] = 2 * oyl * TW + 2 * ox;
    }
    f32x16 acc[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const uint32_t *tile32 = (const uint32_t *)tile;
    // weights in fragment order [ks][plane][mtile][lane] x 16 bytes, fetched one k-step ahead
    uint4 aw_next[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) aw_next[pl] = wfrag[(pl * 2 + m) * 64 + lane];
#pragma unroll 1
    for (int ks = 0; ks < KSTEPS; ++ks) {
      uint4 aw[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) aw[pl] = aw_next[pl];
      if (ks + 1 < KSTEPS) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) aw_next[pl] = wfrag[(((ks + 1) * NPL + pl) * 2 + m) * 64 + lane];
      }
      int R = 2 * ks + h;                    // (c,kh) row of this half-wave's 8 k values
      if (R > 20) R = 20;                    // zero-weight pad row: any finite data
      const int c = (R * 37) >> 8, kh = R - 7 * c;
      const int rowoff = (c * TR + kh) * TW;
      const f16x8 w1 = __builtin_bit_cast(f16x8, aw[0]), w2 = __builtin_bit_cast(f16x8, aw[1]);
#pragma unroll
      for (int i = 0; i < UPW; ++i) {
        if (wave + CONS_WAVES * i < UNITS) {   // wave-uniform
          const int e = (rowoff + pixoff[i]) >> 1;       // dword index: both terms are even
          f16x8 bx[NPL];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            uint4 v;
            v.x = tile32[pl * (PLANE / 2) + e];
            v.y = tile32[pl * (PLANE / 2) + e + 1];
            v.z = tile32[pl * (PLANE / 2) + e + 2];
            v.w = tile32[pl * (PLANE / 2) + e + 3];
            bx[pl] = __builtin_bit_cast(f16x8, v);
          }
          f32x16 a = acc[i];
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, bx[0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, bx[1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, bx[0], a, 0, 0, 0);
          acc[i] = a;
        }
      }
    }
    // epilogue: BN + sign.  C/D layout of the 32x32 MFMA: column = lane&31 (pixel),
    // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (channel within the M-tile).
    const float *sc_l = s_scale + m * 32 + 4 * h, *sh_l = s_shift + m * 32 + 4 * h;   // + (r&3) + 8*(r>>2)
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const int u = wave + CONS_WAVES * i;
      if (u >= UNITS) continue;
      const int t = u >> 1;
      const int pp = 32 * t + col;
      const int oyl = pp / 56, ox = pp - 56 * oyl;
      uint32_t klo = 0, khi = 0;             // lane r keeps the ballot of accumulator register r
      uint32_t pw0 = 0, pw1 = 0;
      static_for<0, 16>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        const float pre = fmaf(acc[i][r], sc_l[(r & 3) + 8 * (r >> 2)], sh_l[(r & 3) + 8 * (r >> 2)]);
        const bool bit = pre >= 0.0f;
        writelane64<r>(klo, khi, __ballot(bit));
        constexpr uint32_t kbit = (r & 3) + 8 * ((r >> 2) & 1);
        if constexpr (r < 8) pw0 |= bit ? (1u << kbit) : 0u;
        else pw1 |= bit ? (1u << kbit) : 0u;
      });
      uint32_t pw = (pw0 | (pw1 << 16)) << (4 * h);
      pw |= (uint32_t)__shfl_xor((int)pw, 32);
      const int q = 2 * m + h;               // half-wave 0 stores group 2m, half-wave 1 group 2m+1
      if (cp) cp[(((size_t)n * 4 + q) * 56 + oy0 + oyl) * 56 + ox] = (uint16_t)(h ? (pw >> 16) : pw);
      if (lane < 16) {                        // lanes 0-31 of the ballot: channel chl, lanes 32-63: chl + 4
        const int chl = m * 32 + (lane & 3) + 8 * (lane >> 2);
        st[chl][t] = klo;
        st[chl + 4][t] = khi;
      }
    }
  };

  // ---- pipeline --------------------------------------------------------------------------------
  __syncthreads();
  if (producer && my_items > 0) build_tile(blockIdx.x, tiles);
  __syncthreads();
  for (int j = 0; j < my_items; ++j) {
    const int item = blockIdx.x + j * gridDim.x;
    if (producer) {
      if (j > 0) emit_rows(item - gridDim.x, stage[(j - 1) & 1]);
      if (j + 1 < my_items) build_tile(item + gridDim.x, tiles + ((j + 1) & 1) * NPL * PLANE);
    } else {
      compute_item(item, tiles + (j & 1) * NPL * PLANE, stage[j & 1]);
    }
    __syncthreads();
  }
  if (producer && my_items > 0) emit_rows(blockIdx.x + (my_items - 1) * gridDim.x, stage[(my_items - 1) & 1]);
}

}  // namespace

// Host side of the operand split: w [64][3][7][7] float32 -> fragment-ordered fp16 planes
// [ks][plane][mtile][lane][8]: lane l of M-tile m holds channel 32m + (l&31), k = 16ks + 8(l>>5) + j,
// k = ((c*7 + kh)*8 + kw); kw = 7 and the 22nd (c,kh) row carry zero weights.  Returns the
// total power-of-two prescale (weights x activations) the caller divides out of the BN scale.
static uint16_t f32_to_f16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  u &= 0x7FFFFFFFu;
  if (u >= 0x47800000u) return sign | 0x7C00u;                    // >= 2^16 (never: prescaled below 2^14)
  if (u < 0x38800000u) {                                          // below 2^-14: fp16 subnormal = round(|f| * 2^24)
    float a;
    memcpy(&a, &u, 4);
    return sign | (uint16_t)nearbyintf(a * 16777216.0f);          // default rounding mode: to nearest even
  }
  u += 0xFFFu + ((u >> 13) & 1u);
  return sign | (uint16_t)((u - 0x38000000u) >> 13);
}
static float f16_to_f32(uint16_t h) {
  const int e = (h >> 10) & 31, mant = h & 1023;
  const float mag = e == 0 ? ldexpf((float)mant, -24) : ldexpf((float)(1024 + mant), e - 25);
  return (h & 0x8000u) ? -mag : mag;
}

float stem_split_weights(const float *w, int p, uint16_t *out) {
  const float ws = weight_prescale(w, (size_t)p * 147);
  for (int ks = 0; ks < KSTEPS; ++ks)
    for (int m = 0; m < 2; ++m)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int ch = 32 * m + (l & 31), R = 2 * ks + (l >> 5), kw = j;
          float v = 0.f;
          if (ch < p && R < 21 && kw < 7) v = w[(size_t)ch * 147 + R * 7 + kw] * ws;     // R = c*7 + kh
          const uint16_t h1 = f32_to_f16_rne(v);
          const uint16_t h2 = f32_to_f16_rne(v - f16_to_f32(h1));
          const uint16_t parts[NPL] = {h1, h2};
          for (int pl = 0; pl < NPL; ++pl) out[((((size_t)ks * NPL + pl) * 2 + m) * 64 + l) * 8 + j] = parts[pl];
        }
  return ws * X_PRESCALE;
}

size_t stem_split_weights_elems() { return (size_t)KSTEPS * NPL * 2 * 64 * 8; }

int launch_stem(const float *x, const void *wfrag, const float *scale, const float *shift, uint64_t *rp,
                uint16_t *cp, int n, int p, hipStream_t s) {
  if (p < 1 || p > 64 || (cp && p != 64)) {
    set_error("stem: p=%d outside [1,64] (channel words need p = 64)", p);
    return TTNET_E_UNSUPPORTED;
  }
  const size_t lds = (size_t)2 * NPL * PLANE * 2 + (size_t)2 * 64 * (NT + 2) * 4 + 128 * 4;
  TT_HIP(hipFuncSetAttribute((const void *)stem_pc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int items = n * (56 / SR);
  hipLaunchKernelGGL(stem_pc_kernel, dim3(std::min(items, 256)), dim3(STEM_THREADS), lds, s, x, (const uint4 *)wfrag, scale,
                     shift, rp, cp, p, n);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
