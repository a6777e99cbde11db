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
// oracle-checked, exact except at near ties.  Input range: |x| < 4094 (fp16 overflow of 16 x); a
// pooled value outside it raises the plan's range flag (ttnet.h: TTNET_E_RANGE), it never passes silently.
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
#include <vector>

#include "ttnet_common.h"

namespace ttnet {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SR = 8;                  // output rows per workgroup
constexpr int TR = 2 * SR + 5;         // pooled rows in the tile
constexpr int TW = 120;                // tile row width in elements (118 used)
constexpr int KSTEPS = 11;             // 22 (c,kh) rows (21 + one zero row), two per MFMA
constexpr int NT = SR * 56 / 32;       // 14 N-tiles of 32 pixels
constexpr int NPL = SPLIT_PLANES;      // fp16 planes per operand
constexpr int ROWP = NPL * TW;         // LDS pitch of a tile row: [plane 0: TW][plane 1: TW], so that one
                                       // address register reaches both planes of a fragment (ds_read2 offsets)
constexpr int TILE = 3 * TR * ROWP;    // elements per tile buffer
constexpr float X_PRESCALE = ACT_PRESCALE;

constexpr int CONS_WAVES = 8, PROD_WAVES = 4, STEM_THREADS = 64 * (CONS_WAVES + PROD_WAVES);
constexpr int UNITS = NT * 2;          // (N-tile, M-tile) pairs of one item: 28
constexpr int UPW = (UNITS + CONS_WAVES - 1) / CONS_WAVES;   // units per consumer wave: 4 (waves 4-7: 3)

// Persistent producer / consumer kernel.  One workgroup per CU walks items (image, block of SR
// output rows).  Producer waves stream the raw float32 rows from HBM, pool them and write the
// two fp16 planes of the NEXT item's tile into the other half of an LDS double buffer;
// consumer waves run the MFMAs and the sign/pack epilogue of the CURRENT item.  One workgroup
// barrier per item.  Consumer wave w owns units w, w+8, ...: all of one M-tile (u & 1 = w & 1), so
// a wave needs only that M-tile's weight fragments; waves w and w+4 share a SIMD and carry 4 + 3
// units.  BatchNorm is folded: its scale into the weights (host), its shift into the initial
// value of the accumulators, so the epilogue is the sign bit alone.
//
// Every vector instruction counts here (MFMA and VALU time add up on a SIMD): wave-uniform
// indices are forced into SGPRs, border handling is a clamp of the load address plus a 0/4
// multiplier instead of per-element selects, sign bits are collected by funnel shifts and turned
// into row words by an in-register bit transpose (no ballots), and everything that does not
// depend on the item is computed once.
//
// U8 = true: the input is the decoder's uint8 HWC image and the last two steps of
// the input pipeline, ToTensor (/255) and Normalize(mean, std) (utils/preprocess.py:104-108),
// are fused in front of the average pool: the four bytes of a pooled pixel and channel are summed
// as integers (v_dot4 with a byte selector) and the sum (0..1020) indexes a table of already
// split values  ((s/4)/255 - mean_c)/std_c x prescale  built on the host in float64.  A quarter of
// the input bytes, fewer vector instructions; the value is the real-arithmetic one rounded once
// (the float32 path rounds each pixel and each add: <= 2e-7 apart on a pooled value).
// CP = also emit the channel-word layout (read only by the two-launch gate kernels of gate.hip: --layers 3 / 4,
// x-small, TTNET_GATE_UNFUSED); the block-fused gate path reads rows alone, and the word formation and its
// cross-lane exchange are then compiled out of the epilogue.
template <bool U8, bool CP>
__global__ __launch_bounds__(STEM_THREADS) void stem_pc_kernel(const void *__restrict__ xin, const uint4 *__restrict__ wfrag,
                                                               const float *__restrict__ init, uint64_t *__restrict__ rp,
                                                               uint16_t *__restrict__ cp, int p, int n_images,
                                                               const uint32_t *__restrict__ norm_tab, uint32_t *range_flag) {
  const float *x = (const float *)xin;
  const uint8_t *xu8 = (const uint8_t *)xin;
  extern __shared__ __align__(16) uint8_t smem[];
  uint16_t *tiles = (uint16_t *)smem;                               // [2][TILE] fp16
  uint32_t(*stage)[64][NT + 2] = (uint32_t(*)[64][NT + 2])(smem + 2 * TILE * 2);   // [2][64][NT+2]
  float *s_init = (float *)(smem + 2 * TILE * 2 + 2 * 64 * (NT + 2) * 4);          // [mtile][half][16] accumulator start values
  uint4 *s_w = (uint4 *)(s_init + 64);                                             // weight fragments [ks][plane][mtile][lane]
  uint32_t *s_norm = (uint32_t *)(s_w + KSTEPS * NPL * 2 * 64);                    // U8: [3][1024] h1 | h2 << 16
  // The weight fragments (44 KiB) stay in LDS for the life of the workgroup: fetched from L2 every
  // k-step they put an L2 round trip (under the producers' HBM traffic) on each of the 11 k-steps
  // of every item -- the matrix pipe then idled for half of the MFMA phase.
  for (int i = threadIdx.x; i < KSTEPS * NPL * 2 * 64; i += STEM_THREADS) s_w[i] = wfrag[i];
  if constexpr (U8)
    for (int i = threadIdx.x; i < 3 * 1024; i += STEM_THREADS) s_norm[i] = norm_tab[i];
  const int H = 224, W = 224;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wave >= CONS_WAVES;
  const int items = n_images * (56 / SR);
  const int my_items = (items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (threadIdx.x < 64) {
    // C/D layout of the 32x32 MFMA: row (channel within the M-tile) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int mm = threadIdx.x >> 5, hh = (threadIdx.x >> 4) & 1, r = threadIdx.x & 15;
    s_init[threadIdx.x] = init[32 * mm + (r & 3) + 8 * (r >> 2) + 4 * hh];
  }
  if (threadIdx.x < 128) {
    stage[threadIdx.x >> 6][threadIdx.x & 63][NT] = 0;
    stage[threadIdx.x >> 6][threadIdx.x & 63][NT + 1] = 0;
  }

  // ---- producer side -----------------------------------------------------------------------
  // tile column px = pooled image column px - 3; a lane handles px = lane and px = lane + 64
  int colc[2];
  float colm[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int ix = lane + 64 * k - 3;
    colc[k] = 2 * min(max(ix, 0), 111);
    colm[k] = (ix >= 0 && ix < 112) ? 0.25f * X_PRESCALE : 0.0f;     // average of four, prescale; 0 in the padding
  }
  // U8: byte offset of the pooled pixel's two raw pixels (6 bytes: r g b r g b) in a raw row; mask
  // of the padding columns
  int colb[2];
  uint32_t colk[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int ix = lane + 64 * k - 3;
    colb[k] = 6 * min(max(ix, 0), 111);
    colk[k] = (ix >= 0 && ix < 112) ? 0xFFFFFFFFu : 0u;
  }
  constexpr int RPW8 = (TR + PROD_WAVES - 1) / PROD_WAVES;          // U8: pooled rows per producer wave (6), 3 channels each
  typedef uint32_t __attribute__((aligned(2))) u32_a2;
  uint32_t qa[RPW8][2][2], qb[RPW8][2][2];                          // [row][chunk][0: bytes 0-3, 1: bytes 4-5] of raw rows 2iy, 2iy+1
  // pooled tile row r = pooled image row 2*oy0 - 3 + r; a producer wave owns the (c, r) rows pw,
  // pw+4, ...  All global loads of an item are issued at once, one item ahead: they are in flight
  // across the workgroup barrier and while the row words of the previous item are emitted, so
  // the HBM latency is not on the per-item critical path (a producer that loads, waits and
  // splits in turn needs 15.5 k cycles per item against 12 k for the consumers).
  constexpr int RPW = (3 * TR + PROD_WAVES - 1) / PROD_WAVES;        // rows per producer wave: 16
  float2 ra[RPW][2], rb[RPW][2];
  bool out_of_range = false;             // a pooled, prescaled value beyond fp16 (|x| >= 4094): see split_out_of_range
  auto issue_loads = [&](int item) {
    const int n = item / (56 / SR), oy0 = (item % (56 / SR)) * SR;
    const int pw = wave - CONS_WAVES;                   // 0..3
    if constexpr (U8) {
#pragma unroll
      for (int bi = 0; bi < RPW8; ++bi) {
        const int r = pw + PROD_WAVES * bi;             // wave-uniform
        const int iy = 2 * oy0 - 3 + r;
        const bool row_ok = r < TR && iy >= 0 && iy < 112;
        const uint8_t *src_row = xu8 + ((size_t)n * H + 2 * (row_ok ? iy : 0)) * (W * 3);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          qa[bi][k][0] = *(const u32_a2 *)(src_row + colb[k]);
          qa[bi][k][1] = *(const uint16_t *)(src_row + colb[k] + 4);
          qb[bi][k][0] = *(const u32_a2 *)(src_row + W * 3 + colb[k]);
          qb[bi][k][1] = *(const uint16_t *)(src_row + W * 3 + colb[k] + 4);
        }
      }
      return;
    }
#pragma unroll
    for (int bi = 0; bi < RPW; ++bi) {
      const int cr = pw + PROD_WAVES * bi;              // wave-uniform
      const int c = cr / TR, r = cr - c * TR;
      const int iy = 2 * oy0 - 3 + r;
      const bool row_ok = cr < 3 * TR && iy >= 0 && iy < 112;
      const float *src_row = x + (((size_t)n * 3 + (row_ok ? c : 0)) * H + 2 * (row_ok ? iy : 0)) * W;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        ra[bi][k] = *(const float2 *)(src_row + colc[k]);
        rb[bi][k] = *(const float2 *)(src_row + W + colc[k]);
      }
    }
  };
  auto split_tile = [&](int item, uint16_t *tile) {
    const int oy0 = (item % (56 / SR)) * SR;
    const int pw = wave - CONS_WAVES;
    if constexpr (U8) {
#pragma unroll
      for (int bi = 0; bi < RPW8; ++bi) {
        const int r = pw + PROD_WAVES * bi;
        if (r < TR) {
          const int iy = 2 * oy0 - 3 + r;
          const uint32_t rowk = (iy >= 0 && iy < 112) ? 0xFFFFFFFFu : 0u;     // wave-uniform
          uint16_t *dst = tile + r * ROWP + lane;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            if (lane + 64 * k < TW) {
              const uint32_t a0 = qa[bi][k][0], a1 = qa[bi][k][1], b0 = qb[bi][k][0], b1 = qb[bi][k][1];
              // bytes: a0 = r0 g0 b0 r1, a1 = g1 b1 (likewise the second raw row)
              uint32_t sum[3];
              sum[0] = __builtin_amdgcn_udot4(a0, 0x01000001u, __builtin_amdgcn_udot4(b0, 0x01000001u, 0u, false), false);
              sum[1] = __builtin_amdgcn_udot4(a0, 0x00000100u, __builtin_amdgcn_udot4(b0, 0x00000100u, a1 & 0xFFu, false), false) +
                       (b1 & 0xFFu);
              sum[2] = __builtin_amdgcn_udot4(a0, 0x00010000u, __builtin_amdgcn_udot4(b0, 0x00010000u, a1 >> 8, false), false) +
                       (b1 >> 8);
              const uint32_t keep = colk[k] & rowk;
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const uint32_t hh = s_norm[c * 1024 + sum[c]] & keep;            // zero padding after the normalisation
                dst[c * TR * ROWP + 64 * k] = (uint16_t)hh;
                dst[c * TR * ROWP + 64 * k + TW] = (uint16_t)(hh >> 16);
              }
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int bi = 0; bi < RPW; ++bi) {
      const int cr = pw + PROD_WAVES * bi;
      if (cr < 3 * TR) {
        const int c = cr / TR, r = cr - c * TR;
        const int iy = 2 * oy0 - 3 + r;
        const float rowm = (iy >= 0 && iy < 112) ? 1.0f : 0.0f;      // wave-uniform
        uint16_t *dst = tile + (c * TR + r) * ROWP + lane;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (lane + 64 * k < TW) {
            // pooled value exactly as the reference forms it (x 0.25), times the exact prescale
            const float v = (((ra[bi][k].x + ra[bi][k].y) + rb[bi][k].x) + rb[bi][k].y) * (colm[k] * rowm);
            out_of_range |= split_out_of_range(v);
            const _Float16 h1 = (_Float16)v;
            const _Float16 h2 = (_Float16)(v - (float)h1);
            dst[64 * k] = __builtin_bit_cast(uint16_t, h1);
            dst[64 * k + TW] = __builtin_bit_cast(uint16_t, h2);
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
  const DwLaneConst tk = dw_lane_const(lane);
  const int m = wave & 1;                                // this wave's M-tile (consumers only)
  // per unit, independent of the item: dword offset of the lane's pixel in the tile, and its
  // offset in a channel-word plane
  int pixdw[UPW], cpoff[UPW];
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    const int u = wave + CONS_WAVES * i, t = u >> 1;
    const int pp = 32 * (u < UNITS ? t : 0) + col;
    const int oyl = pp / 56, ox = pp - 56 * oyl;
    pixdw[i] = (2 * oyl * ROWP + 2 * ox) >> 1;
    cpoff[i] = oyl * 56 + ox;
  }
  auto compute_item = [&](int item, const uint16_t *tile, uint32_t (*st)[NT + 2], const f32x16 &start) {
    const int n = item / (56 / SR), oy0 = (item % (56 / SR)) * SR;
    f32x16 acc[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) acc[i] = start;
    const uint32_t *tile32 = (const uint32_t *)tile;
    // weights in fragment order [ks][plane][mtile][lane] x 16 bytes (LDS), read one k-step ahead
    uint4 aw_next[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) aw_next[pl] = s_w[(pl * 2 + m) * 64 + lane];
#pragma unroll 1
    for (int ks = 0; ks < KSTEPS; ++ks) {
      uint4 aw[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) aw[pl] = aw_next[pl];
      if (ks + 1 < KSTEPS) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) aw_next[pl] = s_w[(((ks + 1) * NPL + pl) * 2 + m) * 64 + lane];
      }
      int R = 2 * ks + h;                    // (c,kh) row of this half-wave's 8 k values
      if (R > 20) R = 20;                    // zero-weight pad row: any finite data
      const int c = (R * 37) >> 8, kh = R - 7 * c;
      const uint32_t *rowp = tile32 + (c * TR + kh) * (ROWP / 2);
      const f16x8 w1 = __builtin_bit_cast(f16x8, aw[0]), w2 = __builtin_bit_cast(f16x8, aw[1]);
      // the B fragments of unit i+1 are read from LDS while the three MFMAs of unit i run
      // (two register sets; a wave that reads and multiplies in turn idles the matrix pipe for
      // an LDS round trip per unit)
      auto read_b = [&](int i, uint4 (&v)[NPL]) {
        const uint32_t *q = rowp + pixdw[i];
        v[0].x = q[0]; v[0].y = q[1]; v[0].z = q[2]; v[0].w = q[3];
        v[1].x = q[TW / 2]; v[1].y = q[TW / 2 + 1]; v[1].z = q[TW / 2 + 2]; v[1].w = q[TW / 2 + 3];
      };
      uint4 bq[2][NPL];
      read_b(0, bq[0]);
#pragma unroll
      for (int i = 0; i < UPW; ++i) {
        if (i + 1 < UPW) read_b(i + 1, bq[(i + 1) & 1]);   // a 4th unit that does not exist repeats unit 0's address
        __builtin_amdgcn_sched_barrier(0);                  // keep the reads ahead of the MFMAs
        if (wave + CONS_WAVES * i < UNITS) {   // wave-uniform
          const f16x8 x1 = __builtin_bit_cast(f16x8, bq[i & 1][0]), x2 = __builtin_bit_cast(f16x8, bq[i & 1][1]);
          f32x16 a = acc[i];
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, x1, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x2, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x1, a, 0, 0, 0);
          acc[i] = a;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // epilogue: sign + pack.  C/D layout of the 32x32 MFMA: column = lane&31 (pixel),
    // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (channel within the M-tile).  A lane collects the
    // sign bits of its 16 registers (one funnel shift each: w = w<<1 | sign): that word is its
    // pixel's share of the channel words; transposed across each 16-lane group it becomes, in lane
    // j, register j's bits over the group's 16 pixels, i.e. the row-word pieces.  (A result of
    // exactly -0.0 would count as negative here and as >= 0 in the reference: |pre| = 0 lies in
    // the near-tie band either way.)
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const int u = wave + CONS_WAVES * i;
      if (u >= UNITS) continue;
      const int t = u >> 1;
      uint32_t neg = 0;                      // bit r = sign bit of register r
      static_for<0, 16>([&](auto rr) {
        constexpr int r = 15 - decltype(rr)::value;
        neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(acc[i][r]), 31);
      });
      const uint32_t bits = ~neg & 0xFFFFu;  // bit r = (acc[r] >= 0)
      if constexpr (CP) {
        // channel word bit of register r: (r&3) + 8*((r>>2)&1) + 4*h within the 16-channel group r>>3
        const uint32_t cw0 = bits & 0xFFu, cw1 = bits >> 8;
        uint32_t pw = ((cw0 & 15u) | ((cw0 & 0xF0u) << 4)) | (((cw1 & 15u) | ((cw1 & 0xF0u) << 4)) << 16);
        pw <<= 4 * h;
        pw |= (uint32_t)__shfl_xor((int)pw, 32);
        const int q = 2 * m + h;             // half-wave 0 stores group 2m, half-wave 1 group 2m+1
        cp[((size_t)n * 4 + q) * (56 * 56) + oy0 * 56 + cpoff[i]] = (uint16_t)(h ? (pw >> 16) : pw);
      }
      // row-word pieces: lane j of a 16-lane group = register j over the group's 16 pixels
      const uint32_t piece = transpose16(bits, tk) & 0xFFFFu;
      const uint32_t other = (uint32_t)__shfl_xor((int)piece, 16);
      if ((lane & 16) == 0) {                // lanes 0-15: channels of half 0, lanes 32-47: half 1 (+4)
        const int j = lane & 15;
        st[m * 32 + (j & 3) + 8 * (j >> 2) + 4 * h][t] = piece | (other << 16);
      }
    }
  };

  // ---- pipeline --------------------------------------------------------------------------------
  // Period j: consumers work on item j (tile buffer j&1); producers emit the row words of item
  // j-1, split item j+1 (loaded during period j-1) into the other buffer and issue the loads of
  // item j+2.  Each role runs its own loop with the same my_items + 2 barriers.
  const int g = gridDim.x, first = blockIdx.x;
  if (producer) {
    if (my_items > 0) issue_loads(first);
    __syncthreads();
    if (my_items > 0) split_tile(first, tiles);
    if (my_items > 1) issue_loads(first + g);
    __syncthreads();
    for (int j = 0; j < my_items; ++j) {
      const int item = first + j * g;
      // split first (its loads were issued most of a period ago), refill the load registers at
      // once, and only then the row words of the previous item: the loads get ~3/4 of a period
      if (j + 1 < my_items) split_tile(item + g, tiles + ((j + 1) & 1) * TILE);
      if (j + 2 < my_items) issue_loads(item + 2 * g);
      if (j > 0) emit_rows(item - g, stage[(j - 1) & 1]);
      __syncthreads();
    }
    if (my_items > 0) emit_rows(first + (my_items - 1) * g, stage[(my_items - 1) & 1]);
    if (out_of_range) *range_flag = 1u;
  } else {
    __syncthreads();
    __syncthreads();
    const f32x16 start = *(const f32x16 *)(s_init + (m * 2 + h) * 16);
    for (int j = 0; j < my_items; ++j) {
      compute_item(first + j * g, tiles + (j & 1) * TILE, stage[j & 1], start);
      __syncthreads();
    }
  }
}

}  // namespace

// Host side of the operand split: w [64][3][7][7] float32 -> fragment-ordered fp16 planes
// [ks][plane][mtile][lane][8]: lane l of M-tile m holds channel 32m + (l&31), k = 16ks + 8(l>>5) + j,
// k = ((c*7 + kh)*8 + kw); kw = 7 and the 22nd (c,kh) row carry zero weights.  init[64]: the
// accumulator start values (the folded BN shift in the prescaled unit).
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

void stem_split_weights(const float *w, const double *scale, const double *shift, int p, uint16_t *out, float *init) {
  // BatchNorm scale folded into the weights (float32 product, like any other float32 rounding of
  // the reference's conv + BN chain), shift into the accumulator start value
  std::vector<float> wf((size_t)p * 147);
  for (int ch = 0; ch < p; ++ch)
    for (int i = 0; i < 147; ++i) wf[(size_t)ch * 147 + i] = (float)((double)w[(size_t)ch * 147 + i] * scale[ch]);
  const float ws = weight_prescale(wf.data(), wf.size());
  for (int ch = 0; ch < 64; ++ch) init[ch] = ch < p ? (float)(shift[ch] * (double)ws * (double)X_PRESCALE) : -1.0f;
  for (int ks = 0; ks < KSTEPS; ++ks)
    for (int m = 0; m < 2; ++m)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int ch = 32 * m + (l & 31), R = 2 * ks + (l >> 5), kw = j;
          float v = 0.f;
          if (ch < p && R < 21 && kw < 7) v = wf[(size_t)ch * 147 + R * 7 + kw] * ws;     // R = c*7 + kh
          const uint16_t h1 = f32_to_f16_rne(v);
          const uint16_t h2 = f32_to_f16_rne(v - f16_to_f32(h1));
          const uint16_t parts[NPL] = {h1, h2};
          for (int pl = 0; pl < NPL; ++pl) out[((((size_t)ks * NPL + pl) * 2 + m) * 64 + l) * 8 + j] = parts[pl];
        }
}

size_t stem_split_weights_elems() { return (size_t)KSTEPS * NPL * 2 * 64 * 8; }

// U8 input: table [3][1024] of split pooled values, indexed by the integer sum of the four bytes
void stem_norm_table(const float mean[3], const float stdv[3], uint32_t *tab) {
  for (int c = 0; c < 3; ++c)
    for (int sidx = 0; sidx < 1024; ++sidx) {
      const double v = ((((double)sidx / 4.0) / 255.0) - (double)mean[c]) / (double)stdv[c] * (double)X_PRESCALE;
      const float vf = sidx <= 1020 ? (float)v : 0.f;
      const uint16_t h1 = f32_to_f16_rne(vf);
      const uint16_t h2 = f32_to_f16_rne(vf - f16_to_f32(h1));
      tab[c * 1024 + sidx] = (uint32_t)h1 | ((uint32_t)h2 << 16);
    }
}

int launch_stem(const void *x, bool x_is_u8, const uint32_t *norm_tab, const void *wfrag, const float *init, uint64_t *rp,
                uint16_t *cp, int n, int p, uint32_t *range_flag, hipStream_t s) {
  if (p < 1 || p > 64 || (cp && p != 64)) {
    set_error("stem: p=%d outside [1,64] (channel words need p = 64)", p);
    return TTNET_E_UNSUPPORTED;
  }
  const size_t lds = (size_t)2 * TILE * 2 + (size_t)2 * 64 * (NT + 2) * 4 + 64 * 4 + (size_t)KSTEPS * NPL * 2 * 64 * 16 +
                     (x_is_u8 ? 3 * 1024 * 4 : 0);
  const int items = n * (56 / SR);
  auto launch = [&](auto kernel) -> int {
    TT_TRY(ensure_dynamic_lds((const void *)kernel, lds));
    hipLaunchKernelGGL(kernel, dim3(std::min(items, 256)), dim3(STEM_THREADS), lds, s, x, (const uint4 *)wfrag, init, rp, cp, p, n,
                       norm_tab, range_flag);
    return TTNET_OK;
  };
  if (x_is_u8) TT_TRY(cp ? launch(stem_pc_kernel<true, true>) : launch(stem_pc_kernel<true, false>));
  else TT_TRY(cp ? launch(stem_pc_kernel<false, true>) : launch(stem_pc_kernel<false, false>));
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
