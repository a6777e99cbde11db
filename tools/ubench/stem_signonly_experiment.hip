// EXPERIMENT (round 3), not part of libttnet.so: the stem with a sign-only first pass (one fp16 product, a Cauchy-Schwarz
// error bound per pixel, branch-free flag words) and an exact second pass over the flagged outputs on the diagonal of
// 16x16x32 MFMA tiles -- and, with -DTT_STEM_PRODUCTS=3, the plain three-product chain in the same restructured kernel
// (B fragment shared by both M-tiles, weights from an LDS image, one copy of the x2 plane).  Built and checked by
// tools/ubench/stem_check.hip; measured in profiles/r03_stem_signonly_experiment.txt; DESIGN.md 8 says why it does not ship.
// Its launch_stem has two more parameters than the shipped one (stats, and the tau / EXACT environment knobs).
//
// Float stem: AvgPool2d(2) -> Conv2d(3,p,7,stride 2,pad 3,no bias) -> BatchNorm2d -> (x >= 0)
// (models/TT_general_imagenet_v2_small.py:168-169, :183-184; binarisation netbin.py:193),
// emitting the packed bits in both layouts the gate path reads (include/ttnet.h).
//
// Arithmetic.  Plain bf16 or fp16 operands flip ~0.07 % of the stem bits; the exact
// f32 MFMA runs at 1/16 of the 16-bit MFMA rate.  Every f32 operand is split into two fp16
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
//
// Shape.  Implicit GEMM  D[channel][pixel] = W[channel][k] * patch[k][pixel]  with
// k = ((c*7 + kh)*8 + slot), slot = kw + 1 (slot 0 carries a zero weight), so that the 8-element
// B fragment of output pixel ox is the 8 consecutive pooled pixels 2ox-4 .. 2ox+3 of one tile row:
// four whole dwords of a row that starts at pooled column -4.  v_mfma_f32_32x32x16_f16: M = 32
// channels, N = 32 output pixels, K = 16 = two (c,kh) rows.  One item = one image x 8 output rows
// (448 pixels = 14 N-tiles).
//
// Round 2 (the in-kernel stamps of tools/ubench/stem_parts.hip) gave the kernel its shape: weights in
// registers, every tile row twice in LDS (the second copy one dword to the left, so that the window of
// an odd pixel is 8-byte aligned too: B fragments by conflict-free ds_read_b64), the BatchNorm shift as
// one more k-row, buffer-load producers that read no input byte twice (a workgroup walks a run of
// consecutive row blocks and copies the five shared tile rows inside LDS).
//
// Round 3: ONE product decides almost every output.  Only the sign of the result is used, so the full
// three-product sum is needed only where the first product alone cannot vouch for the sign:
//  * first pass: acc = w1 x1 (11 MFMAs per 32 x 32 tile instead of 33); a wave now owns an N-tile with
//    BOTH M-tiles (one B fragment feeds two MFMAs: half the fragment reads, one sign/pack transpose
//    for 64 channels);
//  * bound: |w2 x1 + w1 x2| <= sum_k (|w2_k| + 2^-11 |w1_k|) |x1_k| <= V[ch] N[pixel] (Cauchy-Schwarz),
//    N^2 = sum x1^2 over the pixel's window, accumulated from the B fragments the lane holds anyway
//    (v_dot2c_f32_f16), V[ch] = || |w2| + 2^-11 |w1| ||_2 known on the host, which scales every output
//    channel's weights (a positive factor: the sign is unchanged) so that V <= 1 - 2^-9: one threshold
//    per pixel, thr = N (+ a constant for fp16 subnormals).  The BatchNorm shift row is exact in the
//    first pass (three fp16 terms in the w1 plane) and takes no part in the bound;
//  * detection is branch-free: per accumulator register one subtract and one funnel shift collect the sign
//    of |acc| - thr into a flag word per lane and unit (0.7 % of the outputs are flagged on the synthetic model);
//  * after its units a wave compacts the flagged (pixel, channel) pairs into a list in LDS and re-evaluates
//    them 16 at a time on the DIAGONAL of a v_mfma_f32_16x16x32_f16 tile: row i = the weights of entry i's
//    channel (gathered from the LDS image of both weight planes), column i = entry i's window (gathered from
//    the tile), 6 k-steps x the three products w2 x1, w1 x2, w1 x1 = 18 MFMAs per 16 entries; the sign of the
//    result replaces the staged bit.  That chain DEFINES the exact value of an output: TTNET_STEM_EXACT=1
//    raises the threshold above every accumulator, so that every output goes through it (a slow reference
//    mode); the listed path equals it bit for bit because the bound guarantees the sign of every unlisted
//    output (tests/test_gpu_parity.py checks it on 64 images, and with the threshold scaled up);
//  * LDS: x2 keeps a single copy (only the diagonal tiles read it), which pays for the weight image;
//  * the channel-word layout (two-launch gate kernels only) is now made from the rows by rp_to_cp_kernel.
//
// Bound: the HBM stream of the float32 input (602 KB per image) beside 1.2 MFMA flops per algorithmic flop
// (slot / row padding; + the listed tiles); 29.5 MMAC/image.

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "../../scale_imagenet_amd/csrc/ttnet_common.h"

namespace ttnet {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int SR = 8;                   // output rows per item
constexpr int NBLK = 56 / SR;           // row blocks (items) per image
constexpr int TR = 2 * SR + 5;          // pooled rows in the tile
constexpr int PITCH = 60;               // dwords per tile row: 120 fp16 = pooled columns -4 .. 115
constexpr int LROWS = 64;                // rows a copy holds: the tile's 63 and a spare one (a producer wave owns 16)
constexpr int COPY_DW = LROWS * PITCH + 32;  // one copy of a plane; the +32 puts copy 1 thirty-two banks from copy 0
constexpr int NPL = SPLIT_PLANES;       // fp16 planes per operand
// a tile buffer: [x1 copy 0: dword j = pixels (2j, 2j+1)][x1 copy 1: dword j = copy 0's dword j+1][x2 copy 0]
constexpr int P0C1 = COPY_DW, P1C0 = 2 * COPY_DW;
constexpr int TILE_DW = 3 * COPY_DW;    // dwords per tile buffer (46,464 B)
constexpr int KSTEPS = 11;              // 22 (c,kh) rows (21 + the BatchNorm-shift row), two per MFMA
constexpr int NT = SR * 56 / 32;        // 14 N-tiles of 32 pixels
constexpr float X_PRESCALE = ACT_PRESCALE;
constexpr int LIST_CAP = 128;           // listed outputs a consumer wave compacts before it runs their diagonal tiles

// Diagnostic builds only (tools/ubench/stem_parts.hip): parts of the kernel switched off, in-kernel stamps.
#ifndef TT_STEM_SKIP
#define TT_STEM_SKIP 0
#endif
#ifndef TT_STEM_LOAD_AUX
#define TT_STEM_LOAD_AUX 0        /* cache policy of the input loads: 0 default, 2 nt */
#endif
#ifndef TT_STEM_PRODUCTS
#define TT_STEM_PRODUCTS 1
#endif
#ifndef TT_STEM_PRIO
#define TT_STEM_PRIO 2
#endif
constexpr int kStemSkip = TT_STEM_SKIP;   // 1 no global loads, 2 no split, 4 no MFMA, 8 no fragment reads, 16 no epilogue, 32 no row words, 64 no tile stores, 128 no bound / listing
#ifdef TT_STEM_STAMP
#ifdef TT_STEM_REALTIME
#define TT_STEM_CLOCK() (10ull * __builtin_amdgcn_s_memrealtime())      /* ns (100 MHz counter) */
#else
#define TT_STEM_CLOCK() __builtin_amdgcn_s_memtime()                    /* shader cycles */
#endif
__device__ unsigned long long g_stem_stamps[256][2][16];
#define STEM_STAMP(role, j) \
  do { if (lane == 0 && (wave == 0 || wave == CONS_WAVES) && (j) < 16) g_stem_stamps[blockIdx.x][role][j] = TT_STEM_CLOCK(); } while (0)
#else
#define STEM_STAMP(role, j) do {} while (0)
#endif

#ifndef TT_STEM_CONS
#define TT_STEM_CONS 8
#endif
#ifndef TT_STEM_PROD
#define TT_STEM_PROD 4
#endif
constexpr int CONS_WAVES = TT_STEM_CONS, PROD_WAVES = TT_STEM_PROD, STEM_THREADS = 64 * (CONS_WAVES + PROD_WAVES);
constexpr int CONST_DW = LROWS * PITCH; // the constant blocks of the BatchNorm-shift row (in the 32-dword pad behind a copy, buffer 0)
// the weight image: [plane][channel][(c,kh) row 0..21][8 slots] fp16; a channel's rows are 368 bytes apart in
// plane 0 (22 x 16 + 16: the 32 lanes of a half-wave, one channel each, read 16 bytes conflict-free) and 352 in plane 1
constexpr int WROW0 = 368, WROW1 = 352, WPLANE1 = 64 * WROW0, WTAB_BYTES = 64 * (WROW0 + WROW1), WTAB_U4 = WTAB_BYTES / 16;
// LDS map (bytes)
constexpr int LDS_STAGE = 2 * TILE_DW * 4, LDS_WTAB = LDS_STAGE + 2 * 64 * (NT + 2) * 4, LDS_LIST = LDS_WTAB + WTAB_BYTES,
              LDS_NORM = LDS_LIST + CONS_WAVES * LIST_CAP * 2, LDS_END_F32 = LDS_NORM, LDS_END_U8 = LDS_NORM + 3 * 1024 * 4;
static_assert(LDS_END_U8 <= kMaxLds, "LDS budget");

// tile row of (c,kh) row R = c*7 + kh (for the lane's output row 0)
constexpr int tile_row(int R) { return (R / 7) * TR + R % 7; }

// Persistent producer / consumer kernel.  One workgroup per CU walks items (image, block of SR
// output rows).  Producer waves stream the raw rows from HBM, pool them and write the fp16 planes of the
// NEXT item's tile into the other half of an LDS double buffer; consumer waves run the MFMAs, the
// listing, the sign/pack epilogue and the listed re-evaluation of the CURRENT item.  One workgroup
// barrier per item.  Consumer wave w owns N-tiles w, w+8 with both M-tiles; waves w and w+4 share a SIMD.
// BatchNorm is folded: its scale into the weights (host), its shift into one more k-row, so the epilogue
// is the sign bit alone.
//
// U8 = true: the input is the decoder's uint8 HWC image and the last two steps of
// the input pipeline, ToTensor (/255) and Normalize(mean, std) (utils/preprocess.py:104-108),
// are fused in front of the average pool: the four bytes of a pooled pixel and channel are summed
// as integers (v_dot4 with a byte selector) and the sum (0..1020) indexes a table of already
// split values  ((s/4)/255 - mean_c)/std_c x prescale  built on the host in float64.  A quarter of
// the input bytes, fewer vector instructions; the value is the real-arithmetic one rounded once
// (the float32 path rounds each pixel and each add: <= 2e-7 apart on a pooled value).
// stats (may be null): [0] listed outputs, [1] diagonal tiles, [2] list flushes before the end of an item.
// tau: scale on the listing threshold (1; tests raise it; 1e30 = every output through the exact chain).
template <bool U8>
__global__ __launch_bounds__(STEM_THREADS) void stem_pc_kernel(const void *__restrict__ xin, const uint4 *__restrict__ wfrag,
                                                               const float *__restrict__ init, uint64_t *__restrict__ rp,
                                                               int p, int n_images, const uint32_t *__restrict__ norm_tab,
                                                               uint32_t *range_flag, uint32_t *stats, float tau) {
  const float *x = (const float *)xin;
  const uint8_t *xu8 = (const uint8_t *)xin;
  extern __shared__ __align__(16) uint8_t smem[];
  uint32_t *tiles = (uint32_t *)smem;                                              // [2][TILE_DW]
  uint32_t(*stage)[64][NT + 2] = (uint32_t(*)[64][NT + 2])(smem + LDS_STAGE);      // [2][64][NT+2]
  uint4 *wtab = (uint4 *)(smem + LDS_WTAB);                                        // the weight image (WROW0 / WROW1)
  uint32_t *s_norm = (uint32_t *)(smem + LDS_NORM);                                // U8: [3][1024] h1 | h2 << 16
  if constexpr (U8)
    for (int i = threadIdx.x; i < 3 * 1024; i += STEM_THREADS) s_norm[i] = norm_tab[i];
  for (int i = threadIdx.x; i < WTAB_U4; i += STEM_THREADS) wtab[i] = wfrag[i];
  const int H = 224, W = 224;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wave >= CONS_WAVES;
  // The folded BatchNorm shift enters the GEMM as one more k-row: the 22nd (c,kh) row reads a block of
  // constants (slots 0..2 = c, a power of two; x2 = 0) instead of pixels, and its weights are the three
  // fp16 terms of shift / c in the w1 plane (stem_split_weights), so the accumulators start from an inline
  // zero and the first product already carries the whole shift.  The blocks sit in the padding behind
  // x1 copy 0 / x2 copy 0 of buffer 0, which no tile row reaches (only the very last padding dword of
  // x1 copy 0 is ever scribbled on).
  if (threadIdx.x < 16) {
    const uint32_t cb = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)init[0]);
    if (threadIdx.x < 8) tiles[CONST_DW + threadIdx.x] = threadIdx.x == 0 ? (cb | (cb << 16)) : (threadIdx.x == 1 ? cb : 0u);
    else tiles[P1C0 + CONST_DW + (threadIdx.x - 8)] = 0u;
  }
  if (threadIdx.x < 128) {
    stage[threadIdx.x >> 6][threadIdx.x & 63][NT] = 0;
    stage[threadIdx.x >> 6][threadIdx.x & 63][NT + 1] = 0;
  }

  // ---- items ---------------------------------------------------------------------------------
  // A workgroup takes a run of consecutive items in (image, row block) order, so most of its items
  // continue the image of the one before: the five tile rows the two share are then copied from the
  // previous tile inside LDS instead of being loaded, pooled and split again (16 of 21 rows to load:
  // no byte of the input is read twice, and a quarter less producer work).
  const int G = gridDim.x, bid = blockIdx.x;
  const long long total_items = (long long)n_images * NBLK;
  const int first_item = (int)(bid * total_items / G);
  const int my_items = (int)((bid + 1) * total_items / G) - first_item;
  auto item_of = [&](int j, int &n, int &oy0) {
    const int it = first_item + j;
    n = it / NBLK;
    oy0 = (it % NBLK) * SR;
  };
  // item j continues the image of item j-1 of this workgroup
  auto continues = [&](int j) -> bool { return j > 0 && (first_item + j) % NBLK != 0; };

  // ---- producer side -----------------------------------------------------------------------
  // Tile column t = pooled image column t - 4.  Lane l < 60 makes dword l of a row = pooled pixels
  // 2l-4 and 2l-3 = raw columns 4l-8 .. 4l-5: one 16-byte load per raw row (12 bytes of uint8).
  // Loads are buffer loads -- lane offset in a VGPR, row offset in an SGPR, the second raw row in the
  // immediate -- whose bounds check supplies the zero padding: a row outside the image loads with an empty
  // buffer, a lane outside the row with an offset beyond any buffer.
  const bool col_ok = lane >= 2 && lane < 58;
  static_assert(PROD_WAVES * 16 == LROWS, "a producer wave owns 16 consecutive tile rows (float32) / 6 pooled rows (uint8)");
  float4 ra[U8 ? 1 : 16], rb[U8 ? 1 : 16];
  typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
  u32x3 qa[U8 ? 6 : 1], qb[U8 ? 6 : 1];
  bool out_of_range = false;             // (a range overflow found outside the packed check below)
  const int pw = wave - CONS_WAVES;      // 0..3 (producers)
  // rows 16..20 of the previous tile are rows 0..4 of this one: 3 channels x 5 rows x 3 arrays (x1 twice, x2),
  // 240 bytes each, as 16-byte pieces, four row arrays per wave instruction
  auto halo_copy = [&](int js, uint32_t *tile) {
    const uint32_t *prev = tiles + ((js - 1) & 1) * TILE_DW;
    const int piece = lane % 15, which = lane / 15;            // lanes 60-63 idle
    if (which < 4) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int id = 12 * pw + 4 * k + which;                 // 0..47, 45 used: (array, channel, row)
        if (id < 45) {
          const int arr = id / 15, cr5 = id - 15 * arr, c = cr5 / 5, rr = cr5 - 5 * c;
          const int base = arr * COPY_DW + (c * TR + rr) * PITCH + 4 * piece;
          const uint4 v = *(const uint4 *)(prev + base + 16 * PITCH);
          *(uint4 *)(tile + base) = v;
        }
      }
    }
  };
  // uint8 input: a producer wave owns 6 consecutive pooled rows of a whole tile (21: the last wave 3), all three
  // channels of each; an item that continues the previous one needs rows 5..20, 4 per wave.  One pass splits
  // item js and refills each row's registers with the same row of item jl as soon as it has been split (the
  // scheme of f32_pass below).
  const uint32_t lane_off8 = col_ok ? (uint32_t)(12 * lane - 24) : 0x7FFF0000u;
  auto u8_pass = [&](auto split_c, int js, int jl, uint32_t *tile) {
    constexpr bool SPLIT = decltype(split_c)::value;
    int ns = 0, oys = 0, nl = 0, oyl0 = 0;
    if (SPLIT) item_of(js, ns, oys);
    const bool load_ok = jl < my_items;
    item_of(load_ok ? jl : 0, nl, oyl0);
    const bool cont_s = SPLIT && continues(js), cont_l = load_ok && continues(jl);
    const void *img = (const void *)(xu8 + (size_t)nl * (H * W * 3));
    if (lane < PITCH) {
      const uint32_t colk = col_ok ? 0xFFFFFFFFu : 0u;
      auto split_row = [&](auto bic) {
        constexpr int bi = decltype(bic)::value;
        const int r = cont_s ? 5 + 4 * pw + bi : 6 * pw + bi;            // wave-uniform
        if (r < TR) {
          const int iy = 2 * oys - 3 + r;
          const uint32_t keep = (iy >= 0 && iy < 112) ? colk : 0u;       // zero padding after the normalisation
          const u32x3 a = qa[bi], b = qb[bi];
          // bytes of a raw row: a.x = r0 g0 b0 r1, a.y = g1 b1 r2 g2, a.z = b2 r3 g3 b3 (pixels 0,1 -> first pooled pixel)
          uint32_t s0[3], s1[3];
          s0[0] = __builtin_amdgcn_udot4(a.x, 0x01000001u, __builtin_amdgcn_udot4(b.x, 0x01000001u, 0u, false), false);
          s0[1] = __builtin_amdgcn_udot4(a.x, 0x00000100u, __builtin_amdgcn_udot4(b.x, 0x00000100u, 0u, false), false) +
                  __builtin_amdgcn_udot4(a.y, 0x00000001u, __builtin_amdgcn_udot4(b.y, 0x00000001u, 0u, false), false);
          s0[2] = __builtin_amdgcn_udot4(a.x, 0x00010000u, __builtin_amdgcn_udot4(b.x, 0x00010000u, 0u, false), false) +
                  __builtin_amdgcn_udot4(a.y, 0x00000100u, __builtin_amdgcn_udot4(b.y, 0x00000100u, 0u, false), false);
          s1[0] = __builtin_amdgcn_udot4(a.y, 0x00010000u, __builtin_amdgcn_udot4(b.y, 0x00010000u, 0u, false), false) +
                  __builtin_amdgcn_udot4(a.z, 0x00000100u, __builtin_amdgcn_udot4(b.z, 0x00000100u, 0u, false), false);
          s1[1] = __builtin_amdgcn_udot4(a.y, 0x01000000u, __builtin_amdgcn_udot4(b.y, 0x01000000u, 0u, false), false) +
                  __builtin_amdgcn_udot4(a.z, 0x00010000u, __builtin_amdgcn_udot4(b.z, 0x00010000u, 0u, false), false);
          s1[2] = __builtin_amdgcn_udot4(a.z, 0x01000001u, __builtin_amdgcn_udot4(b.z, 0x01000001u, 0u, false), false);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const uint32_t e0 = s_norm[c * 1024 + s0[c]] & keep, e1 = s_norm[c * 1024 + s1[c]] & keep;
            const uint32_t d1 = __builtin_amdgcn_perm(e1, e0, 0x05040100u);      // h1 of both pixels
            const uint32_t d2 = __builtin_amdgcn_perm(e1, e0, 0x07060302u);      // h2 of both pixels
            uint32_t *dst = tile + (c * TR + r) * PITCH + lane;
            dst[0] = d1;
            dst[COPY_DW - 1] = d1;             // copy 1, one dword to the left (lane 0 lands in unused padding)
            dst[P1C0] = d2;
          }
        }
      };
      auto load_row = [&](auto bic) {
        constexpr int bi = decltype(bic)::value;
        const int r = cont_l ? 5 + 4 * pw + bi : 6 * pw + bi;
        const int iy = 2 * oyl0 - 3 + r;
        const bool row_ok = load_ok && r < TR && iy >= 0 && iy < 112;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, row_ok ? H * W * 3 : 0, 0x00020000);
        const int soff = row_ok ? 2 * iy * (W * 3) : 0;
        qa[bi] = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)lane_off8, soff, 0);
        qb[bi] = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)lane_off8 + W * 3, soff, 0);
      };
      static_for<0, 4>([&](auto bic) {
        if constexpr (SPLIT && !(kStemSkip & 2)) split_row(bic);
        load_row(bic);
      });
      // the two slots only a whole tile uses (the first item of a run or of an image)
      if constexpr (SPLIT && !(kStemSkip & 2))
        if (!cont_s) static_for<4, 6>([&](auto bic) { split_row(bic); });
      if (!cont_l) static_for<4, 6>([&](auto bic) { load_row(bic); });
    }
    if (cont_s) halo_copy(js, tile);
  };
  // float32 input.  A producer wave owns 16 consecutive (c, r) rows of the tile (the 64th is a spare).
  // One pass splits item js and, row by row, refills each row's registers with the same row of item
  // jl as soon as it has been split: a load has a whole period to land and HBM always has loads of
  // this wave in flight.  A single wave issues at most one instruction per four cycles, so the pass
  // is written to be short: buffer loads (lane offset in a VGPR, row offset in an SGPR, the second
  // raw row in the immediate) instead of 64-bit address arithmetic, out-of-range rows and columns
  // left to the buffer's bounds check (they read as zero: the padding), the range check on the
  // packed halves, one exec mask for the whole pass.  Everything inside is unconditional, so the
  // waits stay counted (vmcnt(30): all but the 30 youngest); an absent item jl loads with an empty
  // buffer (zeros, no traffic).
  constexpr uint32_t IMG_BYTES = 3u * 224u * 224u * 4u;
  const uint32_t lane_off = (lane >= 2 && lane < 58) ? (uint32_t)(4 * lane - 8) * 4u : 0x7FFF0000u;    // padding columns: out of range
  uint32_t ovf = 0;                      // running packed max of |h1|: 0x7C00 and above in either half = fp16 overflow or NaN
  // Rows of a pass: a whole tile is 63 rows, 16 consecutive ones per wave (slot bi -> tile row 16 pw + bi);
  // an item that continues the previous one needs rows 5..20 of each channel, 12 per wave
  // (slot bi < 12 -> channel (12 pw + bi) / 16, row 5 + (12 pw + bi) % 16), slots 12-15 unused.
  auto slot_row = [&](bool cont, int bi, int &c, int &r) {
    const int full = 16 * pw + bi, part = 12 * pw + bi;
    const int cf = (full >= TR) + (full >= 2 * TR) + (full >= 3 * TR);
    c = cont ? part >> 4 : cf;
    r = cont ? 5 + (part & 15) : full - TR * cf;
  };
  auto f32_pass = [&](auto split_c, int js, int jl, uint32_t *tile) {
    constexpr bool SPLIT = decltype(split_c)::value;
    int ns = 0, oys = 0, nl = 0, oyl0 = 0;
    if (SPLIT) item_of(js, ns, oys);
    const bool load_ok = jl < my_items;
    item_of(load_ok ? jl : 0, nl, oyl0);
    const bool cont_s = SPLIT && continues(js), cont_l = load_ok && continues(jl);
    const void *img = (const void *)(x + (size_t)nl * (3 * H * W));
    if (lane < PITCH) {
      auto split_row = [&](auto bic) {
        constexpr int bi = decltype(bic)::value;
        int c, r;
        slot_row(cont_s, bi, c, r);
        // pooled values exactly as the reference forms them (x 0.25), times the exact prescale
        const float v0 = (((ra[bi].x + ra[bi].y) + rb[bi].x) + rb[bi].y) * (0.25f * X_PRESCALE);
        const float v1 = (((ra[bi].z + ra[bi].w) + rb[bi].z) + rb[bi].w) * (0.25f * X_PRESCALE);
        const _Float16 g0 = (_Float16)v0, g1 = (_Float16)v1;
        const _Float16 l0 = (_Float16)(v0 - (float)g0), l1 = (_Float16)(v1 - (float)g1);
        const uint32_t d1 = (uint32_t)__builtin_bit_cast(uint16_t, g0) | ((uint32_t)__builtin_bit_cast(uint16_t, g1) << 16);
        const uint32_t d2 = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
        typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
        const u16x2 mag = __builtin_bit_cast(u16x2, d1 & 0x7FFF7FFFu), old = __builtin_bit_cast(u16x2, ovf);
        ovf = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(mag, old));
        uint32_t *dst = tile + (c * TR + r) * PITCH + lane;       // (the 64th row of a whole tile, c = 3, is the spare)
        if constexpr (kStemSkip & 64) {
          if (d1 == 0x12345678u && d2 == 0x9ABCDEF0u) dst[0] = d1;
          return;
        }
        dst[0] = d1;
        dst[COPY_DW - 1] = d1;             // copy 1, one dword to the left (lane 0 lands in unused padding)
        dst[P1C0] = d2;
      };
      auto load_row = [&](auto bic) {
        constexpr int bi = decltype(bic)::value;
        int c, r;
        slot_row(cont_l, bi, c, r);
        const int iy = 2 * oyl0 - 3 + r;
        const bool row_ok = load_ok && c < 3 && iy >= 0 && iy < 112;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, row_ok ? (int)IMG_BYTES : 0, 0x00020000);
        const int soff = row_ok ? (c * H + 2 * iy) * W * 4 : 0;
        if constexpr (kStemSkip & 1) {
          ra[bi] = make_float4((float)lane, 1.f, 2.f, (float)jl);
          rb[bi] = make_float4(2.f, (float)jl, 1.f, (float)lane);
        } else {
          const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, soff, TT_STEM_LOAD_AUX);
          const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off + W * 4, soff, TT_STEM_LOAD_AUX);
          ra[bi] = __builtin_bit_cast(float4, a);
          rb[bi] = __builtin_bit_cast(float4, b);
        }
      };
      static_for<0, 12>([&](auto bic) {
        if constexpr (SPLIT && !(kStemSkip & 2)) split_row(bic);
        load_row(bic);
      });
      // the four slots only a whole tile uses (the first item of a run or of an image)
      if constexpr (SPLIT && !(kStemSkip & 2))
        if (!cont_s) static_for<12, 16>([&](auto bic) { split_row(bic); });
      if (!cont_l) static_for<12, 16>([&](auto bic) { load_row(bic); });
    }
    if (cont_s) halo_copy(js, tile);
  };
  // row words of a finished item from the pieces staged by the consumers
  auto emit_rows = [&](int j, const uint32_t (*st)[NT + 2]) {
    if constexpr (kStemSkip & 32) return;
    int n, oy0;
    item_of(j, n, oy0);
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
  const int nunits = (NT - wave + CONS_WAVES - 1) / CONS_WAVES;     // 2 (waves 0-5) or 1
  const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
  // byte offset, inside a tile buffer, of dword 0 of the lane's x1 window in tile row 0:
  // pixel pp of the item -> output row pp / 56, column ox; odd columns read copy 1
  auto pixel_addr = [&](uint32_t pp, uint32_t &par) -> uint32_t {
    const uint32_t oyl = pp / 56u, ox = pp - 56u * oyl;
    par = ox & 1u;
    return 4u * (2u * oyl * PITCH + (ox - par) + par * COPY_DW);
  };
  // The two halves of a wave hold consecutive (c,kh) rows R = 2ks, 2ks+1: one tile row apart, except
  // R = 6,7 (the next channel's first row: 15 tile rows on) and R = 20,21 (21 is the shift row: a constant block).
  const uint32_t hrow1 = (uint32_t)h * (PITCH * 4), hrow15 = (uint32_t)h * (15 * PITCH * 4);
  const uint32_t a_shift = (uint32_t)(CONST_DW * 4 - tile_row(20) * (PITCH * 4));             // x1 constants, less k-step 10's immediate
  // B fragments of the first pass by inline assembly: left to the compiler, pairs of these reads are fused
  // into ds_read2_b64, which moves 128 B/clk where ds_read_b64 moves 256, and the reads drift next to their
  // use.  The waits are counted by hand: LDS operations retire in order, so "all but the N youngest" is safe
  // whatever else the compiler has in flight.  v: low / high half of the lane's x1 window.
  auto read_x1 = [&](uint32_t a1, uint32_t a15, uint32_t a0, auto ksc, unsigned long long (&v)[2]) {
    constexpr int ks = decltype(ksc)::value;
    if constexpr (kStemSkip & 8) {
      v[0] = v[1] = (unsigned long long)a1 * (a15 + ks);
      return;
    }
    constexpr int off = tile_row(2 * ks) * (PITCH * 4);
    const uint32_t a = (ks == 3 ? a15 : (ks == 10 ? a0 : a1)) + smem_base;
    asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"
                 : "=&v"(v[0]), "=&v"(v[1])
                 : "v"(a), "n"(off), "n"(off + 8));
  };
  auto frag = [](unsigned long long lo, unsigned long long hi) -> f16x8 {
    const uint4 u = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    return __builtin_bit_cast(f16x8, u);
  };
  // Epilogue of one unit: sign + pack.  C/D layout of the 32x32 MFMA: column = lane&31 (pixel),
  // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (channel within the M-tile).  A lane collects the
  // sign bits of its 16 registers per M-tile (one funnel shift each: w = w<<1 | sign), M-tile 0 in the low
  // half of a word, M-tile 1 in the high half: that word is its pixel's share of the channel words;
  // transposed across each 16-lane group (both halves at once) it becomes, in lane j, register j's bits over
  // the group's 16 pixels, i.e. the row-word pieces.  (A result of exactly -0.0 would count as negative here
  // and as >= 0 in the reference: |pre| = 0 lies in the near-tie band either way.)  The sign collection runs
  // at once (it frees the accumulators); the transpose -- DPP exchanges and one v_permlane16_swap, nothing
  // that waits on the LDS queue -- is cut into five pieces that ride in the issue gaps of the NEXT unit's
  // MFMAs (epi_piece).
  auto sign_bits = [&](const f32x16 &acc) -> uint32_t {
    if constexpr (kStemSkip & 16) return __float_as_uint(acc[0]);
    uint32_t neg = 0;                      // bit r = sign bit of register r
    static_for<0, 16>([&](auto rr) {
      constexpr int r = 15 - decltype(rr)::value;
      neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(acc[r]), 31);
    });
    return ~neg & 0xFFFFu;                 // bit r = (acc[r] >= 0)
  };
  // piece 0..3: butterfly stage of the 16x16 bit transposes; piece 4: exchange between the two 16-lane
  // rows of a half-wave and the store of the row-word pieces (lanes 0-15: channels of half 0, lanes 32-47: half 1)
  auto epi_piece = [&](auto pc, uint32_t &w, int t, uint32_t (*st)[NT + 2], bool live) {
    constexpr int P = decltype(pc)::value;
    if constexpr (kStemSkip & 16) {
      if (P == 4 && live && w == 0x12345u) st[lane][t] = 1;
      return;
    }
    if constexpr (P < 4) {
      constexpr int S[4] = {8, 4, 2, 1};
      const uint32_t partner = lane_xor16<S[P]>(w);
      const uint32_t moved = __builtin_amdgcn_alignbit(partner, partner, tk.rot[P]);
      w = moved ^ ((moved ^ w) & tk.keep[P]);
    } else {
      const auto sw = __builtin_amdgcn_permlane16_swap(w, w, false, false);   // [1]: the value of lane ^ 16, in lanes 0-15 and 32-47
      if (live && (lane & 16) == 0) {
        const int jj = lane & 15, row = (jj & 3) + 8 * (jj >> 2) + 4 * h;
        st[row][t] = (w & 0xFFFFu) | ((uint32_t)sw[1] << 16);
        st[32 + row][t] = (w >> 16) | ((uint32_t)sw[1] & 0xFFFF0000u);
      }
    }
  };

  // ---- pipeline --------------------------------------------------------------------------------
  // Period j: consumers work on item j (tile buffer j&1); producers emit the row words of item
  // j-1, split item j+1 (loaded during period j-1) into the other buffer and issue the loads of
  // item j+2.  Each role runs its own loop with the same my_items + 2 barriers.
  if (producer) {
    if (TT_STEM_PRIO > 0) __builtin_amdgcn_s_setprio(TT_STEM_PRIO);
    STEM_STAMP(1, 0);
    if (my_items > 0) {
      if constexpr (U8) u8_pass(std::false_type{}, 0, 0, tiles);
      else f32_pass(std::false_type{}, 0, 0, tiles);
    }
    __syncthreads();
    STEM_STAMP(1, 1);
    if (my_items > 0) {
      if constexpr (U8) u8_pass(std::true_type{}, 0, 1, tiles);
      else f32_pass(std::true_type{}, 0, 1, tiles);
    }
    STEM_STAMP(1, 2);
    __syncthreads();
    for (int j = 0; j < my_items; ++j) {
      if (j + 1 < my_items) {
        if constexpr (U8) u8_pass(std::true_type{}, j + 1, j + 2, tiles + ((j + 1) & 1) * TILE_DW);
        else f32_pass(std::true_type{}, j + 1, j + 2, tiles + ((j + 1) & 1) * TILE_DW);
      }
      if (j > 0) emit_rows(j - 1, stage[(j - 1) & 1]);
      STEM_STAMP(1, 3 + j);
      __syncthreads();
    }
    if (my_items > 0) emit_rows(my_items - 1, stage[(my_items - 1) & 1]);
    STEM_STAMP(1, 3 + my_items);
    if (out_of_range || (ovf & 0xFFFFu) >= 0x7C00u || (ovf >> 16) >= 0x7C00u) *range_flag = 1u;
  } else {
    if (TT_STEM_PRIO < 0) __builtin_amdgcn_s_setprio(-TT_STEM_PRIO);
    // The w1 fragments of both M-tiles are read from the LDS image two k-steps ahead, like the B fragments (the
    // register file holds three waves per SIMD, 168 registers each: 88 registers of weights do not fit beside the
    // accumulators of two M-tiles).  Lane (i, h) of k-step ks holds channel i (+32), (c,kh) row 2ks + h: 16 bytes of
    // the image, the 32 channels of a half-wave 368 bytes apart (conflict-free).
    uint32_t wt0 = smem_base + (uint32_t)LDS_WTAB + (uint32_t)col * WROW0 + (uint32_t)h * 16u;
    auto read_w1 = [&](auto ksc, u32x4 &va, u32x4 &vb) {
      constexpr int ks = decltype(ksc)::value;
      const uint32_t a = wt0 + 0u;         // (a local: clang rejects a captured variable as an asm operand of a generic lambda)
      asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                   : "=&v"(va), "=&v"(vb)
                   : "v"(a), "n"(ks * 32), "n"(32 * WROW0 + ks * 32));
    };
    // x2 (one copy): three aligned 8-byte reads from the even dword at or below the window, then a select by parity
    auto read_x2 = [&](uint32_t b1, uint32_t b15, uint32_t b0, auto ksc, unsigned long long (&v)[3]) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int off = tile_row(2 * ks) * (PITCH * 4);
      const uint32_t a = (ks == 3 ? b15 : (ks == 10 ? b0 : b1)) + smem_base;
      asm volatile("ds_read_b64 %0, %3 offset:%4\n\tds_read_b64 %1, %3 offset:%5\n\tds_read_b64 %2, %3 offset:%6"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                   : "v"(a), "n"(off), "n"(off + 8), "n"(off + 16));
    };
    auto read_w2 = [&](auto ksc, u32x4 &va, u32x4 &vb) {
      constexpr int ks = decltype(ksc)::value;
      const uint32_t a = wt0 + (uint32_t)WPLANE1 - (uint32_t)col * (WROW0 - WROW1);
      asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                   : "=&v"(va), "=&v"(vb)
                   : "v"(a), "n"(ks * 32), "n"(32 * WROW1 + ks * 32));
    };
    const float thr_abs = init[1] * tau;
    uint16_t *lst = (uint16_t *)(smem + LDS_LIST) + wave * LIST_CAP;     // this wave's list: pixel | channel << 9
    uint32_t n_listed = 0, n_tiles = 0, n_flush = 0;
    STEM_STAMP(0, 0);
    __syncthreads();
    __syncthreads();
    STEM_STAMP(0, 2);
    // The listed outputs of this wave, 16 at a time on the diagonal of one 16x16x32 MFMA tile: lane (j, q) holds, for
    // k-step s, the 8 slots of (c,kh) row R = 4s + q of entry j's channel (A) and of entry j's window (B); rows 22, 23
    // do not exist (zero weights), row 21 is the shift row (constants).  Everything by 4-byte LDS reads of copy 0: no
    // alignment cases.  D[i][i] = the exact pre-activation of entry i (sign only is used).
    auto run_tiles = [&](uint32_t cnt, uint32_t buf, uint32_t (*st)[NT + 2]) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const uint32_t q = (uint32_t)lane >> 4, jl = (uint32_t)lane & 15u;
      const bool diag_lane = q == (jl >> 2);                            // this lane holds D[jl][jl], in register jl & 3
      // byte offsets of this lane's (c,kh) rows R = 4s + q: in the tile (tile_row(R) x row pitch) and in a channel's weights
      uint32_t toff[6], woff[6];
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const uint32_t R = 4u * s + q;
        toff[s] = (R + 14u * ((R * 37u) >> 8)) * (PITCH * 4);
        woff[s] = (R < 22u ? R : 21u) * 16u;
      }
      const bool row_real = q == 0;            // k-step 5: row 20 is a row of pixels, 21 the shift row, 22 and 23 do not exist
      const bool row_live = q < 2;
      const uint32_t rsel = jl & 3u;
      // two tiles (32 entries) per round: two independent MFMA chains hide each other's latency and the LDS reads
      for (uint32_t base = 0; base < cnt; base += 32) {                 // (wave-uniform)
        uint32_t pp[2], ch[2], a[2], wa[2], wb2[2];
        bool valid[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint32_t e = base + 16u * u + jl;
          valid[u] = e < cnt;
          const uint32_t meta = lst[valid[u] ? e : base];
          pp[u] = meta & 511u;
          ch[u] = meta >> 9;
          const uint32_t oyl = pp[u] / 56u, ox = pp[u] - 56u * oyl;
          a[u] = 4u * (2u * oyl * PITCH + ox) + buf;                       // x1 copy 0: dword 0 of the window in tile row 0
          wa[u] = (uint32_t)LDS_WTAB + ch[u] * WROW0;
          wb2[u] = (uint32_t)(LDS_WTAB + WPLANE1) + ch[u] * WROW1;
        }
        const bool two = base + 16u < cnt;                               // (wave-uniform)
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        static_for<0, 6>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          const bool real = s < 5 || row_real;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const uint32_t xa = real ? a[u] + toff[s] : (uint32_t)(CONST_DW * 4);             // shift row (and the absent rows): constants
            const uint32_t xb = real ? a[u] + toff[s] + P1C0 * 4 : (uint32_t)((P1C0 + CONST_DW) * 4);
            const uint32_t *p1 = (const uint32_t *)(smem + xa), *p2 = (const uint32_t *)(smem + xb);
            const f16x8 x1 = __builtin_bit_cast(f16x8, make_uint4(p1[0], p1[1], p1[2], p1[3]));
            const f16x8 x2 = __builtin_bit_cast(f16x8, make_uint4(p2[0], p2[1], p2[2], p2[3]));
            uint4 u1 = *(const uint4 *)(smem + wa[u] + woff[s]), u2 = *(const uint4 *)(smem + wb2[u] + woff[s]);
            if (s == 5 && !row_live) u1 = u2 = make_uint4(0u, 0u, 0u, 0u);
            const f16x8 w1 = __builtin_bit_cast(f16x8, u1), w2 = __builtin_bit_cast(f16x8, u2);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, x1, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x2, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x1, acc[u], 0, 0, 0);
          }
        });
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float fin = acc[u][0];
          fin = rsel == 1u ? acc[u][1] : fin;
          fin = rsel == 2u ? acc[u][2] : fin;
          fin = rsel == 3u ? acc[u][3] : fin;
          if (valid[u] && diag_lane) {
            const uint32_t bit = 1u << (pp[u] & 31u);
            if ((__float_as_uint(fin) >> 31) == 0u) atomicOr(&st[ch[u]][pp[u] >> 5], bit);
            else atomicAnd(&st[ch[u]][pp[u] >> 5], ~bit);
          }
        }
        n_tiles += two ? 2u : 1u;
      }
    };
    // flag word of a unit -> list entries (one per lane and round), diagonal tiles whenever the list is full
    auto drain = [&](uint32_t flags, uint32_t t, uint32_t &cnt, uint32_t buf, uint32_t (*st)[NT + 2]) {
      const uint32_t pp = 32u * t + (uint32_t)col;
      uint32_t f = flags;
      for (;;) {
        const bool has = f != 0u;
        const unsigned long long mask = __ballot(has);
        if (mask == 0ull) break;
        const uint32_t k = (uint32_t)__popcll(mask);
        if (cnt + k > (uint32_t)LIST_CAP) {
          run_tiles(cnt, buf, st);
          cnt = 0;
          n_flush++;
        }
        if (has) {
          const uint32_t r = (uint32_t)__builtin_ctz(f), rr = r & 15u;
          f &= f - 1u;
          const uint32_t ch = 2u * (r & 16u) + (rr & 3u) + 8u * (rr >> 2) + 4u * (uint32_t)h;
          const uint32_t idx = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
          lst[idx] = (uint16_t)(pp | (ch << 9));
        }
        cnt = __builtin_amdgcn_readfirstlane(cnt + k);
        n_listed += k;
      }
    };
    for (int j = 0; j < my_items; ++j) {
      const uint32_t buf = (uint32_t)(j & 1) * (TILE_DW * 4);
      uint32_t (*st)[NT + 2] = stage[j & 1];
      uint32_t flags[2] = {0u, 0u};        // per unit: bit r = register r of M-tile 0, bit 16 + r of M-tile 1 is listed
#if TT_STEM_PRODUCTS != 3
      // fragments are read two k-steps ahead, across unit boundaries
      unsigned long long q0[2], q1[2];
      u32x4 u0a, u0b, u1a, u1b;
      uint32_t par_unused;
      uint32_t a_cur = pixel_addr(32u * (uint32_t)wave + (uint32_t)col, par_unused) + buf;
      read_x1(a_cur + hrow1, a_cur + hrow15, a_cur, std::integral_constant<int, 0>{}, q0);
      read_w1(std::integral_constant<int, 0>{}, u0a, u0b);
      read_x1(a_cur + hrow1, a_cur + hrow15, a_cur, std::integral_constant<int, 1>{}, q1);
      read_w1(std::integral_constant<int, 1>{}, u1a, u1b);
#endif
      uint32_t pend = 0;                   // sign bits of the previous unit (M-tile 0 | M-tile 1 << 16), their transpose in progress
#if TT_STEM_PRODUCTS == 3
      static_for<0, 2>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i < nunits) {
          const int t = wave + CONS_WAVES * i;
          uint32_t par;
          const uint32_t a = pixel_addr(32u * (uint32_t)t + (uint32_t)col, par) + buf;
          const uint32_t b = a - par * (COPY_DW * 4) + P1C0 * 4;
          const uint32_t a1 = a + hrow1, a15 = a + hrow15, a10 = h ? a_shift : a;
          const uint32_t b1 = b + hrow1, b15 = b + hrow15, b10 = h ? (uint32_t)((P1C0 + CONST_DW) * 4 - tile_row(20) * (PITCH * 4)) : b;
          const int tprev = t - CONS_WAVES;
          unsigned long long f[KSTEPS][2], g[KSTEPS][3];
          u32x4 wa[KSTEPS], wb[KSTEPS], wc[KSTEPS], wd[KSTEPS];
          read_x1(a1, a15, a10, std::integral_constant<int, 0>{}, f[0]);
          read_x2(b1, b15, b10, std::integral_constant<int, 0>{}, g[0]);
          read_w1(std::integral_constant<int, 0>{}, wa[0], wb[0]);
          read_w2(std::integral_constant<int, 0>{}, wc[0], wd[0]);
          f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          f32x16 acc1 = acc0;
          static_for<0, KSTEPS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks + 1 < KSTEPS) {
              read_x1(a1, a15, a10, std::integral_constant<int, ks + 1>{}, f[ks + 1]);
              read_x2(b1, b15, b10, std::integral_constant<int, ks + 1>{}, g[ks + 1]);
              read_w1(std::integral_constant<int, ks + 1>{}, wa[ks + 1], wb[ks + 1]);
              read_w2(std::integral_constant<int, ks + 1>{}, wc[ks + 1], wd[ks + 1]);
              asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(f[ks][0]), "+v"(f[ks][1]), "+v"(g[ks][0]), "+v"(g[ks][1]), "+v"(g[ks][2]),
                           "+v"(wa[ks]), "+v"(wb[ks]), "+v"(wc[ks]), "+v"(wd[ks]));
            } else {
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[ks][0]), "+v"(f[ks][1]), "+v"(g[ks][0]), "+v"(g[ks][1]), "+v"(g[ks][2]),
                           "+v"(wa[ks]), "+v"(wb[ks]), "+v"(wc[ks]), "+v"(wd[ks]));
            }
            const f16x8 x1 = frag(f[ks][0], f[ks][1]);
            const uint32_t d0 = (uint32_t)g[ks][0], d1 = (uint32_t)(g[ks][0] >> 32), d2 = (uint32_t)g[ks][1], d3 = (uint32_t)(g[ks][1] >> 32),
                           d4 = (uint32_t)g[ks][2];
            const f16x8 x2 = __builtin_bit_cast(f16x8, par ? make_uint4(d1, d2, d3, d4) : make_uint4(d0, d1, d2, d3));
            const f16x8 w1a = __builtin_bit_cast(f16x8, wa[ks]), w1b = __builtin_bit_cast(f16x8, wb[ks]);
            const f16x8 w2a = __builtin_bit_cast(f16x8, wc[ks]), w2b = __builtin_bit_cast(f16x8, wd[ks]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2a, x1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2b, x1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1a, x2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1b, x2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1a, x1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1b, x1, acc1, 0, 0, 0);
            if constexpr (ks >= 1 && ks <= 5) epi_piece(std::integral_constant<int, ks - 1>{}, pend, tprev, st, i > 0);
            __builtin_amdgcn_sched_barrier(0);
          });
          pend = sign_bits(acc0) | (sign_bits(acc1) << 16);
        }
      });
#else
      static_for<0, 2>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i < nunits) {
          const int t = wave + CONS_WAVES * i, tn = i + 1 < nunits ? t + CONS_WAVES : t;
          const uint32_t a = a_cur, an = pixel_addr(32u * (uint32_t)tn + (uint32_t)col, par_unused) + buf;
          const uint32_t a1 = a + hrow1, a15 = a + hrow15, a10 = h ? a_shift : a, an1 = an + hrow1, an15 = an + hrow15;
          const int tprev = t - CONS_WAVES;
          const bool last_unit = i + 1 >= nunits;             // (wave-uniform; always true for i = 1)
          unsigned long long f[KSTEPS + 2][2];
          u32x4 wa[KSTEPS + 2], wb[KSTEPS + 2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            f[0][e] = q0[e];
            f[1][e] = q1[e];
          }
          wa[0] = u0a;
          wb[0] = u0b;
          wa[1] = u1a;
          wb[1] = u1b;
          f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          f32x16 acc1 = acc0;
          float n2 = 0.f;                    // sum of x1^2 over this half-wave's rows of the window
          static_for<0, KSTEPS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            // the fragments of k-steps ks+1 and ks+2 (8 reads) may still be in flight.  The last unit of an item reads no
            // further (an asynchronous read whose destination the compiler considers dead lands in a register that has
            // been given to something else by then)
            if constexpr (ks + 2 < KSTEPS) {
              read_x1(a1, a15, a10, std::integral_constant<int, ks + 2>{}, f[ks + 2]);
              read_w1(std::integral_constant<int, ks + 2>{}, wa[ks + 2], wb[ks + 2]);
              if constexpr (!(kStemSkip & 8))
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[ks][0]), "+v"(f[ks][1]), "+v"(wa[ks]), "+v"(wb[ks]));
            } else if (!last_unit) {
              read_x1(an1, an15, an, std::integral_constant<int, ks + 2 - KSTEPS>{}, f[ks + 2]);
              read_w1(std::integral_constant<int, ks + 2 - KSTEPS>{}, wa[ks + 2], wb[ks + 2]);
              if constexpr (!(kStemSkip & 8))
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[ks][0]), "+v"(f[ks][1]), "+v"(wa[ks]), "+v"(wb[ks]));
            } else {
              f[ks + 2][0] = f[ks + 2][1] = 0ull;
              wa[ks + 2] = wb[ks + 2] = u32x4{0u, 0u, 0u, 0u};
              if constexpr (!(kStemSkip & 8)) {
                if constexpr (ks + 1 < KSTEPS) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[ks][0]), "+v"(f[ks][1]), "+v"(wa[ks]), "+v"(wb[ks]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[ks][0]), "+v"(f[ks][1]), "+v"(wa[ks]), "+v"(wb[ks]));
              }
            }
            const f16x8 x1 = frag(f[ks][0], f[ks][1]);
            if constexpr (kStemSkip & 4) {
              acc0[0] += (float)x1[0] + (float)x1[5];
            } else {
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[ks]), x1, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wb[ks]), x1, acc1, 0, 0, 0);
            }
            if constexpr (!(kStemSkip & 128)) {
              // N^2: the shift row (k-step 10, upper half-wave) holds constants, not pixels
              const uint32_t keep = (ks == 10 && h) ? 0u : 0xFFFFFFFFu;
              const uint32_t d[4] = {(uint32_t)f[ks][0], (uint32_t)(f[ks][0] >> 32), (uint32_t)f[ks][1], (uint32_t)(f[ks][1] >> 32)};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const f16x2 v = __builtin_bit_cast(f16x2, ks == 10 ? (d[e] & keep) : d[e]);
                n2 = __builtin_amdgcn_fdot2(v, v, n2, false);
              }
            }
            if constexpr (ks >= 1 && ks <= 5) epi_piece(std::integral_constant<int, ks - 1>{}, pend, tprev, st, i > 0);
            __builtin_amdgcn_sched_barrier(0);     // reads stay two k-steps ahead of their MFMAs, no further
          });
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            q0[e] = f[KSTEPS][e];
            q1[e] = f[KSTEPS + 1][e];
          }
          u0a = wa[KSTEPS];
          u0b = wb[KSTEPS];
          u1a = wa[KSTEPS + 1];
          u1b = wb[KSTEPS + 1];
          if constexpr (!(kStemSkip & 128)) {
            // threshold of this pixel: V <= 1 for every channel (host), so thr = |x1 window| (+ the subnormal constant);
            // v_sqrt_f32 is good to 1 ulp, far inside the 2^-9 slack of V
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(n2), __float_as_uint(n2), false, false);
            const float thr = __builtin_amdgcn_sqrtf(__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * tau + thr_abs;
            if constexpr (!(kStemSkip & 512)) {
              uint32_t fl = 0;                 // bit = sign of |acc| - thr: this output is listed
              static_for<0, 16>([&](auto rr) {
                constexpr int r = 15 - decltype(rr)::value;
                fl = __builtin_amdgcn_alignbit(fl, __float_as_uint(__builtin_fabsf(acc1[r]) - thr), 31);
              });
              static_for<0, 16>([&](auto rr) {
                constexpr int r = 15 - decltype(rr)::value;
                fl = __builtin_amdgcn_alignbit(fl, __float_as_uint(__builtin_fabsf(acc0[r]) - thr), 31);
              });
              flags[i] = fl;
            } else if (thr == 123.456f) {
              acc0[0] = 1.f;
            }
          }
          pend = sign_bits(acc0) | (sign_bits(acc1) << 16);
          a_cur = an;
        }
      });
#endif
      // the last unit's transpose has no MFMAs to hide under
      static_for<0, 5>([&](auto pc) { epi_piece(pc, pend, wave + CONS_WAVES * (nunits - 1), st, true); });
      if constexpr (!(kStemSkip & 256) && TT_STEM_PRODUCTS != 3) {
        uint32_t cnt = 0;                  // entries on this wave's list (wave-uniform)
        drain(flags[0], (uint32_t)wave, cnt, buf, st);
        if (nunits > 1) drain(flags[1], (uint32_t)(wave + CONS_WAVES), cnt, buf, st);
        if (cnt) run_tiles(cnt, buf, st);
      }
      STEM_STAMP(0, 3 + j);
      __syncthreads();
    }
    if (stats && lane == 0) {
      atomicAdd(&stats[0], n_listed);
      atomicAdd(&stats[1], n_tiles);
      atomicAdd(&stats[2], n_flush);
    }
  }
}

}  // namespace

// Host side of the operand split: w [64][3][7][7] float32 -> the weight image the kernel keeps in LDS,
// [plane][channel][(c,kh) row R = c*7 + kh, 0..21][8 slots] fp16 (WROW0 / WROW1 bytes per channel), slot = kw + 1;
// slot 0 carries a zero weight, row 21 the BatchNorm shift.
static uint16_t f32_to_f16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  u &= 0x7FFFFFFFu;
  if (u >= 0x47800000u) return sign | 0x7C00u;                    // >= 2^16 (never: scaled far below)
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

// v -> h1 + h2 (two fp16 terms, 22 bits), from the float64 value
static void split2(double v, uint16_t &h1, uint16_t &h2) {
  h1 = f32_to_f16_rne((float)v);
  h2 = f32_to_f16_rne((float)(v - (double)f16_to_f32(h1)));
}

// The weights of output channel ch are  w * bn_scale * g[ch]  with g > 0 chosen so that the first-pass error
// bound of the kernel is  |acc - exact| <= V N,  V = || |w2| + (2^-11 + 2^-17) |w1| ||_2 <= 1 - 2^-9  (N: the
// 2-norm of the x1 window): only the sign of a channel's result is used, so a positive factor per channel is free,
// and it makes the kernel's threshold independent of the channel.  init[0] = the constant c of the shift row (a
// power of two), init[1] = the additive part of the threshold (fp16 subnormals of x1 / x2: rows where the relative
// bound |x2| <= 2^-11 |x1| does not hold).  Returns false
// if the folded BatchNorm shift is too large for the shift row.
bool stem_split_weights(const float *w, const double *scale, const double *shift, int p, uint16_t *out, float *init) {
  constexpr double REL = 1.0 / 2048.0 + 1.0 / 131072.0;       // |x2| <= 2^-11 |x1|; 2^-17: the float32 roundings of both MFMA chains
  std::vector<double> wd((size_t)64 * 147, 0.0), g(64, 1.0), sh(64, -1.0);
  double thr_abs = 0.0, amax = 0.0;
  for (int ch = 0; ch < p; ++ch) {
    double nrm = 0.0;
    for (int i = 0; i < 147; ++i) {
      wd[(size_t)ch * 147 + i] = (double)w[(size_t)ch * 147 + i] * scale[ch];
      nrm += wd[(size_t)ch * 147 + i] * wd[(size_t)ch * 147 + i];
    }
    nrm = std::sqrt(nrm);
    if (!(nrm > 0.0) || !std::isfinite(nrm)) {                // an all-zero (or broken) filter: the result is the shift alone
      for (int i = 0; i < 147; ++i) wd[(size_t)ch * 147 + i] = 0.0;
      g[ch] = 1.0;
    } else {
      auto bound = [&](double gg, double &l1w, double &l1v) {
        double v2 = 0.0;
        l1w = l1v = 0.0;
        for (int i = 0; i < 147; ++i) {
          uint16_t h1, h2;
          split2(wd[(size_t)ch * 147 + i] * gg, h1, h2);
          const double a1 = std::fabs((double)f16_to_f32(h1)), v = std::fabs((double)f16_to_f32(h2)) + REL * a1;
          v2 += v * v;
          l1w += a1;
          l1v += v;
        }
        return std::sqrt(v2);
      };
      double gg = 2048.0 / nrm, l1w = 0.0, l1v = 0.0, V = 0.0;
      for (int it = 0; it < 8; ++it) {                        // V is (nearly) linear in g: two or three steps settle it
        V = bound(gg, l1w, l1v);
        if (V <= 1.0 - 1.0 / 512.0 && V >= 0.95) break;
        gg *= 0.975 / V;
      }
      while (V > 1.0 - 1.0 / 512.0) {
        gg *= 0.95;
        V = bound(gg, l1w, l1v);
      }
      g[ch] = gg;
      // x1 subnormal or zero (|16 x| < 2^-14): |x2| <= 2^-24 and |x1| < 2^-14 there, instead of the relative bound
      thr_abs = std::max(thr_abs, l1w * (1.0 / 16777216.0) + l1v * (1.0 / 16384.0));
    }
    sh[ch] = shift[ch] * g[ch] * (double)X_PRESCALE;           // the accumulator's unit: g x 16
    amax = std::max(amax, std::fabs(sh[ch]));
  }
  if (!std::isfinite(amax)) return false;
  double c = 1.0;
  while (amax / c > 16384.0 && c < 32768.0) c *= 2.0;
  if (amax / c > 32768.0) return false;
  for (int i = 0; i < 64; ++i) init[i] = 0.f;
  init[0] = (float)c;
  init[1] = std::max((float)(thr_abs * 1.001), 1e-6f);
  memset(out, 0, (size_t)WTAB_BYTES);
  for (int ch = 0; ch < 64; ++ch)
    for (int R = 0; R < 22; ++R) {
      double resid = sh[ch] / c;                               // shift row: three fp16 terms of shift / c in the w1 plane
      for (int j = 0; j < 8; ++j) {
        const int kw = j - 1;
        uint16_t parts[NPL] = {0, 0};
        if (R < 21) {
          if (ch < p && kw >= 0) split2(wd[(size_t)ch * 147 + R * 7 + kw] * g[ch], parts[0], parts[1]);
        } else if (j < 3) {
          parts[0] = f32_to_f16_rne((float)resid);
          resid -= (double)f16_to_f32(parts[0]);
        }
        out[(size_t)ch * (WROW0 / 2) + R * 8 + j] = parts[0];
        out[(size_t)(WPLANE1 / 2) + (size_t)ch * (WROW1 / 2) + R * 8 + j] = parts[1];
      }
    }
  return true;
}

size_t stem_split_weights_elems() { return (size_t)WTAB_BYTES / 2; }

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

// (x, wfrag, init, rp, p, n, norm_tab, range_flag, stats, tau): keep in step with the kernel's signature
int stem_kernel_arg_sizes(const int **sizes) {
  static const int kSizes[10] = {8, 8, 8, 8, 4, 4, 8, 8, 8, 4};
  *sizes = kSizes;
  return 10;
}

// TTNET_STEM_EXACT=1: every output through the exact chain (the reference mode the listed path is tested against;
// hundreds of times slower).  TTNET_STEM_TAU_SCALE=<f >= 1>: multiplies the threshold of the listed path (tests: more
// outputs listed, lists that fill up).  Both are read at every launch (a captured graph keeps what it was captured with).
static float stem_tau_scale() {
  const char *x = getenv("TTNET_STEM_EXACT");
  if (x && x[0] == '1') return 1e30f;
  const char *e = getenv("TTNET_STEM_TAU_SCALE");
  const float f = e ? (float)atof(e) : 1.0f;
  return f >= 1.0f ? f : 1.0f;              // never below the proven bound
}

int launch_stem(const void *x, bool x_is_u8, const uint32_t *norm_tab, const void *wfrag, const float *init, uint64_t *rp,
                uint16_t *cp, int n, int p, uint32_t *range_flag, uint32_t *stats, hipStream_t s) {
  const float tau = stem_tau_scale();
  if (p < 1 || p > 64 || (cp && p != 64)) {
    set_error("stem: p=%d outside [1,64] (channel words need p = 64)", p);
    return TTNET_E_UNSUPPORTED;
  }
  if (((uintptr_t)x & (x_is_u8 ? 3 : 15)) != 0) {
    set_error("stem: the input must be %d-byte aligned", x_is_u8 ? 4 : 16);
    return TTNET_E_INVALID;
  }
  const size_t lds = (size_t)(x_is_u8 ? LDS_END_U8 : LDS_END_F32);
  const int items = n * NBLK;
  const int grid = std::min(items, 256);
  auto launch = [&](auto kernel) -> int {
    TT_TRY(ensure_dynamic_lds((const void *)kernel, lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(STEM_THREADS), lds, s, x, (const uint4 *)wfrag, init, rp, p, n, norm_tab, range_flag,
                       stats, tau);
    return TTNET_OK;
  };
  if (x_is_u8) TT_TRY(launch(stem_pc_kernel<true>));
  else TT_TRY(launch(stem_pc_kernel<false>));
  TT_HIP(hipGetLastError());
  // the channel-word layout (two-launch gate kernels only) from the rows
  if (cp) TT_TRY(launch_rp_to_cp(rp, cp, n, 64, 56, 56, s));
  return TTNET_OK;
}

}  // namespace ttnet
