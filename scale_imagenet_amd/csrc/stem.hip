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
// Round 2 (the in-kernel stamps of tools/ubench/stem_parts.hip): a period of the round-1 kernel was
// max(producers, consumers) with both near 14 k cycles -- the producers waiting for HBM (113 KB per
// item), the consumers ADDING their parts (MFMA 7.4 k + fragment reads 2.8 k + loop overhead 2.5 k
// + epilogue 1.7 k) because the two waves of a SIMD ran the same program in lockstep and eight
// 4-byte LDS reads fed every three MFMAs.  This version:
//  * the weight fragments of a wave's M-tile live in its registers for the whole kernel (88 VGPRs),
//    not in LDS: no A-fragment reads, 44 KiB of LDS freed;
//  * the freed LDS holds a second copy of every tile row, one dword to the left, so that the window
//    of an odd pixel is 8-byte aligned too: a B fragment is four ds_read_b64, conflict-free (row
//    pitch 60 dwords: the step to the next output row is 2*60 - 56 = 64 banks; the copies sit 32
//    banks apart), instead of eight ds_read_b32;
//  * a wave finishes one unit (N-tile x its M-tile) at a time, 33 MFMAs on one accumulator, its
//    fragments read two k-steps ahead; the sign/pack epilogue of a unit then runs beside the other
//    wave's MFMAs;
//  * the BatchNorm shift is one more k-row (a block of constants in LDS, its weights shift / c): the
//    accumulators start from an inline zero;
//  * a producer lane pools two pixels from two 16-byte buffer loads (row offset in an SGPR, padding by
//    the bounds check) and writes packed dwords; a row's registers are refilled as soon as it is split;
//  * a workgroup walks a run of consecutive row blocks and copies the five tile rows two neighbours
//    share inside LDS: no input byte is read twice.
//
// Bound: 16-bit MFMA (2.5 PFLOP/s dense) at 3 MFMA flops per algorithmic flop (3.6 with the slot / row
// padding) at the clock the chip holds under this load, beside the HBM stream of the float32 input;
// 29.5 MMAC/image.  The kernel is power-limited: all-zero input runs the same cycles at 1.93 instead of
// 1.52 GHz.

#include <math.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "ttnet_common.h"

namespace ttnet {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SR = 8;                   // output rows per item
constexpr int NBLK = 56 / SR;           // row blocks (items) per image
constexpr int TR = 2 * SR + 5;          // pooled rows in the tile
constexpr int PITCH = 60;               // dwords per tile row: 120 fp16 = pooled columns -4 .. 115
constexpr int LROWS = 64;                // rows a copy holds: the tile's 63 and a spare one (a producer wave owns 16)
constexpr int COPY_DW = LROWS * PITCH + 32;  // one copy of a plane; the +32 puts copy 1 thirty-two banks from copy 0
constexpr int NPL = SPLIT_PLANES;       // fp16 planes per operand
constexpr int PLANE_DW = 2 * COPY_DW;   // [copy 0: dword j = pixels (2j, 2j+1)][copy 1: dword j = copy 0's dword j+1]
constexpr int TILE_DW = NPL * PLANE_DW; // dwords per tile buffer (60,928 B)
constexpr int KSTEPS = 11;              // 22 (c,kh) rows (21 + one repeated with zero weights), two per MFMA
constexpr int NT = SR * 56 / 32;        // 14 N-tiles of 32 pixels
constexpr float X_PRESCALE = ACT_PRESCALE;

// Diagnostic builds only (tools/ubench/stem_parts.hip): parts of the kernel switched off, in-kernel stamps.
#ifndef TT_STEM_SKIP
#define TT_STEM_SKIP 0
#endif
#ifndef TT_STEM_LOAD_AUX
#define TT_STEM_LOAD_AUX 2        /* cache policy of the input loads: 0 default, 2 nt (round 3: the input is read once, and left to
                                     the default policy its 154 MB per batch push tables, weights and the float table out of the Infinity Cache:
                                     the FORWARD is 5 - 7 % faster with nt, the stem itself unchanged; same-box A/B in profiles/r03_cache_policy.txt) */
#endif
#ifndef TT_STEM_PKCVT
#define TT_STEM_PKCVT 0
#endif
#ifndef TT_STEM_PRIO
#define TT_STEM_PRIO 2
#endif
constexpr int kStemSkip = TT_STEM_SKIP;   // 1 no global loads, 2 no split, 4 no MFMA, 8 no fragment reads, 16 no epilogue, 32 no row words, 64 no tile stores
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
static_assert(CONS_WAVES % 4 == 0, "a consumer wave keeps one M-tile: unit u = wave + CONS_WAVES * i has M-tile u % MT = wave % MT, MT = 1, 2 or 4");
constexpr int CONST_DW = LROWS * PITCH; // the constant block of the BatchNorm-shift row (16 bytes per plane, buffer 0)

// tile row of (c,kh) row R = c*7 + kh (for the lane's output row 0)
constexpr int tile_row(int R) { return (R / 7) * TR + R % 7; }

// Persistent producer / consumer kernel.  One workgroup per CU walks items (image, block of SR
// output rows).  Producer waves stream the raw rows from HBM, pool them and write the two fp16
// planes (two copies each) of the NEXT item's tile into the other half of an LDS double buffer;
// consumer waves run the MFMAs and the sign/pack epilogue of the CURRENT item.  One workgroup
// barrier per item.  MT = M-tiles of 32 output channels (1, 2 or 4: p <= 32, 64, 128).  Consumer wave w owns units
// w, w+8, ...: all of one M-tile (u % MT = w % MT); at MT = 2 waves w and w+4 share a SIMD and carry 4 + 3 units.  BatchNorm is folded: its scale into the
// weights (host), its shift into the initial value of the accumulators, so the epilogue is the
// sign bit alone.
//
// U8 = true: the input is the decoder's uint8 HWC image and the last two steps of
// the input pipeline, ToTensor (/255) and Normalize(mean, std) (utils/preprocess.py:104-108),
// are fused in front of the average pool.  Round 3: TWO products instead of three.  The four bytes of a
// pooled pixel and channel are summed as integers (v_dot4 with a byte selector), s in 0..1020, and the
// tile holds s - s_c (s_c = round(1020 mean_c)) as ONE fp16 plane -- an integer below 2048 is exact in
// fp16 -- while the normalisation moves into the weights and a bias:
//   pre = sum_valid w ((s / 1020 - mean_c) / std_c) + shift
//       = sum_all (w / (1020 std_c)) (s - s_c)  +  [shift + sum_all w d_c]  -  sum_padded w d_c ,   d_c = (s_c - 1020 mean_c) / (1020 std_c)
// (a padded tap holds s - s_c = 0 in the tile; |d_c| <= 2.2e-3).  The first term is w2 x + w1 x on the
// matrix cores, the second the 22nd k-row as for float32 input, the third a per-(border class, channel)
// correction added to the accumulators before the sign: 16 classes (output row 0 / 1 / interior / 55 x
// the same for the column: which taps fall into the padding), a table in LDS.  22 instead of 33 matrix
// instructions per unit, half the fragment reads, one plane written by the producers -- the kernel is
// bound by the energy of its matrix instructions.  Numerically the same quantity as the
// float32 path computes from the normalised tensor, to the same ~1e-6; bits are oracle-checked outside
// the near-tie band (test_uint8_input_fused_normalise).
// CP = also emit the channel-word layout (read only by the two-launch gate kernels of gate.hip: --layers 3 / 4,
// x-small, TTNET_GATE_UNFUSED); the block-fused gate path reads rows alone, and the word formation and its
// cross-lane exchange are then compiled out of the epilogue.
template <bool U8, bool CP, int MT, int AUX = TT_STEM_LOAD_AUX>
__global__ __launch_bounds__(STEM_THREADS) void stem_pc_kernel(const void *__restrict__ xin, const uint4 *__restrict__ wfrag,
                                                               const float *__restrict__ init, uint64_t *__restrict__ rp,
                                                               uint16_t *__restrict__ cp, int p, int n_images,
                                                               const uint32_t *__restrict__ norm_tab, uint32_t *range_flag) {
  const float *x = (const float *)xin;
  const uint8_t *xu8 = (const uint8_t *)xin;
  extern __shared__ __align__(16) uint8_t smem[];
  uint32_t *tiles = (uint32_t *)smem;                                              // [2][TILE_DW]
  static_assert(!CP || MT == 2, "the channel-word layout is built for p = 64");
  constexpr int CH = 32 * MT, UNITS = NT * MT;                                     // channels the kernel carries; (N-tile, M-tile) pairs of one item
  uint32_t(*stage)[CH][NT + 2] = (uint32_t(*)[CH][NT + 2])(smem + 2 * TILE_DW * 4);   // [2][CH][NT+2]
  // U8: norm_tab = [4] centres s_c (int32; the 4th unused), then the border corrections [16 classes][MT][2 halves][16 registers] float32
  constexpr uint32_t CORR_OFF = 2 * TILE_DW * 4 + 2 * CH * (NT + 2) * 4;
  float *s_corr = (float *)(smem + CORR_OFF);
  int s_c[3] = {0, 0, 0};
  if constexpr (U8) {
    for (int i = threadIdx.x; i < 16 * CH; i += STEM_THREADS) s_corr[i] = __uint_as_float(norm_tab[4 + i]);
#pragma unroll
    for (int c = 0; c < 3; ++c) s_c[c] = (int)__builtin_amdgcn_readfirstlane(norm_tab[c]);
  }
  const int H = 224, W = 224;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wave >= CONS_WAVES;
  // The folded BatchNorm shift enters the GEMM as one more k-row: the 22nd (c,kh) row reads a block of
  // constants (slots 0 and 1 = c, a power of two; plane 1 = 0) instead of pixels, and its weights are
  // shift / c split over the two slots (stem_split_weights), so the accumulators start from an inline
  // zero and a unit needs no start values from LDS.  The block sits in the padding behind copy 0 of
  // buffer 0, which no tile row reaches (only the very last padding dword is ever scribbled on).
  if (threadIdx.x < 8) {
    const uint32_t cb = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)init[0]);
    tiles[(threadIdx.x >> 2) * PLANE_DW + CONST_DW + (threadIdx.x & 3)] = threadIdx.x == 0 ? (cb | (cb << 16)) : 0u;
  }
  for (int i = threadIdx.x; i < 2 * CH; i += STEM_THREADS) {
    stage[i / CH][i % CH][NT] = 0;
    stage[i / CH][i % CH][NT + 1] = 0;
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
  // rows 16..20 of the previous tile are rows 0..4 of this one: 3 channels x 5 rows x (2 planes x 2 copies),
  // 240 bytes each, as 16-byte pieces, four row arrays per wave instruction
  auto halo_copy = [&](int js, uint32_t *tile) {
    const uint32_t *prev = tiles + ((js - 1) & 1) * TILE_DW;
    const int piece = lane % 15, which = lane / 15;            // lanes 60-63 idle
    if (which < 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int id = 16 * pw + 4 * k + which;                 // 0..63, 60 used: (array, channel, row)
        if (id < 60) {
          const int arr = id / 15, cr5 = id - 15 * arr, c = cr5 / 5, rr = cr5 - 5 * c;
          const int base = (arr >> 1) * PLANE_DW + (arr & 1) * COPY_DW + (c * TR + rr) * PITCH + 4 * piece;
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
            // s - s_c of both pixels as fp16 (exact: |s - s_c| <= 1020), zero in the padding; one plane
            const float f0 = (float)((int)s0[c] - s_c[c]), f1 = (float)((int)s1[c] - s_c[c]);
            const uint32_t d1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(f0, f1)) & keep;
            uint32_t *dst = tile + (c * TR + r) * PITCH + lane;
            dst[0] = d1;
            dst[COPY_DW - 1] = d1;             // copy 1, one dword to the left (lane 0 lands in unused padding)
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
        qa[bi] = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)lane_off8, soff, AUX);
        qb[bi] = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)lane_off8 + W * 3, soff, AUX);
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
        dst[PLANE_DW] = d2;
        dst[PLANE_DW + COPY_DW - 1] = d2;
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
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, soff, AUX);
          const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off + W * 4, soff, AUX);
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
    for (int idx = threadIdx.x - 64 * CONS_WAVES; idx < CH * SR; idx += 64 * PROD_WAVES) {
      const int ch = idx % CH, row = idx / CH;
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
  const int m = wave % MT;                               // this wave's M-tile (consumers only)
  constexpr int TSTEP = CONS_WAVES / MT;                 // N-tiles between two units of a wave
  const int nunits = (UNITS - wave + CONS_WAVES - 1) / CONS_WAVES;     // MT = 2: 4 (waves 0-3) or 3
  // byte offset, inside a tile buffer, of dword 0 of the lane's window in plane 0 and tile row 0:
  // pixel pp of the item -> output row pp / 56, column ox; odd columns read copy 1
  auto unit_addr = [&](int i) -> uint32_t {
    const uint32_t t = (uint32_t)(wave / MT) + (uint32_t)TSTEP * (uint32_t)i;
    const uint32_t pp = 32u * t + (uint32_t)col, oyl = pp / 56u, ox = pp - 56u * oyl, par = ox & 1u;
    return 4u * (2u * oyl * PITCH + (ox - par) + par * COPY_DW);
  };
  // The two halves of a wave hold consecutive (c,kh) rows R = 2ks, 2ks+1: one tile row apart, except
  // R = 6,7 (the next channel's first row: 15 tile rows on) and R = 20,21 (21 is the zero-weight pad: row 20 again).
  const uint32_t hrow1 = (uint32_t)h * (PITCH * 4), hrow15 = (uint32_t)h * (15 * PITCH * 4);
  // B fragments by inline assembly: left to the compiler, pairs of these reads (the two halves of a
  // window, or two k-steps off one base) are fused into ds_read2_b64, which moves 128 B/clk where
  // ds_read_b64 moves 256, and the reads drift next to their use.  The waits are counted by hand:
  // LDS operations retire in order, so "all but the N youngest" is safe whatever else the compiler has
  // in flight.  v: plane 0 low / high half, plane 1 low / high half of the lane's window.
  auto read_frag = [&](uint32_t a1, uint32_t a15, uint32_t a0, auto ksc, unsigned long long (&v)[4]) {
    constexpr int ks = decltype(ksc)::value;
    if constexpr (kStemSkip & 8) {
      v[0] = v[1] = v[2] = v[3] = (unsigned long long)a1 * (a15 + ks);
      return;
    }
    constexpr int off = tile_row(2 * ks) * (PITCH * 4);
    const uint32_t a = (ks == 3 ? a15 : (ks == 10 ? a0 : a1)) + (uint32_t)(uintptr_t)smem;
    if constexpr (U8) {                  // one plane: two reads, v[2] and v[3] stay unused
      asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=&v"(v[0]), "=&v"(v[1]) : "v"(a), "n"(off), "n"(off + 8));
      return;
    }
    asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%6\n\t"
                 "ds_read_b64 %2, %4 offset:%7\n\tds_read_b64 %3, %4 offset:%8"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(a), "n"(off), "n"(off + 8), "n"(off + PLANE_DW * 4), "n"(off + PLANE_DW * 4 + 8));
  };
  auto frag = [](unsigned long long lo, unsigned long long hi) -> f16x8 {
    const uint4 u = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    return __builtin_bit_cast(f16x8, u);
  };
  // Epilogue of one unit: sign + pack.  C/D layout of the 32x32 MFMA: column = lane&31 (pixel),
  // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (channel within the M-tile).  A lane collects the
  // sign bits of its 16 registers (one funnel shift each: w = w<<1 | sign): that word is its
  // pixel's share of the channel words; transposed across each 16-lane group it becomes, in lane
  // j, register j's bits over the group's 16 pixels, i.e. the row-word pieces.  (A result of
  // exactly -0.0 would count as negative here and as >= 0 in the reference: |pre| = 0 lies in
  // the near-tie band either way.)  The sign collection runs at once (it frees the accumulator);
  // the transpose -- DPP exchanges and one v_permlane16_swap, nothing that waits on the LDS queue --
  // is cut into five pieces that ride in the issue gaps of the NEXT unit's MFMAs (epi_piece).
  auto sign_bits = [&](const f32x16 &acc) -> uint32_t {
    if constexpr (kStemSkip & 16) return __float_as_uint(acc[0]);
    uint32_t neg = 0;                      // bit r = sign bit of register r
    static_for<0, 16>([&](auto rr) {
      constexpr int r = 15 - decltype(rr)::value;
      neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(acc[r]), 31);
    });
    return ~neg & 0xFFFFu;                 // bit r = (acc[r] >= 0)
  };
  auto channel_words = [&](uint32_t bits, int t, int n, int oy0) {
    // channel word bit of register r: (r&3) + 8*((r>>2)&1) + 4*h within the 16-channel group r>>3
    const uint32_t cw0 = bits & 0xFFu, cw1 = bits >> 8;
    uint32_t pw2 = ((cw0 & 15u) | ((cw0 & 0xF0u) << 4)) | (((cw1 & 15u) | ((cw1 & 0xF0u) << 4)) << 16);
    pw2 <<= 4 * h;
    pw2 |= (uint32_t)__shfl_xor((int)pw2, 32);
    const int q = 2 * m + h;             // half-wave 0 stores group 2m, half-wave 1 group 2m+1
    const int pp = 32 * t + col, oyl = pp / 56, ox = pp - 56 * oyl;
    cp[((size_t)n * 4 + q) * (56 * 56) + (oy0 + oyl) * 56 + ox] = (uint16_t)(h ? (pw2 >> 16) : pw2);
  };
  // piece 0..3: butterfly stage of the 16x16 bit transpose; piece 4: exchange between the two 16-lane
  // rows of a half-wave and the store of the row-word piece (lanes 0-15: channels of half 0, lanes 32-47: half 1)
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
      const uint32_t piece = w & 0xFFFFu;
      const auto sw = __builtin_amdgcn_permlane16_swap(piece, piece, false, false);   // [1]: the value of lane ^ 16, in lanes 0-15 and 32-47
      if (live && (lane & 16) == 0) {
        const int jj = lane & 15;
        st[m * 32 + (jj & 3) + 8 * (jj >> 2) + 4 * h][t] = piece | ((uint32_t)sw[1] << 16);
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
    // this wave's weight fragments, [ks][plane][mtile][lane] x 16 bytes in global memory (L2)
    uint4 wreg[KSTEPS][NPL];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) wreg[ks][pl] = wfrag[((ks * NPL + pl) * MT + m) * 64 + lane];
    STEM_STAMP(0, 0);
    __syncthreads();
    __syncthreads();
    STEM_STAMP(0, 2);
    // address of the constant block for the half-wave that holds the shift row (k-step 10, h = 1),
    // less that k-step's row offset, which read_frag adds as an immediate
    const uint32_t a_shift = (uint32_t)(CONST_DW * 4 - tile_row(20) * (PITCH * 4));
    for (int j = 0; j < my_items; ++j) {
      int n, oy0;
      item_of(j, n, oy0);
      const uint32_t buf = (uint32_t)(j & 1) * (TILE_DW * 4);
      uint32_t (*st)[NT + 2] = stage[j & 1];
      // fragments are read two k-steps ahead, across unit boundaries
      unsigned long long q0[4] = {0, 0, 0, 0}, q1[4] = {0, 0, 0, 0};
      {
        const uint32_t a = unit_addr(0) + buf;
        read_frag(a + hrow1, a + hrow15, a, std::integral_constant<int, 0>{}, q0);
        read_frag(a + hrow1, a + hrow15, a, std::integral_constant<int, 1>{}, q1);
      }
      uint32_t pend = 0;                   // sign bits of the previous unit, its transpose in progress
#pragma unroll 1
      for (int i = 0; i < nunits; ++i) {
        const uint32_t a = unit_addr(i) + buf, an = unit_addr(i + 1 < nunits ? i + 1 : i) + buf;
        const uint32_t a1 = a + hrow1, a15 = a + hrow15, a10 = h ? a_shift : a, an1 = an + hrow1, an15 = an + hrow15;
        const int tprev = wave / MT + TSTEP * (i - 1);
        unsigned long long f[KSTEPS + 2][4];
#pragma unroll
        for (int e = 0; e < (U8 ? 2 : 4); ++e) {
          f[0][e] = q0[e];
          f[1][e] = q1[e];
        }
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 cr[4] = {};                    // U8: the border correction of the lane's pixel and sixteen channels
        static_for<0, KSTEPS>([&](auto ksc) {
          constexpr int ks = decltype(ksc)::value;
          if constexpr (U8 && ks == KSTEPS - 3) {
            // the pixel's border class (which taps of its window lie in the padding): row 0 / 1 / interior / 55, column likewise
            const uint32_t tt = (uint32_t)(wave / MT) + (uint32_t)TSTEP * (uint32_t)i;
            const uint32_t pp = 32u * tt + (uint32_t)col, oyl = pp / 56u, ox = pp - 56u * oyl, oy = (uint32_t)oy0 + oyl;
            const uint32_t yc = oy == 0u ? 0u : (oy == 1u ? 1u : (oy == 55u ? 3u : 2u));
            const uint32_t xc = ox == 0u ? 0u : (ox == 1u ? 1u : (ox == 55u ? 3u : 2u));
            const uint32_t ca = (uint32_t)(uintptr_t)smem + CORR_OFF + ((((yc * 4u + xc) * MT + (uint32_t)m) * 2u + (uint32_t)h) * 64u);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                         : "=&v"(cr[0]), "=&v"(cr[1]), "=&v"(cr[2]), "=&v"(cr[3]) : "v"(ca));
          }
          if constexpr (ks + 2 < KSTEPS) read_frag(a1, a15, a10, std::integral_constant<int, ks + 2>{}, f[ks + 2]);
          else read_frag(an1, an15, an, std::integral_constant<int, ks + 2 - KSTEPS>{}, f[ks + 2]);
          const f16x8 w1 = __builtin_bit_cast(f16x8, wreg[ks][0]), w2 = __builtin_bit_cast(f16x8, wreg[ks][1]);
          if constexpr (U8) {
            // LDS operations retire in order; younger than f[ks]: the fragments of k-steps ks+1 and ks+2 (4 reads), and between
            // k-steps KSTEPS-3 and KSTEPS-2 also the four correction reads issued in front of f[KSTEPS-1]
            if constexpr (ks == KSTEPS - 3 || ks == KSTEPS - 2) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[ks][0]), "+v"(f[ks][1]));
            else if constexpr (ks == KSTEPS - 1)
              asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[ks][0]), "+v"(f[ks][1]), "+v"(cr[0]), "+v"(cr[1]), "+v"(cr[2]), "+v"(cr[3]));
            else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[ks][0]), "+v"(f[ks][1]));
            const f16x8 x1 = frag(f[ks][0], f[ks][1]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, x1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x1, acc, 0, 0, 0);
          } else {
            // the fragments of k-steps ks+1 and ks+2 (8 reads) may still be in flight
            if constexpr (!(kStemSkip & 8))
              asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[ks][0]), "+v"(f[ks][1]), "+v"(f[ks][2]), "+v"(f[ks][3]));
            const f16x8 x1 = frag(f[ks][0], f[ks][1]), x2 = frag(f[ks][2], f[ks][3]);
            if constexpr (kStemSkip & 4) {
              acc[0] += (float)x1[0] + (float)x2[1] + (float)w1[2] + (float)w2[3];
            } else {
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, x1, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x2, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x1, acc, 0, 0, 0);
            }
          }
          if constexpr (ks >= 1 && ks <= 5) epi_piece(std::integral_constant<int, ks - 1>{}, pend, tprev, st, i > 0);
          __builtin_amdgcn_sched_barrier(0);     // reads stay two k-steps ahead of their MFMAs, no further
        });
#pragma unroll
        for (int e = 0; e < (U8 ? 2 : 4); ++e) {
          q0[e] = f[KSTEPS][e];
          q1[e] = f[KSTEPS + 1][e];
        }
        if constexpr (U8) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] += __uint_as_float(cr[r >> 2][r & 3]);
        }
        pend = sign_bits(acc);
        if constexpr (CP) channel_words(pend, wave / MT + TSTEP * i, n, oy0);
      }
      // the last unit's transpose has no MFMAs to hide under
      static_for<0, 5>([&](auto pc) { epi_piece(pc, pend, wave / MT + TSTEP * (nunits - 1), st, true); });
      STEM_STAMP(0, 3 + j);
      __syncthreads();
    }
  }
}

}  // namespace

// Host side of the operand split: w [64][3][7][7] float32 -> fragment-ordered fp16 planes
// [ks][plane][mtile][lane][8]: lane l of M-tile m holds channel 32m + (l&31), k = 16ks + 8(l>>5) + j,
// k = ((c*7 + kh)*8 + slot), slot = kw + 1; slot 0 carries a zero weight, the 22nd row the BatchNorm shift.  init[64]: the
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

// init[0] = the constant c of the shift row (a power of two); returns false if the folded BatchNorm
// shift is too large for it (|shift| x weight prescale x 16 >= 2^30).
static int stem_mtiles(int p) { return p <= 32 ? 1 : (p <= 64 ? 2 : 4); }

bool stem_split_weights(const float *w, const double *scale, const double *shift, int p, uint16_t *out, float *init) {
  const int MT = stem_mtiles(p), CH = 32 * MT;
  // BatchNorm scale folded into the weights (float32 product, like any other float32 rounding of
  // the reference's conv + BN chain); the shift becomes the weights of the 22nd (c,kh) row, whose
  // "pixels" are the constant c in slots 0 and 1: slot 0 carries shift / c to 22 bits (two fp16
  // terms), slot 1 what is left of it, so the sum is the float64 shift to well below one float32 ulp.
  std::vector<float> wf((size_t)p * 147);
  for (int ch = 0; ch < p; ++ch)
    for (int i = 0; i < 147; ++i) wf[(size_t)ch * 147 + i] = (float)((double)w[(size_t)ch * 147 + i] * scale[ch]);
  const float ws = weight_prescale(wf.data(), wf.size());
  double sh[128], amax = 0.0;
  for (int ch = 0; ch < CH; ++ch) {
    sh[ch] = ch < p ? shift[ch] * (double)ws * (double)X_PRESCALE : -1.0;      // channels beyond p: bit 0
    amax = std::max(amax, std::fabs(sh[ch]));
  }
  if (!std::isfinite(amax)) return false;
  double c = 1.0;
  while (amax / c > 16384.0 && c < 32768.0) c *= 2.0;
  if (amax / c > 32768.0) return false;
  for (int i = 0; i < 64; ++i) init[i] = 0.f;
  init[0] = (float)c;
  for (int ks = 0; ks < KSTEPS; ++ks)
    for (int m = 0; m < MT; ++m)
      for (int l = 0; l < 64; ++l) {
        const int ch = 32 * m + (l & 31), R = 2 * ks + (l >> 5);
        double resid = 0.0;
        for (int j = 0; j < 8; ++j) {
          const int kw = j - 1;
          double v = 0.0;
          if (R < 21) {
            if (ch < p && kw >= 0) v = (double)(wf[(size_t)ch * 147 + R * 7 + kw] * ws);     // R = c*7 + kh
          } else if (j == 0) {
            v = sh[ch] / c;
          } else if (j == 1) {
            v = resid;
          }
          const uint16_t h1 = f32_to_f16_rne((float)v);
          const uint16_t h2 = f32_to_f16_rne((float)(v - (double)f16_to_f32(h1)));
          if (R == 21 && j == 0) resid = v - (double)f16_to_f32(h1) - (double)f16_to_f32(h2);
          const uint16_t parts[NPL] = {h1, h2};
          for (int pl = 0; pl < NPL; ++pl) out[((((size_t)ks * NPL + pl) * MT + m) * 64 + l) * 8 + j] = parts[pl];
        }
      }
  return true;
}

size_t stem_split_weights_elems() { return (size_t)KSTEPS * NPL * 4 * 64 * 8; }      // (sized for four M-tiles: p <= 128)

// uint8 input (stem_pc_kernel<true, ..>): the normalisation folded into the weights.  out / init as stem_split_weights, from
// w / (1020 std_c) and the interior bias shift + sum_all w d_c; tab = [4] centres s_c (int32) + [16 classes][MT][2][16] float32
// border corrections  - sum_{taps of the class in the padding} w d_c  (x the weight prescale), class = 4 yc + xc, yc / xc =
// 0: output row / column 0 (taps 0-2 padded), 1: row / column 1 (tap 0), 2: interior, 3: row / column 55 (taps 5, 6).
size_t stem_u8_table_elems() { return 4 + (size_t)16 * 128; }
bool stem_split_weights_u8(const float *w, const double *scale, const double *shift, int p, const float mean[3], const float stdv[3],
                           uint16_t *out, float *init, uint32_t *tab) {
  const int MT = stem_mtiles(p), CH = 32 * MT;
  std::vector<float> wf((size_t)p * 147);                 // BatchNorm scale folded in float32, as for float32 input
  for (int ch = 0; ch < p; ++ch)
    for (int i = 0; i < 147; ++i) wf[(size_t)ch * 147 + i] = (float)((double)w[(size_t)ch * 147 + i] * scale[ch]);
  int sc[3];
  double d[3], k[3];
  for (int c = 0; c < 3; ++c) {
    sc[c] = (int)std::lrint(1020.0 * (double)mean[c]);
    sc[c] = std::min(1020, std::max(0, sc[c]));
    k[c] = 1.0 / (1020.0 * (double)stdv[c]);
    d[c] = ((double)sc[c] - 1020.0 * (double)mean[c]) * k[c];
  }
  std::vector<float> w2((size_t)p * 147);                 // only to find the prescale
  for (int ch = 0; ch < p; ++ch)
    for (int i = 0; i < 147; ++i) w2[(size_t)ch * 147 + i] = (float)((double)wf[(size_t)ch * 147 + i] * k[i / 49]);
  const float ws = weight_prescale(w2.data(), w2.size());
  double sh[128], amax = 0.0;
  for (int ch = 0; ch < CH; ++ch) {
    double b = -1.0;                                      // channels beyond p: bit 0
    if (ch < p) {
      b = shift[ch];
      for (int i = 0; i < 147; ++i) b += (double)wf[(size_t)ch * 147 + i] * d[i / 49];
    }
    sh[ch] = b * (double)ws;
    amax = std::max(amax, std::fabs(sh[ch]));
  }
  if (!std::isfinite(amax)) return false;
  double c = 1.0;
  while (amax / c > 16384.0 && c < 32768.0) c *= 2.0;
  if (amax / c > 32768.0) return false;
  for (int i = 0; i < 64; ++i) init[i] = 0.f;
  init[0] = (float)c;
  for (int ks = 0; ks < KSTEPS; ++ks)
    for (int m = 0; m < MT; ++m)
      for (int l = 0; l < 64; ++l) {
        const int ch = 32 * m + (l & 31), R = 2 * ks + (l >> 5);
        double resid = 0.0;
        for (int j = 0; j < 8; ++j) {
          const int kw = j - 1;
          double v = 0.0;
          if (R < 21) {
            if (ch < p && kw >= 0) v = (double)wf[(size_t)ch * 147 + R * 7 + kw] * k[R / 7] * (double)ws;      // R = c*7 + kh
          } else if (j == 0) {
            v = sh[ch] / c;
          } else if (j == 1) {
            v = resid;
          }
          const uint16_t h1 = f32_to_f16_rne((float)v);
          const uint16_t h2 = f32_to_f16_rne((float)(v - (double)f16_to_f32(h1)));
          if (R == 21 && j == 0) resid = v - (double)f16_to_f32(h1) - (double)f16_to_f32(h2);
          const uint16_t parts[NPL] = {h1, h2};
          for (int pl = 0; pl < NPL; ++pl) out[((((size_t)ks * NPL + pl) * MT + m) * 64 + l) * 8 + j] = parts[pl];
        }
      }
  for (int c3 = 0; c3 < 3; ++c3) tab[c3] = (uint32_t)sc[c3];
  tab[3] = 0;
  auto padded = [](int cls, int t) { return cls == 0 ? t <= 2 : (cls == 1 ? t == 0 : (cls == 3 ? t >= 5 : false)); };
  for (int yc = 0; yc < 4; ++yc)
    for (int xc = 0; xc < 4; ++xc)
      for (int m = 0; m < MT; ++m)
        for (int hh = 0; hh < 2; ++hh)
          for (int r = 0; r < 16; ++r) {
            const int ch = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh;
            double corr = 0.0;
            if (ch < p)
              for (int i = 0; i < 147; ++i) {
                const int kh = (i % 49) / 7, kw = i % 7;
                if (padded(yc, kh) || padded(xc, kw)) corr -= (double)wf[(size_t)ch * 147 + i] * d[i / 49];
              }
            const float cf = (float)(corr * (double)ws);
            uint32_t bits;
            memcpy(&bits, &cf, 4);
            tab[4 + ((((size_t)(yc * 4 + xc) * MT + m) * 2 + hh) * 16 + r)] = bits;
          }
  return true;
}

// (x, wfrag, init, rp, cp, p, n, norm_tab, range_flag): keep in step with the kernel's signature
int stem_kernel_arg_sizes(const int **sizes) {
  static const int kSizes[9] = {8, 8, 8, 8, 8, 4, 4, 8, 8};
  *sizes = kSizes;
  return 9;
}

int launch_stem(const void *x, bool x_is_u8, const uint32_t *norm_tab, const void *wfrag, const float *init, uint64_t *rp,
                uint16_t *cp, int n, int p, uint32_t *range_flag, hipStream_t s, int workgroups) {
  if (p < 1 || p > 128 || (cp && p != 64)) {
    set_error("stem: p=%d outside [1,128] (channel words need p = 64)", p);
    return TTNET_E_UNSUPPORTED;
  }
  const int MT = stem_mtiles(p);
  if (((uintptr_t)x & (x_is_u8 ? 3 : 15)) != 0) {
    set_error("stem: the input must be %d-byte aligned", x_is_u8 ? 4 : 16);
    return TTNET_E_INVALID;
  }
  const size_t lds = (size_t)2 * TILE_DW * 4 + (size_t)2 * 32 * MT * (NT + 2) * 4 + (x_is_u8 ? (size_t)16 * 32 * MT * 4 : 0);
  const int items = n * NBLK;
  // Persistent workgroups: one per CU when a forward has the chip to itself; HALF the CUs when the plan keeps several batches in
  // flight.  The kernel is bound by the energy of its matrix instructions, not by per-CU throughput: on 128 CUs it holds a
  // higher clock and takes 73 - 79 us instead of 57, and the other batch's kernels run on the other 128 meanwhile -- the forward
  // with two batches in flight gains 5 - 7 % (profiles/r03_cache_policy.txt, same-box A/B; one batch at a time would lose 4 %; the
  // uint8 kernel, with two products, and the full variant gain nothing and keep 256: the caller decides).
  static const int grid_env = getenv("TTNET_STEM_GRID") ? atoi(getenv("TTNET_STEM_GRID")) : 0;        // (diagnostic override)
  const int grid = std::min(items, std::max(1, grid_env > 0 ? grid_env : (workgroups > 0 ? workgroups : 256)));
  auto launch = [&](auto kernel) -> int {
    TT_TRY(ensure_dynamic_lds((const void *)kernel, lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(STEM_THREADS), lds, s, x, (const uint4 *)wfrag, init, rp, cp, p, n, norm_tab,
                       range_flag);
    return TTNET_OK;
  };
  if (cp) TT_TRY(x_is_u8 ? launch(stem_pc_kernel<true, true, 2>) : launch(stem_pc_kernel<false, true, 2>));
  else if (MT == 1) TT_TRY(x_is_u8 ? launch(stem_pc_kernel<true, false, 1>) : launch(stem_pc_kernel<false, false, 1>));
  else if (MT == 2) {
    static const int nt = getenv("TTNET_STEM_NT") ? atoi(getenv("TTNET_STEM_NT")) : TT_STEM_LOAD_AUX;      // (diagnostic: =0 the default cache policy)
    if (x_is_u8) TT_TRY(nt ? launch(stem_pc_kernel<true, false, 2>) : launch((stem_pc_kernel<true, false, 2, 0>)));
    else TT_TRY(nt ? launch(stem_pc_kernel<false, false, 2>) : launch((stem_pc_kernel<false, false, 2, 0>)));
  }
  else TT_TRY(x_is_u8 ? launch(stem_pc_kernel<true, false, 4>) : launch(stem_pc_kernel<false, false, 4>));
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
