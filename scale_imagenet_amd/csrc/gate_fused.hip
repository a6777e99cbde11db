// Block-fused gate path: one launch per 4-branch block.
//
// Replaces Block_resnet_multihead_general_BN_vf_imgnet_v2small.forward
// (models/TT_general_imagenet_v2_small.py:78-148) for the stride-2 blocks of TT-small: the three
// branch Block_TTs (:88-90), both 2x2 majority pools (:91-94), the shape-keyed zero padding
// (:98-139), the interleave (:142-147) and Block_convf (:148) run inside ONE kernel, so the branch
// tensors out1..out4 never leave the chip (gate.hip's two-launch version writes them to HBM and
// reads them back: 2.5x the algorithmic traffic of SURVEY 8(d)).
//
// Decomposition.  The grouped structure of the network keeps channels local: Block_convf group g
// reads channels 4g..4g+3 of every branch, so the 8 input channels 8s..8s+7 ("strand" s) produce,
// through conv1 / conv2 / the majorities and two convf groups, exactly one 16-channel word of the
// block's output -- with no data from any other strand except the raw input channels of its
// Block_conv3 group.  A workgroup owns (strand, slice of the batch) and walks five table phases of
// 64 KiB each over its images:
//     A   Block_conv3 (own 8 output bits of the 16 -> 16 table, as a byte table) + both majorities
//     B1  Block_conv1 + Block_conv2 of channels 0-3      B2  the same for channels 4-7
//     C1  Block_convf group 2s                            C2  Block_convf group 2s+1, output rows
// The tables alternate between two 64 KiB LDS buffers: while a phase computes, the next phase's
// table streams into the other buffer (direct global->LDS loads) and the next phase's input rows
// into registers, so that a phase itself never waits on vector memory (vmcnt retires in order: a
// wait for an activation load would also wait for the table stream issued before it).  Between the
// phases a pixel's 32 branch bits live in one LDS dword:
//     byte 0  out1 | out2 << 4 of channels 0-3 (B1)       byte 1  the same for channels 4-7 (B2)
//     byte 2  out3 | out4 << 4 of channels 0-3 (A)        byte 3  the same for channels 4-7 (A)
// so that bytes (0,2) / (1,3) are the 16-bit table indices of the two convf groups.
// For the last block (float Block_convf, gate.hip: gate_last*) the dwords are the kernel's output.
//
// Activations in HBM: ONE layout, rows of packed pixels per (image, channel): uint64 for W > 32,
// uint32 for W > 16, uint16 otherwise.  Block_conv3 needs the 16 channels of a pixel as a word:
// 16 lanes transpose their rows in registers (ttnet_common.h: transpose16).
//
// Placement: all batch slices of a strand pair are given block ids that land on one XCD
// (round-robin dispatch, speed only), so a table is pulled from HBM into one L2 once and the
// Block_conv3 rows shared by the two strands of a pair are fetched once.
//
// Bound: integer VALU (index forming, bit extraction, transposes) / LDS gather; HBM traffic is the
// algorithmic minimum + ~10 % row padding.

#include <type_traits>

#include "ttnet_common.h"

namespace ttnet {

namespace {

// tools/ubench/fused_phases.hip builds this file with parts switched off (an additive decomposition of
// the kernel's time): 1 no table streams, 2 no phase A, 4 no phase B, 8 no phase C (bit mask).  0 in the library.
#ifndef TT_FUSED_SKIP
#define TT_FUSED_SKIP 0
#endif
constexpr int kSkip = TT_FUSED_SKIP;

constexpr int kFT = 1024;              // threads per workgroup: 16 waves, 4 per SIMD
constexpr int kFBuf = 65536;           // one table buffer
constexpr int kFMaxScratch = 160 * 1024 - 2 * kFBuf;

template <int W>
using row_t = std::conditional_t<(W > 32), uint64_t, std::conditional_t<(W > 16), uint32_t, uint16_t>>;

// images per round of a workgroup: one depthwise (channel, output row) task per thread, the branch dwords
// of a round in the LDS left over by the two table buffers
template <int HO>
constexpr int fused_round() {
  constexpr int RG = (HO + 3) / 4, a = kFT / (16 * RG), b = kFMaxScratch / (HO * HO * 4);
  return a < b ? (a < 32 ? a : 32) : (b < 32 ? b : 32);
}

struct FusedArgs {
  int n, C, slices, off34, R;
  const void *x;              // rows [n][C][H]
  const uint8_t *img_c3;      // [C/8][65536]: Block_conv3 of the strand's group, own 8 output bits
  const uint8_t *img_dw;      // [C/8][2][2048][4][2] dwords: conv1 / conv2 bit tables of 4 channels, interleaved
  const uint8_t *t_cf;        // [C/4][65536] bytes (binarised Block_convf); unused by a last block
  void *y;                    // rows [n][2C][HO] (binarised blocks)
  uint32_t *idx;              // [n][C/8][HO*HO] branch dwords: the output of a last block; optional tap otherwise
};

__device__ inline void dma_table(uint8_t *dst, const uint8_t *src) {
  if constexpr (kSkip & 1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kFBuf / 1024 / (kFT / 64); ++k) {
    const int chunk = wave + k * (kFT / 64);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)chunk * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void *)(dst + chunk * 1024), 16, 0, 0);
  }
}

// phase boundary: every table piece and input row this wave asked for has landed, then the workgroup meets
__device__ inline void phase_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__device__ inline uint32_t spread_nibbles(uint32_t x) {      // nibbles 0..3 of x -> low nibbles of bytes 0..3
  uint32_t y = (x | (x << 8)) & 0x00FF00FFu;
  return (y | (y << 4)) & 0x0F0F0F0Fu;
}

// ---- depthwise lookups --------------------------------------------------------------------------
// Index forming by a 4 x 8 nibble transpose: for even output columns ox = 2k the 4-bit window row
// (input columns 2ox-2 .. 2ox+1) is nibble k of the row shifted left by the 2 padding columns; for odd
// columns it is nibble k of the unshifted row.  With four such rows R0..R3 in registers,
//     t0 = bfi(0x0F0F0F0F, R0, R1 << 4)   bytes j = [R1.nib(2j) : R0.nib(2j)]
//     t1 = bfi(0x0F0F0F0F, R0 >> 4, R1)   bytes j = [R1.nib(2j+1) : R0.nib(2j+1)]
// (likewise u0, u1 from R2, R3) and one byte permute joins a t byte and a u byte into a complete
// 16-bit index, two per dword: 12 instructions for 8 indices.  A lookup is one 8-byte LDS read (the
// conv1 and conv2 dwords of the channel sit side by side), a bit-field extract whose offset is the low
// 5 bits of the index register as it stands, and a shift-or into the output row.
template <int WO, int OX0, int OX1>
__device__ inline void dw_lookup_pair(uint32_t pair, const uint8_t *tab, uint32_t c8, uint32_t &acc1, uint32_t &acc2) {
  if constexpr (OX0 < WO) {
    const uint2 w = *(const uint2 *)(tab + ((pair & 0xFFE0u) | c8));
    acc1 |= __builtin_amdgcn_ubfe(w.x, pair, 1) << OX0;
    acc2 |= __builtin_amdgcn_ubfe(w.y, pair, 1) << OX0;
  }
  if constexpr (OX1 < WO) {
    const uint32_t hi = pair >> 16;
    const uint2 w = *(const uint2 *)(tab + ((hi & 0xFFE0u) | c8));
    acc1 |= __builtin_amdgcn_ubfe(w.x, hi, 1) << OX1;
    acc2 |= __builtin_amdgcn_ubfe(w.y, hi, 1) << OX1;
  }
}

// eight nibble positions of four rows; nibble k is output column OXB + OXS * k
template <int WO, int OXB, int OXS>
__device__ inline void dw_eight(uint32_t R0, uint32_t R1, uint32_t R2, uint32_t R3, const uint8_t *tab, uint32_t c8,
                                uint32_t &acc1, uint32_t &acc2) {
  if constexpr (OXB >= WO) return;
  constexpr uint32_t M = 0x0F0F0F0Fu;
  const uint32_t t0 = (R0 & M) | ((R1 << 4) & ~M), u0 = (R2 & M) | ((R3 << 4) & ~M);
  dw_lookup_pair<WO, OXB, OXB + 2 * OXS>(__builtin_amdgcn_perm(u0, t0, 0x05010400u), tab, c8, acc1, acc2);
  dw_lookup_pair<WO, OXB + 4 * OXS, OXB + 6 * OXS>(__builtin_amdgcn_perm(u0, t0, 0x07030602u), tab, c8, acc1, acc2);
  if constexpr (OXB + OXS < WO) {
    const uint32_t t1 = ((R0 >> 4) & M) | (R1 & ~M), u1 = ((R2 >> 4) & M) | (R3 & ~M);
    dw_lookup_pair<WO, OXB + OXS, OXB + 3 * OXS>(__builtin_amdgcn_perm(u1, t1, 0x05010400u), tab, c8, acc1, acc2);
    dw_lookup_pair<WO, OXB + 5 * OXS, OXB + 7 * OXS>(__builtin_amdgcn_perm(u1, t1, 0x07030602u), tab, c8, acc1, acc2);
  }
}

// one output row (all WO columns) of conv1 and conv2 for one channel, from its four window rows
template <int W, int WO>
__device__ inline void dw_row_both(const row_t<W> (&r)[4], const uint8_t *tab, uint32_t c8, uint32_t &acc1, uint32_t &acc2) {
  acc1 = 0;
  acc2 = 0;
  if constexpr (W > 32) {
    uint32_t plo[4], phi[4], qlo[4], qhi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      qlo[k] = (uint32_t)r[k];
      qhi[k] = (uint32_t)((uint64_t)r[k] >> 32);
      plo[k] = qlo[k] << 2;
      phi[k] = __builtin_amdgcn_alignbit(qhi[k], qlo[k], 30);
    }
    dw_eight<WO, 0, 2>(plo[0], plo[1], plo[2], plo[3], tab, c8, acc1, acc2);      // even columns 0..14
    dw_eight<WO, 16, 2>(phi[0], phi[1], phi[2], phi[3], tab, c8, acc1, acc2);     // even columns 16..30
    dw_eight<WO, 1, 2>(qlo[0], qlo[1], qlo[2], qlo[3], tab, c8, acc1, acc2);      // odd columns 1..15
    dw_eight<WO, 17, 2>(qhi[0], qhi[1], qhi[2], qhi[3], tab, c8, acc1, acc2);     // odd columns 17..31
  } else {
    uint32_t p[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      q[k] = (uint32_t)r[k];
      p[k] = q[k] << 2;
    }
    dw_eight<WO, 0, 2>(p[0], p[1], p[2], p[3], tab, c8, acc1, acc2);
    dw_eight<WO, 1, 2>(q[0], q[1], q[2], q[3], tab, c8, acc1, acc2);
    static_assert(WO <= 16, "a 32-bit input row gives at most 16 output columns");
  }
}

__device__ inline uint32_t maj4_bytes(uint32_t v, uint32_t nb) {
  // v = this lane's column, nb = the next column; byte 0 / 1 = upper / lower input row (bytes 2 / 3:
  // the same for the second half of the chunk).  Result in bytes 0 and 2: at least two of four set.
  const uint32_t x = v & nb, y = v | nb;
  return (x | (x >> 8) | (y & (y >> 8))) & 0x00FF00FFu;
}

template <int H, int HO, bool LAST>
__global__ __launch_bounds__(kFT) void gate_block_kernel(FusedArgs a) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int W = H, WO = HO, HP = H / 2, WP = W / 2, PIX = HO * WO;
  constexpr int RG = (HO + 3) / 4;                     // groups of 4 output rows
  constexpr int LPR = WO <= 16 ? 16 : 32, RPW = 64 / LPR;
  using TI = row_t<W>;
  using TOut = row_t<WO>;
  uint8_t *const buf0 = lds, *const buf1 = lds + kFBuf;
  uint32_t *const S = (uint32_t *)(lds + 2 * kFBuf);   // [R][PIX] branch dwords

  // ---- which strand, which images (placement: see the file header) ----------------------------
  const int strands = a.C / 8, P = strands / 2;
  int st, sl;
  {
    const int b = blockIdx.x, x = b & 7, k = b >> 3;
    if (P >= 8 && P % 8 == 0) {                          // (p = 64: 8, 16, 32 strand pairs; other widths take the plain order)
      const int m = P / 8, pair = x + 8 * (k % m), kk = k / m;
      st = 2 * pair + (kk & 1);
      sl = kk >> 1;
    } else if (P == 4 && (a.slices & 1) == 0) {
      st = 2 * (x >> 1) + (k & 1);
      sl = (x & 1) * (a.slices / 2) + (k >> 1);
    } else {
      st = b % strands;
      sl = b / strands;
    }
  }
  const int n0 = (int)((long long)sl * a.n / a.slices), n1 = (int)((long long)(sl + 1) * a.n / a.slices);
  if (n0 >= n1) return;
  const int q = st >> 1, half = st & 1;
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const DwLaneConst lk = dw_lane_const(lane);
  // Each phase derives its indices from an opaque copy of the thread id: otherwise the compiler hoists
  // every phase's (loop-invariant) task decomposition and addresses out of the round loop and keeps them
  // live across all phases -- more than the 128 registers a 16-wave workgroup has per lane.
  auto opaque_tid = [&]() {
    int t = tid0;
    asm volatile("" : "+v"(t));
    return t;
  };
  const TI *const x = (const TI *)a.x;

  const uint8_t *const src_c3 = a.img_c3 + (size_t)st * kFBuf;
  const uint8_t *const src_dw0 = a.img_dw + (size_t)(2 * st) * kFBuf, *const src_dw1 = src_dw0 + kFBuf;
  const uint8_t *const src_cf0 = LAST ? nullptr : a.t_cf + (size_t)(2 * st) * kFBuf, *const src_cf1 = LAST ? nullptr : src_cf0 + kFBuf;

  // ---- register-resident inputs of the phases (loaded one phase ahead) -------------------------
  constexpr int RMAX = fused_round<HO>();                  // images per round (host: launch_block_t)
  constexpr int TA = (RMAX * HP + kFT / 16 - 1) / (kFT / 16);   // phase-A tasks (image, row pair) per 16-lane group
  static_assert(RMAX * RG * 16 <= kFT, "one depthwise task per thread");
  TI ra[TA][2];
  TI rb0[4], rb1[4];
  const int R = a.R;

  auto load_a = [&](int i0, int rn) {
    // task (image, input row pair) per 16-lane group; lane c = channel 16q + c of the Block_conv3 group
    const int tid = opaque_tid();
    const int c = tid & 15, g = tid >> 4;
#pragma unroll
    for (int p = 0; p < TA; ++p) {
      const int gt = p * (kFT / 16) + g;
      const bool ok = gt < rn * HP;
      const int im = ok ? gt / HP : 0, py = ok ? gt - (gt / HP) * HP : 0;
      const TI *src = x + ((size_t)(n0 + i0 + im) * a.C + 16 * q + c) * H + 2 * py;
      ra[p][0] = ok ? src[0] : (TI)0;
      ra[p][1] = ok ? src[1] : (TI)0;
    }
  };
  auto load_b = [&](int i0, int rn, int sub, TI (&rb)[4]) {
    const int tid = opaque_tid();
    const int c4 = tid & 3, r = (tid >> 2) & 3, gi = tid >> 4;
    const int im = gi / RG, rg = gi - im * RG;
    const int oy = 4 * rg + r;
    const bool ok = im < rn && oy < HO;
    const TI *src = x + ((size_t)(n0 + i0 + (ok ? im : 0)) * a.C + 8 * st + 4 * sub + c4) * H;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const int iy = 2 * oy - 2 + kh;
      rb[kh] = (ok && iy >= 0 && iy < H) ? src[iy] : (TI)0;
    }
  };

  // ---- phase A: Block_conv3 + the two majority pools -> bytes 2, 3 --------------------------------
  auto phase_a = [&](int rn, const uint8_t *tab) {
    if constexpr (kSkip & 2) return;
    // zero border of out3 / out4 (ZeroPad2d((1,0,1,0)) at W = 56, (0,1,0,1) otherwise, :98-139)
    const int tid = opaque_tid();
    for (int t = tid; t < rn * (HO + WO - 1); t += kFT) {
      const int im = t / (HO + WO - 1), e = t - im * (HO + WO - 1);
      const int edge = a.off34 ? 0 : HO - 1;
      const int pix = e < WO ? edge * WO + e : (e - WO + (a.off34 ? 1 : 0)) * WO + edge;
      *(uint16_t *)((uint8_t *)(S + im * PIX + pix) + 2) = 0;
    }
    const int g = tid >> 4, j = tid & 15;
    // Four transposes at a time (their LDS round trips overlap): the two rows x two 32-pixel chunks of a
    // task at W = 56, the two rows of two tasks otherwise.  Unit u of a batch = one (task, chunk).
    constexpr int CHUNKS = W > 32 ? 2 : 1, TPB = 2 / CHUNKS;          // tasks per batch
    static_assert(TA % TPB == 0, "phase A walks whole batches");
#pragma unroll
    for (int bt = 0; bt < TA / TPB; ++bt) {
      uint32_t d[4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = bt * TPB + (CHUNKS == 2 ? 0 : u), k = CHUNKS == 2 ? u : 0;
        d[2 * u] = (uint32_t)((uint64_t)ra[p][0] >> (32 * k));
        d[2 * u + 1] = (uint32_t)((uint64_t)ra[p][1] >> (32 * k));
      }
      // lane c holds 32 pixels of channel c; transposed: lane j holds the 16-channel words of pixels
      // 32k + j (low half) and 32k + 16 + j (high half)
      transpose16_multi<4>(d, lk);
      uint32_t v3[2], v4[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t t0 = d[2 * u], t1 = d[2 * u + 1];
        const uint32_t e00 = tab[t0 & 0xFFFFu], e01 = tab[t0 >> 16], e10 = tab[t1 & 0xFFFFu], e11 = tab[t1 >> 16];
        v3[u] = e00 | (e10 << 8) | (e01 << 16) | (e11 << 24);                                 // conv3: rows x halves
        v4[u] = __builtin_amdgcn_perm(t1, t0, half ? 0x07030501u : 0x06020400u);              // raw input, own byte
      }
      // the next column's values (lane j ^ 1): both lanes of a column pair get the same majorities
      uint32_t n3[2], n4[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        n3[u] = lane_xor16<1>(v3[u]);
        n4[u] = lane_xor16<1>(v4[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = bt * TPB + (CHUNKS == 2 ? 0 : u), k = CHUNKS == 2 ? u : 0;
        const int gt = p * (kFT / 16) + g;
        const bool ok = gt < rn * HP;                    // uniform over the 16-lane group
        const int im = ok ? gt / HP : 0, py = ok ? gt - (gt / HP) * HP : 0;
        const uint32_t m3 = maj4_bytes(v3[u], n3[u]), m4 = maj4_bytes(v4[u], n4[u]);
        // the even lane of a pair stores the pooled pixel of the low half, the odd lane that of the high half
        const uint32_t sh = (uint32_t)(j & 1) * 16u;
        const uint32_t t = ((m3 >> sh) & 0xFFu) | (((m4 >> sh) & 0xFFu) << 8);     // nibbles [o4.hi o4.lo o3.hi o3.lo]
        // bytes 2 / 3 of the branch dword: (out3 | out4 << 4) of channels 0-3 / 4-7 = the middle nibbles swapped
        const uint32_t dlt = ((t >> 4) ^ t) & 0x00F0u;
        const uint32_t v = t ^ dlt ^ (dlt << 4);
        const int px = 16 * k + 8 * (j & 1) + (j >> 1);
        if (ok && px < WP) {
          const int pix = (py + a.off34) * WO + px + a.off34;
          *(uint16_t *)((uint8_t *)(S + im * PIX + pix) + 2) = (uint16_t)v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);               // one batch at a time: keeps the live set small
    }
  };

  // ---- phase B: Block_conv1 + Block_conv2 of four channels -> byte `sub` -----------------------------
  // B1 only looks up (its two output rows stay in registers); B2 looks up, then transposes all four rows
  // together and stores bytes 0 and 1 of every pixel as one 16-bit word.
  uint32_t bacc[2][2];
  auto phase_b_lookup = [&](const uint8_t *tab, int sub, const TI (&rb)[4]) {
    if constexpr (kSkip & 4) return;
    const int tid = opaque_tid();
    dw_row_both<W, WO>(rb, tab, (uint32_t)(tid & 3) * 8u, bacc[sub][0], bacc[sub][1]);
  };
  auto phase_b_store = [&](int rn) {
    if constexpr (kSkip & 4) return;
    const int tid = opaque_tid();
    const int j = tid & 15, gi = tid >> 4;
    const int im = gi / RG, rg = gi - im * RG;
    // 16-lane group = 4 channels x 4 output rows (lane = c + 4 r).  Transposed: lane j holds, for
    // columns j (low half) and 16 + j (high half), nibble r = the 4 channels of output row 4 rg + r
    uint32_t t[4] = {bacc[0][0], bacc[0][1], bacc[1][0], bacc[1][1]};
    transpose16_multi<4>(t, lk);
    uint8_t *dst = (uint8_t *)(S + im * PIX + 4 * rg * WO + j);
    auto put = [&](uint32_t w0, uint32_t w1, int col0) {      // w0 / w1: byte r = (out1 | out2 << 4) of channels 0-3 / 4-7
      if (im < rn && col0 + j < WO) {
        const uint32_t lo = __builtin_amdgcn_perm(w1, w0, 0x05010400u), hi = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
        if (4 * rg + 0 < HO) *(uint16_t *)(dst + (0 * WO + col0) * 4) = (uint16_t)lo;
        if (4 * rg + 1 < HO) *(uint16_t *)(dst + (1 * WO + col0) * 4) = (uint16_t)(lo >> 16);
        if (4 * rg + 2 < HO) *(uint16_t *)(dst + (2 * WO + col0) * 4) = (uint16_t)hi;
        if (4 * rg + 3 < HO) *(uint16_t *)(dst + (3 * WO + col0) * 4) = (uint16_t)(hi >> 16);
      }
    };
    put(spread_nibbles(t[0] & 0xFFFFu) | (spread_nibbles(t[1] & 0xFFFFu) << 4),
        spread_nibbles(t[2] & 0xFFFFu) | (spread_nibbles(t[3] & 0xFFFFu) << 4), 0);
    if constexpr (WO > 16)
      put(spread_nibbles(t[0] >> 16) | (spread_nibbles(t[1] >> 16) << 4), spread_nibbles(t[2] >> 16) | (spread_nibbles(t[3] >> 16) << 4), 16);
  };

  // ---- phases C1 / C2: Block_convf group 2s + k -> output channels 16 s + 8 k .. + 7, as rows -----------------
  // A lane looks up two pixels, the same column in two rows, and carries their bytes as one 16-bit
  // value; the 16 x 16 transpose of a 16-lane group then leaves channel j of the first row in lane j
  // and of the second row in lane 8 + j.  A wave task = 2 * RPW row pairs... i.e. 2 * RPW rows x LPR columns.
  auto phase_c = [&](int i0, int rn, const uint8_t *tab, int kgrp) {
    if constexpr (kSkip & 8) return;
    const int tid = opaque_tid();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane / LPR, ox = lane % LPR;
    TOut *const y = (TOut *)a.y;
    constexpr int CH2 = (HO + 2 * RPW - 1) / (2 * RPW);       // wave tasks per image (2 RPW rows each)
    constexpr int U = 2, NW = kFT / 64;                       // wave tasks per trip
    const uint32_t sel = kgrp ? 0x0C0C0301u : 0x0C0C0200u;    // index of group k: bytes (k, 2 + k) of the branch dword
    if (kgrp == 0 && a.idx)                                   // parity tap: the branch dwords as the lookups see them
      for (int t = tid; t < rn * PIX; t += kFT) a.idx[((size_t)(n0 + i0 + t / PIX) * strands + st) * PIX + (t % PIX)] = S[t];
    for (int wt0 = wave; wt0 < rn * CH2; wt0 += U * NW) {
      uint32_t w[U], d0[U], d1[U];
      int im[U], oy[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int wt = wt0 + u * NW;                          // (wave-uniform)
        im[u] = wt / CH2;
        oy[u] = (wt - im[u] * CH2) * 2 * RPW + sub;           // this lane's first row; its second is RPW below
        const bool live = wt < rn * CH2 && ox < WO;
        d0[u] = live && oy[u] < HO ? S[im[u] * PIX + oy[u] * WO + ox] : 0u;
        d1[u] = live && oy[u] + RPW < HO ? S[im[u] * PIX + (oy[u] + RPW) * WO + ox] : 0u;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        w[u] = ox < WO ? (uint32_t)tab[__builtin_amdgcn_perm(0u, d0[u], sel)] | ((uint32_t)tab[__builtin_amdgcn_perm(0u, d1[u], sel)] << 8)
                       : 0u;                                  // (pixels beyond the row stay zero: the next block's padding)
      // lane = pixel, bit = (row, channel) -> lane 8 q + j = channel j of row q, bit = pixel of the 16-lane group
      transpose16_multi<U>(w, lk);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t piece = w[u] & 0xFFFFu;
        uint32_t rowbits = piece;
        if constexpr (LPR == 32) rowbits |= (uint32_t)__shfl_xor((int)piece, 16) << 16;
        const int row = oy[u] + ((lane >> 3) & 1) * RPW;
        if (wt0 + u * NW < rn * CH2 && (lane & (LPR - 1)) < 16 && row < HO)
          y[((uint32_t)(n0 + i0 + im[u]) * (uint32_t)(2 * a.C) + 16 * st + 8 * kgrp + (lane & 7)) * HO + row] = (TOut)rowbits;
      }
    }
  };

  // ---- last block: the branch dwords are the output -----------------------------------------------------
  auto phase_out = [&](int i0, int rn) {
    const int tid = opaque_tid();
    for (int t = tid; t < rn * PIX; t += kFT)
      a.idx[((size_t)(n0 + i0 + t / PIX) * strands + st) * PIX + (t % PIX)] = S[t];
  };

  // ---- the pipeline ------------------------------------------------------------------------------------
  // Every phase boundary is one phase_sync(): the table of the phase about to start and its input rows
  // have landed, every wave has left the previous phase (its table buffer and, after C2, the scratch are
  // free).  Right after it the NEXT phase's rows and table are requested; the phase itself then works
  // from registers and LDS only.  (Measured alternative: 512-thread workgroups, two per CU, one table
  // buffer each, the other workgroup computing while one streams -- 28 us instead of 21 for the first
  // block at B = 256: twice the table bytes per CU, and the L2 -> LDS stream, ~100 GB/s per CU, binds.)
  const int total = n1 - n0;
  int cur = 0;
  load_a(0, min(R, total));
  dma_table(buf0, src_c3);
  for (int i0 = 0; i0 < total; i0 += R) {
    const int rn = min(R, total - i0);
    const bool more = i0 + R < total;
    uint8_t *tb = cur ? buf1 : buf0, *nb = cur ? buf0 : buf1;
    phase_sync();
    load_b(i0, rn, 0, rb0);
    dma_table(nb, src_dw0);
    phase_a(rn, tb);
    cur ^= 1; tb = cur ? buf1 : buf0; nb = cur ? buf0 : buf1;
    phase_sync();
    load_b(i0, rn, 1, rb1);
    dma_table(nb, src_dw1);
    phase_b_lookup(tb, 0, rb0);
    cur ^= 1; tb = cur ? buf1 : buf0; nb = cur ? buf0 : buf1;
    phase_sync();
    if constexpr (LAST) {
      if (more) {
        load_a(i0 + R, min(R, total - i0 - R));
        dma_table(nb, src_c3);
      }
      phase_b_lookup(tb, 1, rb1);
      phase_b_store(rn);
      cur ^= 1;
      phase_sync();
      phase_out(i0, rn);
    } else {
      dma_table(nb, src_cf0);
      phase_b_lookup(tb, 1, rb1);
      phase_b_store(rn);
      cur ^= 1; tb = cur ? buf1 : buf0; nb = cur ? buf0 : buf1;
      phase_sync();
      dma_table(nb, src_cf1);
      phase_c(i0, rn, tb, 0);
      cur ^= 1; tb = cur ? buf1 : buf0; nb = cur ? buf0 : buf1;
      phase_sync();
      if (more) {
        load_a(i0 + R, min(R, total - i0 - R));
        dma_table(nb, src_c3);
      }
      phase_c(i0, rn, tb, 1);
      cur ^= 1;
    }
  }
}

// ---- table images (finalize) ---------------------------------------------------------------------------
// img_c3[strand][idx] = byte (strand & 1) of the 16-bit Block_conv3 entry of group strand >> 1
__global__ void c3_image_kernel(const uint16_t *__restrict__ t_c3, uint8_t *__restrict__ img, int strands) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)strands * 65536) return;
  const int st = (int)(t >> 16);
  const uint32_t idx = (uint32_t)(t & 0xFFFF);
  img[t] = (uint8_t)(t_c3[(size_t)(st >> 1) * 65536 + idx] >> (8 * (st & 1)));
}
// img_dw[strand][sub][row][ch][branch] (dwords) from the striped bit tables [C/16][2048][16]
__global__ void dw_image_kernel(const uint32_t *__restrict__ t1, const uint32_t *__restrict__ t2, uint32_t *__restrict__ img, int C) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one dword of the image
  if (t >= (size_t)C * 2048 * 2) return;
  const int br = (int)(t & 1), ch = (int)((t >> 1) & 3), row = (int)((t >> 3) & 2047), ss = (int)(t >> 14);   // ss = 2 strand + sub
  const int channel = 4 * ss + ch;
  const uint32_t *src = br ? t2 : t1;
  img[t] = src[((size_t)(channel >> 4) * 2048 + row) * 16 + (channel & 15)];
}

// ---- parity taps ------------------------------------------------------------------------------------------
// branch dwords [n][C/8][PIX] -> row-packed uint64 [n][C][HO] of branch br (0..3 = out1..out4)
__global__ void branch_rows_kernel(const uint32_t *__restrict__ idx, uint64_t *__restrict__ rows, int n, int C, int HO, int br) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * C * HO) return;
  const int oy = (int)(t % HO), c = (int)((t / HO) % C), im = (int)(t / ((size_t)HO * C));
  const int st = c >> 3, k = c & 7;
  // byte of the dword: (k >> 2) for out1 / out2, 2 + (k >> 2) for out3 / out4; nibble: odd branches high
  const int shift = 8 * ((br >> 1) * 2 + (k >> 2)) + 4 * (br & 1) + (k & 3);
  const uint32_t *src = idx + ((size_t)im * (C / 8) + st) * HO * HO + (size_t)oy * HO;
  uint64_t r = 0;
  for (int ox = 0; ox < HO; ++ox) r |= (uint64_t)((src[ox] >> shift) & 1u) << ox;
  rows[t] = r;
}
template <typename T>
__global__ void widen_rows_kernel(const T *__restrict__ src, uint64_t *__restrict__ dst, size_t count) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) dst[t] = (uint64_t)src[t];
}
template <typename T>
__global__ void narrow_rows_kernel(const uint64_t *__restrict__ src, T *__restrict__ dst, size_t count) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) dst[t] = (T)src[t];
}

template <int H, int HO, bool LAST>
int launch_block_t(const FusedBlockArgs &f, hipStream_t s) {
  auto k = gate_block_kernel<H, HO, LAST>;
  constexpr int PIX = HO * HO;
  const int strands = f.C / 8;
  int R = fused_round<HO>();
  int slices = std::max(1, std::min(f.n, 256 / strands));
  if (strands == 8 && slices > 1 && (slices & 1)) slices -= 1;       // (the pair placement wants an even count)
  R = std::max(1, std::min(R, (f.n + slices - 1) / slices));
  const size_t lds = (size_t)2 * kFBuf + (size_t)R * PIX * 4;
  TT_TRY(ensure_dynamic_lds((const void *)k, kMaxLds));
  FusedArgs a{};
  a.n = f.n; a.C = f.C; a.slices = slices; a.off34 = f.off34; a.R = R;
  a.x = f.x; a.img_c3 = (const uint8_t *)f.img_c3; a.img_dw = (const uint8_t *)f.img_dw; a.t_cf = f.t_cf; a.y = f.y; a.idx = f.idx;
  hipLaunchKernelGGL(k, dim3(strands * slices), dim3(kFT), lds, s, a);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace

int row_bytes(int W) { return W > 32 ? 8 : (W > 16 ? 4 : 2); }

bool fused_block_supported(int C, int H, int Ho, int stride, int pad, int kh, int kw) {
  if (stride != 2 || pad != 2 || kh != 4 || kw != 4 || C % 16) return false;
  return (H == 56 && Ho == 29) || (H == 29 && Ho == 15) || (H == 15 && Ho == 8) || (H == 8 && Ho == 5);
}

int launch_gate_block(const FusedBlockArgs &f, hipStream_t s) {
  if (f.last && !f.idx) {
    set_error("gate_block: a last block needs its branch-dword buffer");
    return TTNET_E_INVALID;
  }
#define TT_FUSED_CASE(h, ho)                                                         \
  if (f.H == h && f.Ho == ho) return f.last ? launch_block_t<h, ho, true>(f, s) : launch_block_t<h, ho, false>(f, s)
  if (f.H == 56 && f.Ho == 29 && !f.last) return launch_block_t<56, 29, false>(f, s);     // (the first block is never the last)
  TT_FUSED_CASE(29, 15);
  TT_FUSED_CASE(15, 8);
  TT_FUSED_CASE(8, 5);
#undef TT_FUSED_CASE
  set_error("gate_block: no fused kernel for %dx%d -> %dx%d", f.H, f.H, f.Ho, f.Ho);
  return TTNET_E_UNSUPPORTED;
}

int launch_fused_images(const void *t_dw1, const void *t_dw2, const void *t_c3, int C, void *img_dw, void *img_c3, hipStream_t s) {
  {
    const size_t t = (size_t)(C / 8) * 65536;
    hipLaunchKernelGGL(c3_image_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, (const uint16_t *)t_c3, (uint8_t *)img_c3,
                       C / 8);
  }
  {
    const size_t t = (size_t)C * 2048 * 2;
    hipLaunchKernelGGL(dw_image_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, (const uint32_t *)t_dw1,
                       (const uint32_t *)t_dw2, (uint32_t *)img_dw, C);
  }
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_branch_rows(const uint32_t *idx, uint64_t *rows, int n, int C, int Ho, int branch, hipStream_t s) {
  const size_t t = (size_t)n * C * Ho;
  hipLaunchKernelGGL(branch_rows_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, idx, rows, n, C, Ho, branch);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_widen_rows(const void *src, uint64_t *dst, size_t count, int W, hipStream_t s) {
  const unsigned g = (unsigned)((count + 255) / 256);
  if (W > 32) hipLaunchKernelGGL(widen_rows_kernel<uint64_t>, dim3(g), dim3(256), 0, s, (const uint64_t *)src, dst, count);
  else if (W > 16) hipLaunchKernelGGL(widen_rows_kernel<uint32_t>, dim3(g), dim3(256), 0, s, (const uint32_t *)src, dst, count);
  else hipLaunchKernelGGL(widen_rows_kernel<uint16_t>, dim3(g), dim3(256), 0, s, (const uint16_t *)src, dst, count);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_narrow_rows(const uint64_t *src, void *dst, size_t count, int W, hipStream_t s) {
  const unsigned g = (unsigned)((count + 255) / 256);
  if (W > 32) hipLaunchKernelGGL(narrow_rows_kernel<uint64_t>, dim3(g), dim3(256), 0, s, src, (uint64_t *)dst, count);
  else if (W > 16) hipLaunchKernelGGL(narrow_rows_kernel<uint32_t>, dim3(g), dim3(256), 0, s, src, (uint32_t *)dst, count);
  else hipLaunchKernelGGL(narrow_rows_kernel<uint16_t>, dim3(g), dim3(256), 0, s, src, (uint16_t *)dst, count);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
