// Gate path: the binarised 4-branch blocks evaluated as truth-table lookups on packed bits.
//
// Replaces Block_resnet_multihead_general_BN_vf_imgnet_v2small.forward
// (models/TT_general_imagenet_v2_small.py:78-148) and the Block_TT.forward calls inside it
// (models/TT_FHE_SMALL.py:307-320).  All work here is integer / bitwise and bit exact.
//
// Data layout in HBM (include/ttnet.h): depthwise windows read row-packed planes
// (uint64 per image row); the grouped 1x1 blocks read group-planar channel words
// (uint16 = the 16 input channels of one group = the table index itself).
//
// Two launches per block:
//   stage 1  heterogeneous workgroups: "dw" units evaluate Block_conv1 / Block_conv2 for 16
//            channels of one branch, "pw" units evaluate Block_conv3 + both 2x2 majorities
//            for one 16-channel group.  Every unit keeps its 128 KiB of tables in LDS
//            (staged with direct global->LDS loads) and walks a slice of the batch.
//   stage 2  Block_convf over the four branches (two 64 KiB groups per workgroup), emitting
//            the next block's input in both layouts (words directly, rows by ballot).
//
// Bound: HBM nominally (packed activations + tables once per batch, SURVEY 8(d):
// 74,592 B/image + 14.2 MB); in practice index-forming VALU + LDS gather issue.

#include <stdlib.h>
#include <type_traits>

#include "ttnet_common.h"

namespace ttnet {

namespace {

constexpr int kGateThreads = 1024;
constexpr int kTableLds = 131072;

// Linear async copy global -> LDS, 16 B per lane per instruction (global_load_lds_dwordx4);
// bytes is a multiple of 1024.  Caller waits with wait_lds_stage() before reading.
__device__ inline void stage_lds_async(uint8_t *lds, const uint8_t *src, int bytes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int chunk = wave; chunk < bytes / 1024; chunk += nwaves) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)chunk * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void *)(lds + chunk * 1024), 16, 0, 0);
  }
}
__device__ inline void wait_lds_stage() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__device__ inline uint32_t maj4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  // at least two of four set, bitwise (act(AvgPool2d(2)(x) - 0.5), :93-94)
  return (a & b) | (c & d) | ((a | b) & (c | d));
}

// 4-bit (KW-bit) field of a padded row (lo,hi) at constant bit offset SH
template <int SH, int KW>
__device__ inline uint32_t row_field(uint32_t lo, uint32_t hi) {
  constexpr uint32_t mask = (1u << KW) - 1u;
  if constexpr (SH + KW <= 32) return (lo >> SH) & mask;
  else if constexpr (SH >= 32) return (hi >> (SH - 32)) & mask;
  else return __builtin_amdgcn_alignbit(hi, lo, SH) & mask;
}

// One output row per lane, all WO columns, 16 columns at a time: first the window indices and
// table reads of the chunk are issued (independent ds_reads in flight), then the bits are
// balloted column by column.  (Used by the stride-1 blocks of --layers 3/4: 57 / 30 columns.)
template <int KH, int KW, int STRIDE, int WO>
__device__ inline void dw_row(const uint32_t (&lo)[KH], const uint32_t (&hi)[KH], const uint32_t *tab32, uint32_t c,
                              uint32_t &keep_lo, uint32_t &keep_hi) {
  static_for<0, (WO + 15) / 16>([&](auto chunk) {
    constexpr int C0 = decltype(chunk)::value * 16, N = (WO - C0) < 16 ? (WO - C0) : 16;
    uint32_t word[N], sel[N];
    static_for<0, N>([&](auto i) {
      constexpr int I = decltype(i)::value, OX = C0 + I;
      uint32_t idx = 0;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) idx |= row_field<OX * STRIDE, KW>(lo[kh], hi[kh]) << (kh * KW);
      sel[I] = idx;
      word[I] = tab32[((idx >> 5) << 4) + c];      // striped table: dword w of channel c at [w*16 + c]
    });
    static_for<0, N>([&](auto i) {
      constexpr int I = decltype(i)::value, OX = C0 + I;
      const uint64_t m = __ballot((word[I] >> (sel[I] & 31)) & 1u);
      writelane64<OX>(keep_lo, keep_hi, m);
    });
  });
}

// ---- fast path for 4x4 / stride 2 / pad 2 windows (the small model) -----------------------
// A lane owns one (channel, output row).  Columns are evaluated in pairs (p, p+8): column
// p+8 is the same window 16 bits further along the rows, so with the four padded rows
// pre-shifted by 4*kh one funnel shift per row yields the nibbles of BOTH windows, already
// in place, in the low and high half of a register:
//     idx2 = (A0 & 0x000F000F) | (A1 & 0x00F000F0) | (A2 & 0x0F000F00) | (A3 & 0xF000F000)
// Each lane accumulates its own output row (bit x = column x); the 16 lanes of a channel
// group then transpose their 16x16 bit blocks in registers (4 butterfly stages, both
// halves at once) so that lane j holds the channel words of columns j and 16+j.
template <int P, bool HIGH_HALF>
__device__ inline void dw_pair(const uint32_t (&d)[4][3], const uint32_t *tab32_bytes, uint32_t c4, uint32_t &acc) {
  // shift of the pair base column P: 2P bits; funnel over dwords (j, j+1)
  constexpr int SH = 2 * P, J = SH / 32, R = SH % 32;
  uint32_t idx2;
  {
    const uint32_t a0 = R ? __builtin_amdgcn_alignbit(d[0][J + 1], d[0][J], R) : d[0][J];
    const uint32_t a1 = R ? __builtin_amdgcn_alignbit(d[1][J + 1], d[1][J], R) : d[1][J];
    const uint32_t a2 = R ? __builtin_amdgcn_alignbit(d[2][J + 1], d[2][J], R) : d[2][J];
    const uint32_t a3 = R ? __builtin_amdgcn_alignbit(d[3][J + 1], d[3][J], R) : d[3][J];
    idx2 = (a0 & 0x000F000Fu) | (a1 & 0x00F000F0u) | (a2 & 0x0F000F00u) | (a3 & 0xF000F000u);
  }
  const uint8_t *tb = (const uint8_t *)tab32_bytes;
  // striped table: dword (idx>>5) of channel c at byte ((idx>>5)*16 + c)*4
  const uint32_t w_lo = *(const uint32_t *)(tb + (((idx2 & 0xFFE0u) << 1) | c4));
  acc |= ((w_lo >> (idx2 & 31u)) & 1u) << P;
  if constexpr (HIGH_HALF) {
    const uint32_t w_hi = *(const uint32_t *)(tb + (((idx2 >> 15) & 0x1FFC0u) | c4));
    acc |= ((w_hi >> ((idx2 >> 16) & 31u)) & 1u) << (P + 8);
  }
}

// rows: the lane's four input rows (bit x = column x); returns the channel words of columns
// (lane&15) [low half] and 16 + (lane&15) [high half] of this lane's output row.
template <int WO>
__device__ inline uint32_t dw_row_4x4s2(const uint64_t (&rows)[4], const uint32_t *tab32, const DwLaneConst &k) {
  uint32_t d[4][3];
#pragma unroll
  for (int kh = 0; kh < 4; ++kh) {
    // padded row (2 zero columns on the left) shifted left by 4*kh: up to 60 + 2 + 12 = 74 bits
    const uint32_t lo = (uint32_t)rows[kh], hi = (uint32_t)(rows[kh] >> 32);
    const int sh = 2 + 4 * kh;
    d[kh][0] = lo << sh;
    d[kh][1] = __builtin_amdgcn_alignbit(hi, lo, 32 - sh);
    d[kh][2] = hi >> (32 - sh);
  }
  uint32_t acc = 0;
  static_for<0, 8>([&](auto p) { dw_pair<decltype(p)::value, (WO > 8)>(d, tab32, k.c4, acc); });
  if constexpr (WO > 16) static_for<16, 24>([&](auto p) { dw_pair<decltype(p)::value, (WO > 24)>(d, tab32, k.c4, acc); });
  // 16x16 bit transpose across the 16 lanes of the channel group, both halves at once
  constexpr int S[4] = {8, 4, 2, 1};
  static_for<0, 4>([&](auto i) {
    constexpr int I = decltype(i)::value;
    const uint32_t partner = (uint32_t)__builtin_amdgcn_ds_swizzle((int)acc, 0x1F | (S[I] << 10));
    const uint32_t moved = __builtin_amdgcn_alignbit(partner, partner, k.rot[I]);
    acc = moved ^ ((moved ^ acc) & k.keep[I]);
  });
  return acc;
}

// Both depthwise branches at once.  Block_conv1 and Block_conv2 read the same input windows, so
// the window index (the expensive part: four funnel shifts and four masked merges per column pair)
// is formed once and looked up in both tables.  The LDS table set of such a unit is striped as
// slot = 8 * branch + (channel & 7): the dwords of conv1 and conv2 for one index sit 32 bytes apart
// and come with one ds_read2_b32.
template <int P, bool HIGH_HALF>
__device__ inline void dw_pair2(const uint32_t (&d)[4][3], const uint32_t *tab32_bytes, uint32_t c4, uint32_t &acc1,
                                uint32_t &acc2) {
  constexpr int SH = 2 * P, J = SH / 32, R = SH % 32;
  uint32_t idx2;
  {
    const uint32_t a0 = R ? __builtin_amdgcn_alignbit(d[0][J + 1], d[0][J], R) : d[0][J];
    const uint32_t a1 = R ? __builtin_amdgcn_alignbit(d[1][J + 1], d[1][J], R) : d[1][J];
    const uint32_t a2 = R ? __builtin_amdgcn_alignbit(d[2][J + 1], d[2][J], R) : d[2][J];
    const uint32_t a3 = R ? __builtin_amdgcn_alignbit(d[3][J + 1], d[3][J], R) : d[3][J];
    idx2 = (a0 & 0x000F000Fu) | (a1 & 0x00F000F0u) | (a2 & 0x0F000F00u) | (a3 & 0xF000F000u);
  }
  const uint8_t *tb = (const uint8_t *)tab32_bytes;
  {
    const uint32_t *row = (const uint32_t *)(tb + (((idx2 & 0xFFE0u) << 1) | c4));
    const uint32_t w1 = row[0], w2 = row[8];
    acc1 |= ((w1 >> (idx2 & 31u)) & 1u) << P;
    acc2 |= ((w2 >> (idx2 & 31u)) & 1u) << P;
  }
  if constexpr (HIGH_HALF) {
    const uint32_t *row = (const uint32_t *)(tb + (((idx2 >> 15) & 0x1FFC0u) | c4));
    const uint32_t w1 = row[0], w2 = row[8], sh = idx2 >> 16;
    acc1 |= ((w1 >> (sh & 31u)) & 1u) << (P + 8);
    acc2 |= ((w2 >> (sh & 31u)) & 1u) << (P + 8);
  }
}

// rows: the lane's four input rows.  Lane = (channel & 7, row slot); a 16-lane group is 8 channels x
// two consecutive output rows.  Returns, for branch b, the transposed word of lane j = lane & 15:
// byte 0 / 1 = the 8 channels of column j in the even / odd row of the group, bytes 2 / 3 the same
// for column 16 + j.
template <int WO>
__device__ inline void dw_rows_4x4s2_both(const uint64_t (&rows)[4], const uint32_t *tab32, const DwLaneConst &k,
                                          uint32_t &out1, uint32_t &out2) {
  uint32_t d[4][3];
#pragma unroll
  for (int kh = 0; kh < 4; ++kh) {
    const uint32_t lo = (uint32_t)rows[kh], hi = (uint32_t)(rows[kh] >> 32);
    const int sh = 2 + 4 * kh;
    d[kh][0] = lo << sh;
    d[kh][1] = __builtin_amdgcn_alignbit(hi, lo, 32 - sh);
    d[kh][2] = hi >> (32 - sh);
  }
  uint32_t acc1 = 0, acc2 = 0;
  static_for<0, 8>([&](auto p) { dw_pair2<decltype(p)::value, (WO > 8)>(d, tab32, k.c4, acc1, acc2); });
  if constexpr (WO > 16) static_for<16, 24>([&](auto p) { dw_pair2<decltype(p)::value, (WO > 24)>(d, tab32, k.c4, acc1, acc2); });
  out1 = transpose16(acc1, k);
  out2 = transpose16(acc2, k);
}

// ---- stage 1: depthwise Block_conv1/2 units and Block_conv3+majority units --------------------
// 1-D grid: blocks [0, n_dw*slices_dw) are depthwise units (unit u = b % n_dw, slice = b / n_dw)
// and are dispatched first; the remaining blocks are conv3 units (q = b' % n_pw, slice = b' / n_pw).
// dw, 4x4 / stride 2 (the common case): unit = (group q = u>>1, half = u&1) = 8 channels x BOTH
// branches; lane = (channel & 7, row slot); each lane walks one output row of its channel for
// conv1 and conv2 with one window index per column pair (dw_pair2) and writes one byte (its 8
// channels) of the output words.  Other geometries (stride 1): unit = (q, branch), lane =
// (channel c = lane&15, row slot = lane>>4), output words by ballot (dw_row).
template <int KH, int KW, int STRIDE, int PAD, int H, int HO>
__global__ __launch_bounds__(kGateThreads) void gate_stage1_kernel(GateBlockArgs a, int n_dw, int dw_blocks, int slices_dw,
                                                                  int slices_pw) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int W = H, WO = HO;
  const int Q = a.C / 16;
  // (the wave index is forced into an SGPR: everything derived from it -- task, image, row group --
  // is then scalar arithmetic instead of per-lane VALU work)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = kGateThreads / 64;

  if ((int)blockIdx.x < dw_blocks) {
    const int unit = blockIdx.x % n_dw;
    const int sl = blockIdx.x / n_dw;                   // batch slices of (almost) equal size
    const int n0 = (int)((long long)sl * a.n / slices_dw);
    if (n0 >= a.n) return;
    const int n1 = (int)((long long)(sl + 1) * a.n / slices_dw);
    if constexpr (KH == 4 && KW == 4 && STRIDE == 2 && PAD == 2 && WO <= 32) {
      // unit = (group q, half h): 8 channels, BOTH branches (see dw_pair2)
      const int q = unit >> 1, half = unit & 1;
      {   // LDS block i (16 bytes = 4 slots) of striped row w = i >> 2: slots 0-7 from conv1's table, 8-15 from conv2's
        const uint8_t *t1 = a.t_dw1 + (size_t)q * kTableLds + 32 * half, *t2 = a.t_dw2 + (size_t)q * kTableLds + 32 * half;
        for (int chunk = wave; chunk < kTableLds / 1024; chunk += nwaves) {
          const int i = chunk * 64 + lane, w = i >> 2, part = i & 3;
          const uint8_t *src = (part < 2 ? t1 : t2) + (size_t)w * 64 + 16 * (part & 1);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                           (__attribute__((address_space(3))) void *)(lds + chunk * 1024), 16, 0, 0);
        }
      }
      const uint32_t ch8 = lane & 7, slot = lane >> 3;
      constexpr int rows8 = (HO + 7) / 8;
      const int tasks = (n1 - n0) * rows8;
      auto load_rows = [&](int t, uint64_t (&r)[KH]) {
        const int n = n0 + t / rows8, oy = (t % rows8) * 8 + (int)slot;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          const int iy = oy * STRIDE - PAD + kh;
          r[kh] = (t < tasks && oy < HO && iy >= 0 && iy < H) ? a.x_rp[((size_t)n * a.C + 16 * q + 8 * half + ch8) * H + iy] : 0ull;
        }
      };
      uint64_t cur[KH], nxt[KH];
      load_rows(wave, cur);
      DwLaneConst lk = dw_lane_const(lane);
      lk.c4 = ch8 << 2;
      wait_lds_stage();
      const uint32_t *tab32 = (const uint32_t *)lds;
      const int col = lane & 15, rsub = 2 * (lane >> 4);
      for (int t = wave; t < tasks; t += nwaves) {
        load_rows(t + nwaves, nxt);                       // prefetch the next task's rows
        const int n = n0 + t / rows8, row = (t % rows8) * 8 + rsub;
        uint32_t w1, w2;
        dw_rows_4x4s2_both<WO>(cur, tab32, lk, w1, w2);
        // lane j = column j (and 16 + j) of rows `row`, `row + 1`: one byte (this unit's 8 channels) of each word
        const size_t at = ((((size_t)n * Q + q) * HO + row) * WO + col) * 2 + half;
        uint8_t *d1 = (uint8_t *)a.o1 + at, *d2 = (uint8_t *)a.o2 + at;
        if (row < HO && col < WO) {
          d1[0] = (uint8_t)w1;
          d2[0] = (uint8_t)w2;
          if (row + 1 < HO) {
            d1[2 * WO] = (uint8_t)(w1 >> 8);
            d2[2 * WO] = (uint8_t)(w2 >> 8);
          }
          if (16 + col < WO) {
            d1[32] = (uint8_t)(w1 >> 16);
            d2[32] = (uint8_t)(w2 >> 16);
            if (row + 1 < HO) {
              d1[2 * WO + 32] = (uint8_t)(w1 >> 24);
              d2[2 * WO + 32] = (uint8_t)(w2 >> 24);
            }
          }
        }
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) cur[kh] = nxt[kh];
      }
    } else {
    const int q = unit >> 1, branch = unit & 1;
    const uint8_t *tab = (branch ? a.t_dw2 : a.t_dw1) + (size_t)q * kTableLds;
    uint16_t *out = branch ? a.o2 : a.o1;
    stage_lds_async(lds, tab, kTableLds);

    const uint32_t c = lane & 15, slot = lane >> 4;
    constexpr int rows4 = (HO + 3) / 4;
    const int tasks = (n1 - n0) * rows4;
    auto load_rows = [&](int t, uint64_t (&r)[KH]) {
      const int n = n0 + t / rows4, oy = (t % rows4) * 4 + (int)slot;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        const int iy = oy * STRIDE - PAD + kh;
        r[kh] = (t < tasks && oy < HO && iy >= 0 && iy < H) ? a.x_rp[((size_t)n * a.C + 16 * q + c) * H + iy] : 0ull;
      }
    };
    uint64_t cur[KH], nxt[KH];
    load_rows(wave, cur);
    wait_lds_stage();
    const uint32_t *tab32 = (const uint32_t *)lds;
    for (int t = wave; t < tasks; t += nwaves) {
      load_rows(t + nwaves, nxt);                       // prefetch the next task's rows
      const int n = n0 + t / rows4, oyb = (t % rows4) * 4;
      const bool valid = oyb + (int)slot < HO;
      const uint64_t vmask = __ballot(valid);
      uint32_t lo[KH], hi[KH];
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        const uint64_t rp = cur[kh] << PAD;
        lo[kh] = (uint32_t)rp;
        hi[kh] = (uint32_t)(rp >> 32);
      }
      uint32_t keep_lo = 0, keep_hi = 0;
      dw_row<KH, KW, STRIDE, WO>(lo, hi, tab32, c, keep_lo, keep_hi);
      if (lane < WO) {
        const uint64_t keep = (((uint64_t)keep_hi << 32) | keep_lo) & vmask;   // rows past HO: no bits
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (oyb + s < HO) out[(((size_t)n * Q + q) * HO + oyb + s) * WO + lane] = (uint16_t)(keep >> (16 * s));
      }
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) cur[kh] = nxt[kh];
    }
    }
  } else {
    // Block_conv3 (16 -> 16 bits per pixel and group) + majority pools of conv3(x) and of x
    const int b = blockIdx.x - dw_blocks;
    const int q = b % Q;
    const int sl = b / Q;
    const int n0 = (int)((long long)sl * a.n / slices_pw);
    if (n0 >= a.n) return;
    const int n1 = (int)((long long)(sl + 1) * a.n / slices_pw);
    stage_lds_async(lds, (const uint8_t *)(a.t_c3 + (size_t)q * 65536), kTableLds);
    wait_lds_stage();
    const uint16_t *tab = (const uint16_t *)lds;
    if constexpr (STRIDE == 1) {
      // stride-1 block (:95-96): out3 = conv3(x), out4 = x, both at the input size, written at the
      // branch-padding offset (the zero border of the HO x WO planes is never touched)
      constexpr int per = H * W, U = 4;
      const int tasks = (n1 - n0) * per;
      for (int t0 = threadIdx.x; t0 < tasks; t0 += U * kGateThreads) {
        uint32_t w[U];
        size_t dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int t = t0 + u * kGateThreads;
          const bool live = t < tasks;
          const int tt = live ? t : 0;
          const int n = n0 + tt / per, r = tt % per, y = r / W, x = r % W;
          w[u] = a.x_cp[(((size_t)n * Q + q) * H + y) * W + x];
          dst[u] = live ? (((size_t)n * Q + q) * HO + y + a.off34) * WO + x + a.off34 : (size_t)-1;
        }
        uint32_t r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = tab[w[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (dst[u] != (size_t)-1) {
            a.o3[dst[u]] = (uint16_t)r[u];
            a.o4[dst[u]] = (uint16_t)w[u];
          }
        }
      }
      return;
    }
    constexpr int HP = H / 2, WP = W / 2, per = HP * WP, U = 4;
    const int tasks = (n1 - n0) * per;
    // U pooled pixels per thread and trip: 4U global loads, then 4U table reads, in flight together
    for (int t0 = threadIdx.x; t0 < tasks; t0 += U * kGateThreads) {
      uint32_t w[U][4];
      size_t dst[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * kGateThreads;
        const bool live = t < tasks;
        const int tt = live ? t : 0;
        const int n = n0 + tt / per, r = tt % per, py = r / WP, px = r % WP;
        const uint16_t *src = a.x_cp + (((size_t)n * Q + q) * H + 2 * py) * W + 2 * px;
        w[u][0] = src[0]; w[u][1] = src[1]; w[u][2] = src[W]; w[u][3] = src[W + 1];
        dst[u] = live ? (((size_t)n * Q + q) * HO + py + a.off34) * WO + px + a.off34 : (size_t)-1;
      }
      uint32_t r[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int d = 0; d < 4; ++d) r[u][d] = tab[w[u][d]];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (dst[u] != (size_t)-1) {
          a.o3[dst[u]] = (uint16_t)maj4(r[u][0], r[u][1], r[u][2], r[u][3]);
          a.o4[dst[u]] = (uint16_t)maj4(w[u][0], w[u][1], w[u][2], w[u][3]);
        }
      }
    }
  }
}

// ---- stage 2: Block_convf of a binarised block ----------------------------------------------
// The reference interleaves to channel 4c+branch (:144-147) and groups 16 of those: group g
// reads channels 4g..4g+3 of each branch.  Internal index = nib(out1) | nib(out2)<<4 |
// nib(out3)<<8 | nib(out4)<<12, each nibble LSB = channel 4g.  Two groups (2 x 64 KiB of
// 8-bit entries) per workgroup produce one output channel word; the row layout of the same
// 16 channels comes from 16 ballots (a wave covers whole image rows).
// CG = output bits per convf group: 8 (Cout = 2C: two groups fill one 16-channel output word), or 4
// for the stride-1 block of --layers 4 whose convf keeps the channel count (Cout = C: the two
// groups of a workgroup fill one byte of an output word).
template <int HO, int CG = 8>
__global__ __launch_bounds__(kGateThreads) void gate_pf_kernel(GateBlockArgs a, const uint8_t *t_cf, uint16_t *out_cp,
                                                              uint64_t *out_rp, int slices) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int WO = HO;
  constexpr int LPR = WO <= 16 ? 16 : (WO <= 32 ? 32 : 64), RPW = 64 / LPR, chunks = (HO + RPW - 1) / RPW;
  const int j = blockIdx.x;              // output word; groups 2j, 2j+1
  const int Q = a.C / 16, Cout = CG * (a.C / 4), Qout = Cout / 16;
  const int n0 = (int)((long long)blockIdx.y * a.n / slices);
  if (n0 >= a.n) return;
  const int n1 = (int)((long long)(blockIdx.y + 1) * a.n / slices);
  stage_lds_async(lds, t_cf + (size_t)(2 * j) * 65536, kTableLds);
  const int wq = j >> 1, sh = 8 * (j & 1);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = kGateThreads / 64;
  const int sub = lane / LPR, ox = lane % LPR;
  const int lane_off = sub * WO + ox;                   // the lane's pixel inside a chunk of RPW rows
  const DwLaneConst lk = dw_lane_const(lane);
  const int tasks = (n1 - n0) * chunks;
  auto load_words = [&](int t, uint32_t (&b)[4], size_t &pix, bool &valid) {
    const int n = n0 + t / chunks, oyb = (t % chunks) * RPW;       // wave-uniform
    valid = t < tasks && oyb + sub < HO && ox < WO;
    const size_t base = (((size_t)n * Q + wq) * HO + oyb) * WO;     // uniform part of the address
    pix = base + lane_off;
    b[0] = valid ? (a.o1 + base)[lane_off] : 0; b[1] = valid ? (a.o2 + base)[lane_off] : 0;
    b[2] = valid ? (a.o3 + base)[lane_off] : 0; b[3] = valid ? (a.o4 + base)[lane_off] : 0;
  };
  // The per-task work (two lookups, 16 ballots) is shorter than a global-load round trip, so
  // the four input words are fetched PF tasks ahead (a register ring; the loop is unrolled by PF
  // so that the ring index is static).
  constexpr int PF = 4;
  uint32_t ring[PF][4];
  size_t ring_pix[PF];
  bool ring_valid[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) load_words(wave + d * nwaves, ring[d], ring_pix[d], ring_valid[d]);
  wait_lds_stage();
  for (int t0 = wave; t0 < tasks; t0 += PF * nwaves) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
    const int t = t0 + d * nwaves;
    if (t >= tasks) break;                               // wave-uniform
    uint32_t cur[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = ring[d][i];
    const bool valid = ring_valid[d];
    load_words(t + PF * nwaves, ring[d], ring_pix[d], ring_valid[d]);      // refill this slot
    const int n = n0 + t / chunks;
    uint32_t r = 0;
    if (valid) {
      const uint32_t b1 = (cur[0] >> sh) & 0xFF, b2 = (cur[1] >> sh) & 0xFF;
      const uint32_t b3 = (cur[2] >> sh) & 0xFF, b4 = (cur[3] >> sh) & 0xFF;
      const uint32_t i0 = (b1 & 15) | ((b2 & 15) << 4) | ((b3 & 15) << 8) | ((b4 & 15) << 12);
      const uint32_t i1 = (b1 >> 4) | ((b2 >> 4) << 4) | ((b3 >> 4) << 8) | ((b4 >> 4) << 12);
      const int oyb = (t % chunks) * RPW;
      if constexpr (CG == 8) {
        r = lds[i0] | ((uint32_t)lds[65536 + i1] << 8);
        (out_cp + (((size_t)n * Qout + j) * HO + oyb) * WO)[lane_off] = (uint16_t)r;
      } else {
        r = (lds[i0] & 15u) | (((uint32_t)lds[65536 + i1] & 15u) << 4);
        ((uint8_t *)out_cp + ((((size_t)n * Qout + (j >> 1)) * HO + oyb) * WO) * 2 + (j & 1))[2 * lane_off] = (uint8_t)r;
      }
    }
    // rows of the next block's row-packed input: transpose the (pixel lane) x (channel bit) words of
    // every 16-lane group in registers, then lane j < 2*CG holds channel j's bits of the group's 16
    // pixels; the pieces of a row's groups are joined across lanes.  (16 ballots + 32 v_writelane
    // per task before: four times the instructions.)
    const uint32_t piece = transpose16(r, lk) & 0xFFFFu;
    uint64_t rowbits;
    if constexpr (LPR == 16) {
      rowbits = piece;
    } else if constexpr (LPR == 32) {
      const uint32_t other = (uint32_t)__shfl_xor((int)piece, 16);
      rowbits = piece | (other << 16);                   // used by the lanes with (lane & 16) == 0
    } else {
      const int j = lane & 15;
      const uint32_t p1 = (uint32_t)__shfl((int)piece, j + 16), p2 = (uint32_t)__shfl((int)piece, j + 32),
                     p3 = (uint32_t)__shfl((int)piece, j + 48);
      rowbits = (uint64_t)(piece | (p1 << 16)) | ((uint64_t)(p2 | (p3 << 16)) << 32);   // lanes 0-15
    }
    if ((lane & (LPR - 1)) < 2 * CG) {
      const int oy = (t % chunks) * RPW + sub;
      if (oy < HO) out_rp[((size_t)n * Cout + 2 * CG * j + (lane & 15)) * HO + oy] = rowbits;
    }
    }
  }
}

// ---- Block_convf of the LAST block: float outputs through a 4 MiB-per-group table ---------
// relu(bn2(conv2(gelu(bn1(conv1(bits)))))) for 16 input bits -> 16 floats is a 64-byte
// row; the table for all 64 groups is 256 MiB and sits in HBM / Infinity Cache.  The
// following AvgPool2d(2) (:197) is fused: four rows are gathered and averaged.
// 16 lanes share one (image, group, pooled pixel) and read one 64-byte row together.
// table index of Block_convf group g at a pixel from the branch dword of gate_fused.hip (strand g >> 1:
// bytes (0,2) are the index of its even group, bytes (1,3) of its odd group)
__device__ inline uint32_t last_index_from_dword(uint32_t d, int g) {
  return __builtin_amdgcn_perm(0u, d, (g & 1) ? 0x0C0C0301u : 0x0C0C0200u);
}

__global__ __launch_bounds__(256) void gate_last_kernel(GateBlockArgs a, const float *t_last, uint16_t *feat_frag,
                                                        uint32_t *range_flag, const uint32_t *__restrict__ bidx) {
  const int Q = a.C / 16, G = a.C / 4;
  const int Hp = a.Ho / 2, Wp = a.Wo / 2, PP = Hp * Wp;
  const size_t task = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int k = threadIdx.x & 15;
  const size_t total = (size_t)a.n * G * PP;
  if (task >= total) return;
  const int pp = task % PP, g = (task / PP) % G, n = task / ((size_t)PP * G);
  const int py = pp / Wp, px = pp % Wp;
  const int wq = g >> 2, sh = 4 * (g & 3);
  const float *tab = t_last + (size_t)g * 65536 * 16;
  float v[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t idx;
    if (bidx) {
      idx = last_index_from_dword(bidx[((size_t)n * (a.C / 8) + (g >> 1)) * a.Ho * a.Wo + (2 * py + (d >> 1)) * a.Wo + 2 * px + (d & 1)], g);
    } else {
      const size_t pix = (((size_t)n * Q + wq) * a.Ho + 2 * py + (d >> 1)) * a.Wo + 2 * px + (d & 1);
      idx = ((a.o1[pix] >> sh) & 15) | (((a.o2[pix] >> sh) & 15) << 4) | (((a.o3[pix] >> sh) & 15) << 8) |
            (((a.o4[pix] >> sh) & 15) << 12);
    }
    v[d] = tab[(size_t)idx * 16 + k];
  }
  const float f = (((v[0] + v[1]) + v[2]) + v[3]) * 0.25f;
  // fp16 x 2 split, stored where lin1's MFMA fragments expect it: row = image, k-step = g*PP + pp, k
  store_feature(feat_frag, n, G * PP, g * PP + pp, k, f, range_flag);
}

// The same for the 8x8 -> 4x4 geometry of the default network: one workgroup per (channel word,
// image), thread = (pooled pixel, k).  The table indices of the 8x8 plane are formed once and
// shared through LDS; what is left per thread is the 64-byte-row gathers.
// (The generic kernel above spends ~220 VALU instructions per wave on index arithmetic and is
// VALU bound: 28 us at B = 256.)
template <bool NT>
__global__ __launch_bounds__(256) void gate_last8_kernel(GateBlockArgs a, const float *__restrict__ t_last,
                                                         uint16_t *__restrict__ feat_frag, uint32_t *range_flag,
                                                         const uint32_t *__restrict__ bidx) {
  // one workgroup = the four groups that share a channel word (g = 4 wq + s) of one image: wave s
  // forms the 64 table indices of group s (one pixel per lane), then every thread (pooled pixel, k)
  // gathers its four rows for each of the four groups -- sixteen 64-byte-row reads in flight
  __shared__ uint32_t s_off[4][64];
  const int wq = blockIdx.x, n = blockIdx.y, G = 4 * gridDim.x, Q = a.C / 16;
  {
    const int sgrp = threadIdx.x >> 6, pixl = threadIdx.x & 63, sh = 4 * sgrp;
    uint32_t idx;
    if (bidx) {
      idx = last_index_from_dword(bidx[((size_t)n * (2 * Q) + 2 * wq + (sgrp >> 1)) * 64 + pixl], sgrp);
    } else {
      const size_t pix = ((size_t)n * Q + wq) * 64 + pixl;
      idx = ((a.o1[pix] >> sh) & 15) | (((a.o2[pix] >> sh) & 15) << 4) | (((a.o3[pix] >> sh) & 15) << 8) |
            (((a.o4[pix] >> sh) & 15) << 12);
    }
    s_off[sgrp][pixl] = idx * 16;
  }
  __syncthreads();
  const int k = threadIdx.x & 15, pp = threadIdx.x >> 4;
  const int p00 = 16 * (pp >> 2) + 2 * (pp & 3);            // pixel (2py, 2px) of the 8x8 plane
  float v[4][4];
#pragma unroll
  for (int sgrp = 0; sgrp < 4; ++sgrp) {
    const float *tab = t_last + (size_t)(4 * wq + sgrp) * 65536 * 16 + k;
    auto ld = [&](uint32_t off) { return NT ? __builtin_nontemporal_load(tab + off) : tab[off]; };
    v[sgrp][0] = ld(s_off[sgrp][p00]);
    v[sgrp][1] = ld(s_off[sgrp][p00 + 1]);
    v[sgrp][2] = ld(s_off[sgrp][p00 + 8]);
    v[sgrp][3] = ld(s_off[sgrp][p00 + 9]);
  }
#pragma unroll
  for (int sgrp = 0; sgrp < 4; ++sgrp) {
    const float f = (((v[sgrp][0] + v[sgrp][1]) + v[sgrp][2]) + v[sgrp][3]) * 0.25f;
    store_feature(feat_frag, n, G * 16, (4 * wq + sgrp) * 16 + pp, k, f, range_flag);
  }
}

// ---- layout conversions (parity taps and ttnet_forward_from_stem_bits only) -----------------
__global__ void cp_to_rp_kernel(const uint16_t *cp, uint64_t *rp, int n, int C, int H, int W) {
  const int Q = C / 16;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * Q * H) return;
  const int y = t % H, q = (t / H) % Q, img = t / ((size_t)H * Q);
  uint64_t rows[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) rows[k] = 0;
  for (int x = 0; x < W; ++x) {
    const uint32_t w = cp[(((size_t)img * Q + q) * H + y) * W + x];
#pragma unroll
    for (int k = 0; k < 16; ++k) rows[k] |= (uint64_t)((w >> k) & 1u) << x;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) rp[((size_t)img * C + 16 * q + k) * H + y] = rows[k];
}

__global__ void rp_to_cp_kernel(const uint64_t *rp, uint16_t *cp, int n, int C, int H, int W) {
  const int Q = C / 16;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * Q * H) return;
  const int y = t % H, q = (t / H) % Q, img = t / ((size_t)H * Q);
  uint64_t rows[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) rows[k] = rp[((size_t)img * C + 16 * q + k) * H + y];
  for (int x = 0; x < W; ++x) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) w |= (uint32_t)((rows[k] >> x) & 1ull) << k;
    cp[(((size_t)img * Q + q) * H + y) * W + x] = (uint16_t)w;
  }
}

// batch slices per table set so that `units` table sets spread over at most `target` workgroups
int slices_for(int n, int units, int target) { return std::max(1, std::min(n, target / std::max(1, units))); }

template <typename K>
int allow_big_lds(K kernel, size_t bytes) {
  return ensure_dynamic_lds((const void *)kernel, bytes);
}

template <int H, int HO, int STRIDE = 2>
int launch_stage1_t(const GateBlockArgs &a, hipStream_t s) {
  // 1 workgroup per CU (128 KiB of tables in LDS), and every workgroup pays ~2 us of table
  // staging: the whole launch is one round of the chip, 208 depthwise + 48 conv3 workgroups
  // (measured best split at B = 256; a second round of conv3 blocks cost 3-5 us per launch).
  const int n_dw = (a.C / 16) * 2, n_pw = a.C / 16;
  const int sl_dw = slices_for(a.n, n_dw, 208), sl_pw = slices_for(a.n, n_pw, 48);
  const int dw_blocks = n_dw * sl_dw, pw_blocks = n_pw * sl_pw;
  auto k = gate_stage1_kernel<4, 4, STRIDE, 2, H, HO>;
  TT_TRY(allow_big_lds(k, kTableLds));
  hipLaunchKernelGGL(k, dim3(dw_blocks + pw_blocks), dim3(kGateThreads), kTableLds, s, a, n_dw, dw_blocks, sl_dw, sl_pw);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

template <int HO, int CG = 8>
int launch_pf_t(const GateBlockArgs &a, const uint8_t *t_cf, uint16_t *out_cp, uint64_t *out_rp, hipStream_t s) {
  const int units = a.C / 8;
  const int slices = slices_for(a.n, units, 256);
  auto k = gate_pf_kernel<HO, CG>;
  TT_TRY(allow_big_lds(k, kTableLds));
  hipLaunchKernelGGL(k, dim3(units, slices), dim3(kGateThreads), kTableLds, s, a, t_cf, out_cp, out_rp, slices);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace

int launch_gate_stage1(const GateBlockArgs &a, hipStream_t s) {
  if (a.C % 16 || a.H != a.W || a.Ho != a.Wo || a.kh1 != 4 || a.kw1 != 4 || a.kh2 != 4 || a.kw2 != 4 ||
      (a.stride != 2 && a.stride != 1) || a.pad != 2) {
    set_error("gate_stage1: unsupported geometry C=%d %dx%d k=%dx%d", a.C, a.H, a.W, a.kh1, a.kw1);
    return TTNET_E_UNSUPPORTED;
  }
  if (a.stride == 1) {                     // --layers 3/4 (TT_general_imagenet_v2_small.py:178-181)
    if (a.H == 56 && a.Ho == 57) return launch_stage1_t<56, 57, 1>(a, s);
    if (a.H == 29 && a.Ho == 30) return launch_stage1_t<29, 30, 1>(a, s);
    set_error("gate_stage1: no stride-1 kernel for %dx%d -> %dx%d", a.H, a.W, a.Ho, a.Wo);
    return TTNET_E_UNSUPPORTED;
  }
  if (a.H == 57 && a.Ho == 29) return launch_stage1_t<57, 29>(a, s);
  if (a.H == 30 && a.Ho == 16) return launch_stage1_t<30, 16>(a, s);
  if (a.H == 16 && a.Ho == 9) return launch_stage1_t<16, 9>(a, s);
  if (a.H == 56 && a.Ho == 29) return launch_stage1_t<56, 29>(a, s);
  if (a.H == 29 && a.Ho == 15) return launch_stage1_t<29, 15>(a, s);
  if (a.H == 15 && a.Ho == 8) return launch_stage1_t<15, 8>(a, s);
  if (a.H == 8 && a.Ho == 5) return launch_stage1_t<8, 5>(a, s);
  set_error("gate_stage1: no kernel for %dx%d -> %dx%d", a.H, a.W, a.Ho, a.Wo);
  return TTNET_E_UNSUPPORTED;
}

int launch_gate_pf(const GateBlockArgs &a, const uint8_t *t_cf, uint16_t *out_cp, uint64_t *out_rp, hipStream_t s) {
  if (a.cf_bits == 4) {
    if (a.Ho == 30) return launch_pf_t<30, 4>(a, t_cf, out_cp, out_rp, s);
    set_error("gate_pf: no 4-bit-group kernel for %dx%d", a.Ho, a.Wo);
    return TTNET_E_UNSUPPORTED;
  }
  if (a.cf_bits != 8) {
    set_error("gate_pf: %d output bits per convf group", a.cf_bits);
    return TTNET_E_UNSUPPORTED;
  }
  if (a.Ho == 57) return launch_pf_t<57>(a, t_cf, out_cp, out_rp, s);
  if (a.Ho == 30) return launch_pf_t<30>(a, t_cf, out_cp, out_rp, s);
  if (a.Ho == 16) return launch_pf_t<16>(a, t_cf, out_cp, out_rp, s);
  if (a.Ho == 29) return launch_pf_t<29>(a, t_cf, out_cp, out_rp, s);
  if (a.Ho == 15) return launch_pf_t<15>(a, t_cf, out_cp, out_rp, s);
  if (a.Ho == 8) return launch_pf_t<8>(a, t_cf, out_cp, out_rp, s);
  set_error("gate_pf: no kernel for %dx%d", a.Ho, a.Wo);
  return TTNET_E_UNSUPPORTED;
}

int launch_gate_last(const GateBlockArgs &a, const float *t_last, void *feat_frag, uint32_t *range_flag, hipStream_t s,
                     const uint32_t *idx) {
  if (a.Ho == 8 && a.Wo == 8 && a.n <= 65535) {
    static const bool nt = getenv("TTNET_LAST_NT") && atoi(getenv("TTNET_LAST_NT"));      // (diagnostic: non-temporal table reads)
    if (nt) hipLaunchKernelGGL(gate_last8_kernel<true>, dim3(a.C / 16, a.n), dim3(256), 0, s, a, t_last, (uint16_t *)feat_frag, range_flag, idx);
    else hipLaunchKernelGGL(gate_last8_kernel<false>, dim3(a.C / 16, a.n), dim3(256), 0, s, a, t_last, (uint16_t *)feat_frag, range_flag, idx);
    TT_HIP(hipGetLastError());
    return TTNET_OK;
  }
  const size_t tasks = (size_t)a.n * (a.C / 4) * (a.Ho / 2) * (a.Wo / 2);
  const size_t threads = tasks * 16;
  hipLaunchKernelGGL(gate_last_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, t_last, (uint16_t *)feat_frag,
                     range_flag, idx);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_cp_to_rp(const uint16_t *cp, uint64_t *rp, int n, int C, int H, int W, hipStream_t s) {
  const size_t t = (size_t)n * (C / 16) * H;
  hipLaunchKernelGGL(cp_to_rp_kernel, dim3((unsigned)((t + 127) / 128)), dim3(128), 0, s, cp, rp, n, C, H, W);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_rp_to_cp(const uint64_t *rp, uint16_t *cp, int n, int C, int H, int W, hipStream_t s) {
  const size_t t = (size_t)n * (C / 16) * H;
  hipLaunchKernelGGL(rp_to_cp_kernel, dim3((unsigned)((t + 127) / 128)), dim3(128), 0, s, rp, cp, n, C, H, W);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
