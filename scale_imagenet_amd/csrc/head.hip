// Classifier head: lin1 (no bias) -> BatchNorm1d -> 0.47 + 0.50 x + 0.09 x^2 -> lin2 (+bias)
// (Classifier_scale.forward, models/TT_general_imagenet_v2_small.py:229-236;
//  Polynome_ACT.forward :213-215).
//
// Both linears are C[M][N] = A[M][K] * B[N][K]^T.  The 1e-5 logit tolerance rules out plain
// bf16 operands.
//   lin1 (K = 16384, 94 % of the head's flops) runs on the 16-bit matrix cores with both
//        operands prescaled by a power of two and split into two fp16 terms, the three products
//        of weight >= 2^-11 kept (scheme of stem.hip; on the synthetic model the logits then sit
//        <= 1.6e-6 from the float64 head, the reference's own float32 head 6e-6 .. 1.1e-5):
//        gemm_f16x2_kernel.  Operands are stored pre-split in MFMA fragment order
//        ([tile32][kstep16][plane][lane][8 fp16]) so that every global->LDS transfer and every
//        LDS fragment read is a linear 1 KiB block.
//   lin2 (K = 1000, padded to 1008) uses the same split operands in one small kernel
//        (lin2_f16x2_kernel): 32x32 output tile per workgroup, K split over its 8 waves,
//        partial tiles summed through LDS in wave order, bias fused.
// M = images is small (256), so lin1's K is split across workgroups to fill the 256 CUs; the
// partial slabs are summed in a fixed order by head_mid_kernel (bitwise reproducible, no float
// atomics), which also applies BatchNorm1d + the polynomial and writes lin2's A operand.
//
// Bound: lin1 16-bit MFMA (3 MFMA flops per algorithmic flop) / HBM (66 MB of split weights
// read once per batch); lin2 latency.

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "ttnet_common.h"

namespace ttnet {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- fp16 x 2 split GEMM in fragment order ----------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int NP = SPLIT_PLANES;

// tools/ubench/lin1_parts.hip builds this file with parts of the GEMM switched off (bit mask: 1 no A stream,
// 2 no B stream, 4 no MFMAs, 8 no slab stores).  0 in the library.
#ifndef TT_LIN1_SKIP
#define TT_LIN1_SKIP 0
#endif
constexpr int kLin1Skip = TT_LIN1_SKIP;

#ifndef TT_LIN1_BM
#define TT_LIN1_BM 128
#endif
#ifndef TT_LIN1_WGS
#define TT_LIN1_WGS 256
#endif
constexpr int G_BM = TT_LIN1_BM, G_BN = 128;          // workgroup tile: (4 | 8) x 4 MFMA tiles of 32x32
constexpr int G_MT = G_BM / 32, G_NT = G_BN / 32;
constexpr int G_KS = 1;                               // k-steps (of 16) per LDS stage
constexpr int G_CHUNK = 1024;                         // one fragment block: 64 lanes x 16 B
constexpr int G_STAGE = (G_MT + G_NT) * G_KS * NP * G_CHUNK;  // 16 KiB (128 x 128 tile)
#ifndef TT_LIN1_STAGES
#define TT_LIN1_STAGES 4
#endif
constexpr int G_STAGES = TT_LIN1_STAGES;              // LDS ring: G_STAGES - 1 stages issued ahead.  The kernel is bound by how fast a CU takes its
                                                      // 1 MiB in (about 31 GB/s per CU, 8 TB/s over the chip): in-kernel stamps give 1,000 cycles per
                                                      // k-step at 4 stages, 970 at 6, 870 at 8 (which fills the LDS), against 384 of MFMAs
#ifndef TT_LIN1_WAVES
#define TT_LIN1_WAVES (TT_LIN1_BM / 32)
#endif
constexpr int G_WAVES = TT_LIN1_WAVES;                 // wave w: M-tile w % G_MT, N-tiles (w / G_MT) * NPW .. + NPW - 1
constexpr int G_WN = G_WAVES / G_MT, NPW = G_NT / G_WN;
static_assert(G_WAVES % G_MT == 0 && G_NT % G_WN == 0, "waves tile the workgroup's output");
constexpr int G_CHUNKS = (G_MT + G_NT) * G_KS * NP;   // 16 fragment blocks per stage
// Dedicated loader waves (round 3 experiment, kept as a build option): a direct-to-LDS load costs the wave that issues it
// 60 - 185 cycles of issue (MI355X_MICROARCH.md, cycle constants), four of them per k-step beside twelve MFMAs of 32 cycles; with
// TT_LIN1_LOADERS = 4, waves G_WAVES .. issue every load and nothing else.  Measured (tools/ubench/lin1_parts.hip, stamps): the
// kernel alone 42 -> 40 us, but the forward with two batches in flight 1.79 -> 1.66 M images/s (eight waves per workgroup
// leave less of the CU to the other batch's kernels): 0 = every wave loads, as shipped.
#ifndef TT_LIN1_LOADERS
#define TT_LIN1_LOADERS 0
#endif
constexpr int G_LOADERS = TT_LIN1_LOADERS;
constexpr int G_THREADS = 64 * (G_WAVES + G_LOADERS);
constexpr int G_LW = G_LOADERS ? G_LOADERS : G_WAVES; // waves that load
constexpr int G_LOADS = (G_CHUNKS + G_LW - 1) / G_LW; // direct-to-LDS loads per loading wave and stage (any
                                                      // surplus slots load into a scratch block so every wave counts the same)
constexpr int G_LDS = G_STAGES * G_STAGE + G_CHUNK;

#ifdef TT_LIN1_STAMP
__device__ unsigned long long g_lin1_stamps[512][2];      // (diagnostic builds: shader cycles and ns of the main loop per workgroup)
#endif
// Af: [mtile32][KS][NP][64][8] fp16, Bf: [ntile32][KS][NP][64][8] fp16, part: slabs in accumulator order (epilogue)
template <int AUX_A, int AUX_B>
__global__ __launch_bounds__(G_THREADS) void gemm_f16x2_kernel(const uint8_t *__restrict__ Af, const uint8_t *__restrict__ Bf,
                                                         float *__restrict__ part, int M, int N, int KS, int ks_per,
                                                         int n_tiles, int m_tiles, int splits) {
  extern __shared__ __align__(16) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = G_LOADERS ? wave >= G_WAVES : true, consumer = wave < G_WAVES;
  const int lw = G_LOADERS ? (loader ? wave - G_WAVES : 0) : wave;      // index among the loading waves
  // 1-D grid, XCD-aware: blocks b and b+8 share an XCD (round-robin dispatch), so the N-tiles
  // of one (M-tile, K-slice) -- which all read the same A slice -- are given ids that are equal
  // mod 8: the slice is then fetched into that XCD's L2 once instead of once per N-tile.
  // (Placement only changes speed, never results.)
  int ntile, mtile, slice;
  {
    const int b = blockIdx.x, groups = m_tiles * splits;          // groups of n_tiles blocks share A
    const int lanes8 = groups < 8 ? groups : 8;
    const int round = b / (lanes8 * n_tiles), rem = b % (lanes8 * n_tiles);
    int grp = round * lanes8 + rem % lanes8;
    ntile = rem / lanes8;
    if (round == groups / lanes8) {                               // ragged last round (fewer than 8 groups left): EVERY block of it
      const int left = groups - round * lanes8;                   // takes the narrower interleave (round 2 remapped only the
      grp = round * lanes8 + rem % left;                          // blocks whose group index overflowed: with 4 of 8 groups left,
      ntile = rem / left;                                         // (group, N-tile) pairs were then computed twice and others never)
    }
    mtile = grp / splits;
    slice = grp % splits;
  }
  const int mt0 = mtile * G_MT, nt0 = ntile * G_NT;
  const int ks_beg = slice * ks_per, iters = ks_per / G_KS;

  // Each wave owns G_LOADS fixed chunks (operand, tile32, plane) of every stage; only the k-step
  // advances, by one fragment block row (NP planes x 1 KiB), so the per-stage issue is an add and a
  // direct-to-LDS load per chunk.  (One wave per SIMD: address arithmetic is not hidden.)
  const uint8_t *chunk_src[G_LOADS];
  int chunk_lds[G_LOADS];
#pragma unroll
  for (int jj = 0; jj < G_LOADS; ++jj) {
    const int c = lw + G_LW * jj;
    const bool real = c < G_CHUNKS;
    const bool isA = c < G_MT * G_KS * NP;
    const int cc = real ? (isA ? c : c - G_MT * G_KS * NP) : 0;
    const int pl = cc % NP, kk = (cc / NP) % G_KS, tl = cc / (NP * G_KS);
    chunk_src[jj] = (isA || !real ? Af : Bf) +
                    ((((size_t)((isA || !real ? mt0 : nt0) + tl) * KS + ks_beg + kk) * NP + pl) * 64 + lane) * 16;
    chunk_lds[jj] = real ? c * G_CHUNK : -1;
  }
  // AUX_A / AUX_B: cache policy of the operand loads (0 default, 2 nt)
  auto issue = [&](int it, int buf) {
    static_for<0, G_LOADS>([&](auto jc) {
      constexpr int jj = decltype(jc)::value;
      constexpr bool allA = G_LW * jj + G_LW - 1 < G_MT * G_KS * NP, allB = G_LW * jj >= G_MT * G_KS * NP;
      if constexpr (kLin1Skip & 3) {
        const bool isA = lw + G_LW * jj < G_MT * G_KS * NP;
        if (((kLin1Skip & 1) && isA) || ((kLin1Skip & 2) && !isA)) return;
      }
      const auto src = (const __attribute__((address_space(1))) void *)(chunk_src[jj] + (size_t)it * (G_KS * NP * G_CHUNK));
      const auto dst = (__attribute__((address_space(3))) void *)(lds + (chunk_lds[jj] >= 0 ? buf * G_STAGE + chunk_lds[jj] : G_STAGES * G_STAGE));
      if constexpr (allA) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, AUX_A);
      else if constexpr (allB) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, AUX_B);
      else __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    });
  };

  constexpr int MPW = 1;                                // M-tiles per wave
  const int wm = wave % G_MT, wn = wave / G_MT;
  f32x16 acc[MPW][NPW];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Software pipeline, two levels.
  //  * global -> LDS: G_STAGES - 1 stages are issued ahead (an HBM round trip is longer than a stage of MFMAs); a counted
  //    s_waitcnt retires the oldest needed one and a raw s_barrier (no vmcnt(0) drain, unlike __syncthreads with LDS-DMA pending)
  //    publishes it.  The barrier at the top of k-step `it` publishes stage it + 1: the fragments of a k-step are read
  //    during the MFMAs of the k-step before.
  //  * LDS -> registers (round 3): the in-kernel stamps of tools/ubench/lin1_parts.hip (-DTT_LIN1_STAMP) gave 739 cycles per
  //    k-step with NO global loads at all -- ten 16-byte fragment reads, a wait for all of them, then twelve MFMAs of 32 cycles
  //    (384) -- and 912 with the loads: the matrix pipe idled while the fragments arrived.  Now a wave reads the B fragments of
  //    N-tile j + 2 (of the next k-step for j = 2, 3) and the next A fragments before the three MFMAs of N-tile j; the compiler's
  //    counted lgkmcnt waits then cover reads issued two tiles (192 matrix cycles) earlier.
  // Slot (it + 3) % 4 = (it - 1) % 4 is refilled after the barrier of k-step it: its last fragment reads were waited for by the
  // MFMAs of k-step it - 1, before that barrier.
  static_assert(G_KS == 1 && MPW == 1 && NPW == 4, "the fragment pipeline below is written for one k-step per stage and four N-tiles per wave");
  auto wait_landed = [&](int younger) {                // all but the `younger` most recently issued stages have landed
    if (younger >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * G_LOADS) : "memory");
    else if (younger == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * G_LOADS) : "memory");
    else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * G_LOADS) : "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G_LOADS) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G_LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(G_STAGES >= 4 && G_STAGES <= 8 && 5 * G_LOADS < 64, "G_STAGES - 1 stages issued ahead; the barrier of k-step it frees slot (it - 1) % G_STAGES");
#ifdef TT_LIN1_STAMP
  const unsigned long long st_c = __builtin_amdgcn_s_memtime(), st_r = __builtin_amdgcn_s_memrealtime();
#endif
  if (loader) {
    for (int p = 0; p < G_STAGES - 1 && p < iters; ++p) issue(p, p);
    wait_landed(min(G_STAGES - 2, iters - 1));         // stage 0
  }
  __builtin_amdgcn_s_barrier();
  // The consumer side is written with inline-assembly reads and hand-counted waits (the scheme of stem.hip): left to the
  // compiler the reads sink next to their use behind s_waitcnt lgkmcnt(0).  LDS operations retire in order, so "all but the N
  // youngest" is exact.  Queue on entering a k-step: B tile 0 (2 reads), A (2), B tile 1 (2) of this k-step, issued during the last.
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 af[2][NP], bq[NPW][NP];      // A fragments of even / odd k-steps (two register sets: a copy would read registers whose load is in flight)
#define TT_DSRD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(off))
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + (uint32_t)lane * 16u;
  const uint32_t offA = (uint32_t)(wm * NP) * G_CHUNK, offB = (uint32_t)(G_MT * NP + wn * NPW * NP) * G_CHUNK;
  auto tile = [&](int j, const u32x4 (&a_cur)[NP]) {
    if constexpr (kLin1Skip & 4) {
      acc[0][j][0] += __uint_as_float(a_cur[0][0] ^ bq[j][0][0] ^ bq[j][1][0] ^ a_cur[1][0]);      // keep the fragment reads alive
    } else {
      // the three products of a tile go to the same accumulator, low terms first (a single chain of this instruction
      // issues back to back: MI355X_MICROARCH.md, cycle constants)
      const f16x8 a0 = __builtin_bit_cast(f16x8, a_cur[0]), a1 = __builtin_bit_cast(f16x8, a_cur[1]);
      const f16x8 b0 = __builtin_bit_cast(f16x8, bq[j][0]), b1 = __builtin_bit_cast(f16x8, bq[j][1]);
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[0][j], 0, 0, 0);
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][j], 0, 0, 0);
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // Every k-step reads ahead, the last one too (it reads a slot that holds older data and nobody uses the values): one
  // code path, so that the compiler never has to reconcile two register assignments with a copy of a register whose load is
  // still in flight -- which is what the peeled last k-step of the first version of this loop did.
  auto kstep = [&](auto par_c, int it) {
    constexpr int P = decltype(par_c)::value;          // it % 2
    u32x4 (&a_cur)[NP] = af[P];
    u32x4 (&a_nxt)[NP] = af[1 - P];
    const uint32_t sb = lds0 + (uint32_t)(it % G_STAGES) * G_STAGE + offB;
    const uint32_t sn = lds0 + (uint32_t)((it + 1) % G_STAGES) * G_STAGE;
    const uint32_t snb = sn + offB, sna = sn + offA;
    TT_DSRD(bq[2][0], sb, 4 * G_CHUNK);
    TT_DSRD(bq[2][1], sb, 5 * G_CHUNK);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(a_cur[0]), "+v"(a_cur[1]));      // in flight: B tiles 1, 2
    tile(0, a_cur);
    TT_DSRD(bq[3][0], sb, 6 * G_CHUNK);
    TT_DSRD(bq[3][1], sb, 7 * G_CHUNK);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bq[1][0]), "+v"(bq[1][1]));      // B tiles 2, 3
    tile(1, a_cur);
    TT_DSRD(bq[0][0], snb, 0);
    TT_DSRD(bq[0][1], snb, G_CHUNK);
    TT_DSRD(a_nxt[0], sna, 0);
    TT_DSRD(a_nxt[1], sna, G_CHUNK);
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bq[2][0]), "+v"(bq[2][1]));      // B tile 3; next k-step: B tile 0, A
    tile(2, a_cur);
    TT_DSRD(bq[1][0], snb, 2 * G_CHUNK);
    TT_DSRD(bq[1][1], snb, 3 * G_CHUNK);
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bq[3][0]), "+v"(bq[3][1]));      // next k-step: B tile 0, A, B tile 1
    tile(3, a_cur);
  };
  if (consumer) {
    const uint32_t s0 = lds0;
    TT_DSRD(bq[0][0], s0 + offB, 0);
    TT_DSRD(bq[0][1], s0 + offB, G_CHUNK);
    TT_DSRD(af[0][0], s0 + offA, 0);
    TT_DSRD(af[0][1], s0 + offA, G_CHUNK);
    TT_DSRD(bq[1][0], s0 + offB, 2 * G_CHUNK);
    TT_DSRD(bq[1][1], s0 + offB, 3 * G_CHUNK);
  }
  auto step = [&](auto par_c, int it) {
    if (loader) wait_landed(max(0, min(it + G_STAGES - 2, iters - 1) - (it + 1)));      // stage it + 1 (the later ones may still be in flight)
    __builtin_amdgcn_s_barrier();                      // stage it + 1 landed for every wave; slot (it - 1) % 4 is free
    if (loader && it + G_STAGES - 1 < iters) issue(it + G_STAGES - 1, (it + G_STAGES - 1) % G_STAGES);
    if (consumer) kstep(par_c, it);
  };
#pragma unroll 1
  for (int it = 0; it < iters; it += 2) {
    step(std::integral_constant<int, 0>{}, it);
    if (it + 1 < iters) step(std::integral_constant<int, 1>{}, it + 1);
  }
  // the reads the last k-step issued ahead: wait for them with their registers still allocated
  if (consumer)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(af[0][0]), "+v"(af[0][1]),
                 "+v"(af[1][0]), "+v"(af[1][1]));
#undef TT_DSRD
  // The partial tile goes out as the accumulators stand -- slab layout [slice][tile32 row][tile32 col][reg / 4][lane][4]
  // -- so that a wave's store is one linear 1 KiB block of 16 bytes per lane (row-major slabs took sixteen
  // 4-byte stores per tile and lane: 8 of the kernel's 45 us at B = 256 were store issue).  Rows / columns of the
  // padding (M to 256, N to 128) are written too; head_mid_kernel reads the same order and drops them.
  // C/D layout: col = lane&31 (N), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (M)
#ifdef TT_LIN1_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 512) {
    g_lin1_stamps[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - st_c;
    g_lin1_stamps[blockIdx.x][1] = 10ull * (__builtin_amdgcn_s_memrealtime() - st_r);
  }
#endif
  if (!consumer) return;
  float4 *dst = (float4 *)part + (size_t)slice * (m_tiles * G_MT) * (n_tiles * G_NT) * 256;
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      float4 *tile = dst + ((size_t)(mt0 + wm + i) * (n_tiles * G_NT) + nt0 + wn * NPW + j) * 256 + lane;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if ((kLin1Skip & 8) && g) continue;
        tile[g * 64] = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
    }
}

// float32 row-major [R][ld] x prescale -> fragment-ordered fp16 planes [tiles][K/16][NP][64][8]
// (rows >= R and k >= kvalid: 0; K is the padded multiple of 16)
__global__ void split_to_frag_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, int R, int K, int tiles,
                                     float prescale, int ld, int kvalid) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one (tile, kstep, lane, j)
  const int KS = K / 16;
  if (i >= (size_t)tiles * KS * 64 * 8) return;
  const int j = i % 8, ln = (i / 8) % 64;
  const size_t tk = i / 512;
  const int ks = tk % KS, tl = tk / KS;
  const int row = tl * 32 + (ln & 31), k = ks * 16 + 8 * (ln >> 5) + j;
  const float v = row < R && k < kvalid ? src[(size_t)row * ld + k] * prescale : 0.f;
  uint16_t h1, h2;
  split_f16x2(v, h1, h2);
  const size_t base = ((size_t)tl * KS + ks) * NP;
  dst[((base + 0) * 64 + ln) * 8 + j] = h1;
  dst[((base + 1) * 64 + ln) * 8 + j] = h2;
}

// fragment-ordered planes of the features -> float32 [n][(16g+k)*PP + pp] (reference Flatten order)
__global__ void frag_to_ref_kernel(const uint16_t *__restrict__ af, float *__restrict__ out, int n, int G, int PP) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)G * PP * 16;
  if (t >= (size_t)n * per) return;
  const int kk = t % 16, pp = (t / 16) % PP, g = (t / (16 * (size_t)PP)) % G;
  const int img = t / per;
  out[(size_t)img * per + ((size_t)(16 * g + kk)) * PP + pp] = load_feature(af, img, G * PP, g * PP + pp, kk);
}

// sum of lin1's K-slices (fixed order) -> BatchNorm1d -> polynomial -> lin2's A operand
// (fragment order, KS2 = ceil(N/16) k-steps; the k padding stays zero from allocation)
// part: lin1's slabs in accumulator order [slice][mt32][nt32][reg / 4][lane][4] (gemm_f16x2_kernel's epilogue); a
// thread owns one (tile, reg / 4, lane) = four rows of one column and reads 16 bytes per slice
__global__ void head_mid_kernel(const float4 *__restrict__ part, int splits, const float *__restrict__ scale,
                                const float *__restrict__ shift, uint16_t *__restrict__ mid_frag, int M, int N,
                                int polynomial, uint32_t *range_flag, int mt32, int nt32) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per_slice = (size_t)mt32 * nt32 * 256;
  if (i >= per_slice) return;
  const int lane = i & 63, g = (i >> 6) & 3;
  const int nt = (int)((i >> 8) % nt32), mt = (int)((i >> 8) / nt32);
  const int col = nt * 32 + (lane & 31), row0 = mt * 32 + 8 * g + 4 * (lane >> 5);
  if (col >= N || row0 >= M) return;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
  for (int z = 0; z < splits; ++z) {     // fixed order: reproducible
    const float4 v = part[(size_t)z * per_slice + i];
    s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
  }
  const float sc = scale[col], sh = shift[col];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (row0 + e >= M) break;
    float y = fmaf((float)s[e], sc, sh);
    if (polynomial) {                    // vAlexnet's Classifier_scale has no activation (:671-675)
      // 0.47 + 0.50 * x + 0.09 * x ** 2, evaluated left to right in fp32 like the reference
      const float t = __fadd_rn(0.47f, __fmul_rn(0.50f, y));
      y = __fadd_rn(t, __fmul_rn(0.09f, __fmul_rn(y, y)));
    }
    store_feature(mid_frag, row0 + e, (N + 15) / 16, col >> 4, col & 15, y, range_flag);
  }
}

// ---- lin2: out[M][N] = A[M][K] * B[N][K]^T * inv + bias, split operands in fragment order ----
// A: [ceil(M/32)][KS][NP][64][8], B: [ceil(N/32)][KS][NP][64][8].  One 32x32 output tile per
// workgroup (256 workgroups at M = 256, N = 1000: the kernel is bound by how fast one CU can pull
// its 2 x 126 KiB of fragments out of L2, so the tiles are small and spread over all CUs), K
// split over the 8 waves, partial tiles summed through LDS in wave order, bias fused.
constexpr int L2_WAVES = 8, L2_STAGES = 4;
__global__ __launch_bounds__(64 * L2_WAVES) void lin2_f16x2_kernel(const uint4 *__restrict__ A, const uint4 *__restrict__ B,
                                                                   const float *__restrict__ bias, float inv,
                                                                   float *__restrict__ out, int M, int N, int KS) {
  __shared__ __align__(16) float red[L2_WAVES * 16 * 64];            // [wave][reg][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mt = blockIdx.y, nt = blockIdx.x;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nk = wave < KS ? (KS - wave + L2_WAVES - 1) / L2_WAVES : 0;      // this wave's k-steps: wave + 8 i
  uint4 fa[L2_STAGES][NP], fb[L2_STAGES][NP];
  auto load = [&](int i, uint4 (&a)[NP], uint4 (&b)[NP]) {
    const int ks = wave + L2_WAVES * i;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      a[pl] = A[(((size_t)mt * KS + ks) * NP + pl) * 64 + lane];
      b[pl] = B[(((size_t)nt * KS + ks) * NP + pl) * 64 + lane];
    }
  };
  static_for<0, L2_STAGES>([&](auto ss) {
    constexpr int sl = decltype(ss)::value;
    if (sl < nk) load(sl, fa[sl], fb[sl]);
  });
  for (int i0 = 0; i0 < nk; i0 += L2_STAGES) {
    static_for<0, L2_STAGES>([&](auto ss) {
      constexpr int sl = decltype(ss)::value;
      if (i0 + sl < nk) {
        const f16x8 a0 = __builtin_bit_cast(f16x8, fa[sl][0]), a1 = __builtin_bit_cast(f16x8, fa[sl][1]);
        const f16x8 b0 = __builtin_bit_cast(f16x8, fb[sl][0]), b1 = __builtin_bit_cast(f16x8, fb[sl][1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
        if (i0 + sl + L2_STAGES < nk) load(i0 + sl + L2_STAGES, fa[sl], fb[sl]);
      }
    });
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * 64; e += 64 * L2_WAVES) {
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < L2_WAVES; ++w) sum += (double)red[w * (16 * 64) + e];      // wave order: reproducible
    const int ln = e & 63, r = e >> 6;
    const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    const int col = nt * 32 + (ln & 31);
    if (row < M && col < N) out[(size_t)row * N + col] = (float)(sum * (double)inv + (double)bias[col]);
  }
}

__global__ void permute_lin1_kernel(const float *__restrict__ w1, float *__restrict__ w1p, int O, int G, int PP) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t K = (size_t)G * PP * 16;
  if (i >= (size_t)O * K) return;
  const size_t o = i / K, f = i % K;
  const int k = f % 16, pp = (f / 16) % PP, g = f / (16 * (size_t)PP);
  w1p[i] = w1[o * K + ((size_t)(16 * g + k)) * PP + pp];
}

}  // namespace

int gemm_f16x2_splits(int M, int N, int KS) {
  // the most K-slices (a divisor of the k-step count, slices of at least 8 k-steps) that still give at most one
  // workgroup per CU -- but never fewer than 8 where the shape allows it: a slice is one float32 accumulation chain per
  // output, and chains longer than 2048 terms cost accuracy (batch 600: 4 slices of 4096 put the logits 1.03e-5 from
  // the 16-slice result of batch 256); the slices themselves are summed in float64 by head_mid_kernel
  const int tiles = ((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
  const int want = std::max(TT_LIN1_WGS / std::max(1, tiles), 8);
  for (int s = std::min(std::min(want, KS / 8), 128); s > 1; --s)   // (<= 128: the reduce pass reads every slab)
    if (KS % s == 0) return s;
  return 1;
}

size_t gemm_f16x2_part_elems(int M, int N, int KS) {
  const size_t mt = (M + G_BM - 1) / G_BM, nt = (N + G_BN - 1) / G_BN;
  return (size_t)gemm_f16x2_splits(M, N, KS) * mt * G_BM * nt * G_BN;
}

int launch_gemm_f16x2(const void *Af, const void *Bf, float *part, int M, int N, int K, int splits, hipStream_t s) {
  const int KS = K / 16;
  if (K % 16 || KS % splits || (KS / splits) % G_KS) {
    set_error("gemm_f16x2: K=%d not divisible into %d slices of %d k-steps", K, splits, G_KS);
    return TTNET_E_UNSUPPORTED;
  }
  // TTNET_LIN1_NT=<a><b> (diagnostic): non-temporal loads of the A / B operand stream.  Measured (profiles/r03_cache_policy.txt): nt weights
  // cost the forward with two batches in flight 8 % -- the 66 MB of weights are what the Infinity Cache should keep
  static const int nt = getenv("TTNET_LIN1_NT") ? atoi(getenv("TTNET_LIN1_NT")) : 0;
  const int n_tiles = (N + G_BN - 1) / G_BN, m_tiles = (M + G_BM - 1) / G_BM;
  auto launch = [&](auto kernel) -> int {
    TT_TRY(ensure_dynamic_lds((const void *)kernel, G_LDS));
    hipLaunchKernelGGL(kernel, dim3(n_tiles * m_tiles * splits), dim3(G_THREADS), G_LDS, s, (const uint8_t *)Af, (const uint8_t *)Bf, part, M, N,
                       KS, KS / splits, n_tiles, m_tiles, splits);
    return TTNET_OK;
  };
  if (nt == 11) TT_TRY(launch(gemm_f16x2_kernel<2, 2>));
  else if (nt == 1) TT_TRY(launch(gemm_f16x2_kernel<0, 2>));
  else if (nt == 10) TT_TRY(launch(gemm_f16x2_kernel<2, 0>));
  else TT_TRY(launch(gemm_f16x2_kernel<0, 0>));
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

size_t frag_elems(int rows, int K) { return (size_t)((rows + 31) / 32) * (K / 16) * NP * 64 * 8; }

float weight_prescale(const float *w, size_t n) {
  float amax = 0.f;
  for (size_t i = 0; i < n; ++i) amax = fmaxf(amax, fabsf(w[i]));
  if (!(amax > 0.f) || !std::isfinite(amax)) return 1.0f;
  int e;
  frexpf(amax, &e);                      // amax = f * 2^e, f in [0.5, 1)
  int k = 14 - e;                        // amax * 2^k in [8192, 16384)
  k = k > 60 ? 60 : (k < -60 ? -60 : k);
  return ldexpf(1.0f, k);
}

int launch_split_to_frag(const float *src, void *dst, int R, int K, int rows_padded, float prescale, hipStream_t s, int ld,
                         int kvalid) {
  if (K % 16) {
    set_error("split_to_frag: K=%d is not a multiple of 16", K);
    return TTNET_E_UNSUPPORTED;
  }
  const int tiles = (rows_padded + 31) / 32;
  const size_t t = (size_t)tiles * (K / 16) * 64 * 8;
  hipLaunchKernelGGL(split_to_frag_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, src, (uint16_t *)dst, R, K, tiles,
                     prescale, ld > 0 ? ld : K, kvalid > 0 ? kvalid : K);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_frag_to_reference_order(const void *af, float *out, int n, int G, int PP, hipStream_t s) {
  const size_t t = (size_t)n * G * PP * 16;
  hipLaunchKernelGGL(frag_to_ref_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, (const uint16_t *)af, out, n, G, PP);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_head_mid(const float *part, int splits, const float *scale, const float *shift, void *mid_frag, int M, int N,
                    int polynomial, uint32_t *range_flag, hipStream_t s) {
  const int mt32 = (M + G_BM - 1) / G_BM * G_MT, nt32 = (N + G_BN - 1) / G_BN * G_NT;       // the slabs' padded tile grid
  const size_t t = (size_t)mt32 * nt32 * 256;
  hipLaunchKernelGGL(head_mid_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, (const float4 *)part, splits, scale, shift,
                     (uint16_t *)mid_frag, M, N, polynomial, range_flag, mt32, nt32);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_lin2_f16x2(const void *mid_frag, const void *w2f, const float *bias, float inv, float *out, int M, int N, int K,
                      hipStream_t s) {
  const int KS = (K + 15) / 16;
  hipLaunchKernelGGL(lin2_f16x2_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(64 * L2_WAVES), 0, s,
                     (const uint4 *)mid_frag, (const uint4 *)w2f, bias, inv, out, M, N, KS);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_permute_lin1(const float *w1, float *w1p, int O, int G, int PP, hipStream_t s) {
  const size_t t = (size_t)O * G * PP * 16;
  hipLaunchKernelGGL(permute_lin1_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, w1, w1p, O, G, PP);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
