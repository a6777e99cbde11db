// Internal declarations shared by the HIP translation units of libttnet.so.
// gfx950 (MI355X) only: 64-wide wavefronts, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/ttnet.h"

namespace ttnet {

void set_error(const char *fmt, ...);

#define TT_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t e__ = (call);                                                               \
    if (e__ != hipSuccess) {                                                               \
      ::ttnet::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                         __LINE__);                                                        \
      return TTNET_E_HIP;                                                                  \
    }                                                                                      \
  } while (0)

#define TT_TRY(expr)          \
  do {                        \
    int s__ = (expr);         \
    if (s__ != TTNET_OK) return s__; \
  } while (0)

constexpr int kWave = 64;
constexpr int kMaxLds = 160 * 1024;

// Geometry of one Block_TT (models/TT_FHE_SMALL.py:281-305).
struct BlockGeom {
  std::string name;
  int in_planes = 0, out_planes = 0, kh = 1, kw = 1, stride = 1, pad = 0, groups = 1;
  bool last = false;
  int cin_g() const { return in_planes / groups; }
  int mid_g() const { return 8 * in_planes / groups; }
  int cout_g() const { return out_planes / groups; }
  int nbits() const { return cin_g() * kh * kw; }
  // internal table entry width in bits: 1 (bit-packed along the index), 8 or 16
  int entry_bits() const { return last ? 32 * cout_g() : (cout_g() == 1 ? 1 : (cout_g() <= 8 ? 8 : 16)); }
  size_t table_bytes() const {
    size_t entries = (size_t)groups << nbits();
    if (last) return entries * cout_g() * sizeof(float);
    if (entry_bits() == 1) {                       // one dword minimum per group (16-entry tables)
      const size_t per = ((size_t)1 << nbits()) / 8;
      return (size_t)groups * (per < 4 ? 4 : per);
    }
    return entries * entry_bits() / 8;
  }
};

#if defined(__HIPCC__)
// Lane LANE of (lo,hi) := a wave-uniform 64-bit value (v_writelane_b32; hipcc exposes no
// builtin).  gfx950 needs 2 wait states between a VALU write of an SGPR (the v_cmp of a
// ballot) and a VALU read of it; hipcc pads that for its own instructions but not inside
// an asm statement, hence the s_nop.
template <int LANE>
__device__ inline void writelane64(uint32_t &lo, uint32_t &hi, uint64_t sval) {
  asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
      : "+v"(lo), "+v"(hi)
      : "s"((uint32_t)sval), "s"((uint32_t)(sval >> 32)), "n"(LANE));
}

// compile-time loop: f(integral_constant<int, I>) for I in [0, N)
template <int I, int N, typename F>
__device__ inline void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- 16x16 bit-matrix transpose across a 16-lane group (gate.hip, stem.hip) -----------------
// Four butterfly stages (lane distance 8, 4, 2, 1): a lane exchanges words with its partner
// (ds_swizzle), funnel-rotates the partner's word and keeps its own bits at the positions
// selected by the stage mask.  Both 16-bit halves of the register are transposed at once.
struct DwLaneConst {
  uint32_t c4;          // byte offset of this lane's channel inside a striped table row
  uint32_t rot[4];      // funnel-rotate amount of butterfly stage s = 8,4,2,1
  uint32_t keep[4];     // bits this lane keeps in stage s
};

__device__ inline DwLaneConst dw_lane_const(uint32_t lane) {
  DwLaneConst k;
  k.c4 = (lane & 15) << 2;
  constexpr uint32_t M[4] = {0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};   // bit positions with (pos & s) == 0
  constexpr uint32_t S[4] = {8, 4, 2, 1};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool low = (lane & S[i]) == 0;
    k.rot[i] = low ? 32 - S[i] : S[i];          // low lane takes partner << s, high lane partner >> s
    k.keep[i] = low ? M[i] : ~M[i];
  }
  return k;
}

// 16x16 bit transpose across a 16-lane group, both 16-bit halves at once
__device__ inline uint32_t transpose16(uint32_t acc, const DwLaneConst &k) {
  constexpr int S[4] = {8, 4, 2, 1};
  static_for<0, 4>([&](auto i) {
    constexpr int I = decltype(i)::value;
    const uint32_t partner = (uint32_t)__builtin_amdgcn_ds_swizzle((int)acc, 0x1F | (S[I] << 10));
    const uint32_t moved = __builtin_amdgcn_alignbit(partner, partner, k.rot[I]);
    acc = moved ^ ((moved ^ acc) & k.keep[I]);            // keep ? acc : moved, one v_bitop3_b32
  });
  return acc;
}

// The value of lane (l ^ S) within each row of 16 lanes, by DPP (a VALU move: no trip through the LDS
// crossbar, whose issue rate -- shared by the four SIMDs of a CU -- bounds kernels that transpose a lot)
template <int S>
__device__ inline uint32_t lane_xor16(uint32_t v) {
  static_assert(S == 8 || S == 4 || S == 2 || S == 1, "butterfly distances");
  if constexpr (S == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);           // row_ror:8
  else if constexpr (S == 4)                                                                              // l ^ 7, then l ^ 3
    return (uint32_t)__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
  else if constexpr (S == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
  else return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);                            // quad_perm [1,0,3,2]
}

// N independent 16x16 bit transposes (both 16-bit halves of each register), stage by stage, exchanges by DPP
template <int N>
__device__ inline void transpose16_multi(uint32_t (&acc)[N], const DwLaneConst &k) {
  constexpr int S[4] = {8, 4, 2, 1};
  static_for<0, 4>([&](auto i) {
    constexpr int I = decltype(i)::value;
    uint32_t partner[N];
#pragma unroll
    for (int n = 0; n < N; ++n) partner[n] = lane_xor16<S[I]>(acc[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const uint32_t moved = __builtin_amdgcn_alignbit(partner[n], partner[n], k.rot[I]);
      acc[n] = moved ^ ((moved ^ acc[n]) & k.keep[I]);
    }
  });
}

#endif

// Raise a kernel's dynamic LDS limit (> 64 KiB) once per kernel and process: the attribute call
// costs host microseconds that a launch-bound forward cannot afford on every launch (plan.hip).
int ensure_dynamic_lds(const void *kernel, size_t bytes);

// ---- launchers (defined next to their kernels) ------------------------------------------

// lut_build.hip
struct LutBuildArgs {
  const float *w1;      // conv1.weight [G*mid_g][n] (canonical column order c,kh,kw)
  const float *w2;      // conv2.weight [G*cout_g][mid_g]
  const double *s1, *t1;  // folded bn1 [G*mid_g]
  const double *s2, *t2;  // folded bn2 [G*cout_g]
  const uint8_t *perm;  // [n] internal index bit p -> canonical input column
  int groups, n, mid_g, cout_g, last;
  void *table;          // internal layout (see BlockGeom::entry_bits)
  unsigned *near_ties;  // one counter
};
int launch_lut_build(const LutBuildArgs &a, hipStream_t s);

// ---- split-operand format shared by the stem and lin1 (arithmetic: see stem.hip) -----------
// A float32 value, prescaled by a power of two, is carried as two fp16 terms h1 + h2
// (22 significant bits).  Activations use the fixed prescale ACT_PRESCALE; each weight tensor
// uses its own (weight_prescale); the product of the two is divided out of the BatchNorm scale
// that follows.  Fragment order of an operand: [tile32][kstep16][plane][lane][8 fp16], lane l
// holding row 32*tile + (l & 31), k = 16*kstep + 8*(l >> 5) + j.
constexpr int SPLIT_PLANES = 2;
constexpr float ACT_PRESCALE = 16.0f;      // |activation| < 4094

__device__ inline void split_f16x2(float v, uint16_t &h1, uint16_t &h2) {
  const _Float16 a = (_Float16)v;
  const _Float16 b = (_Float16)(v - (float)a);
  h1 = __builtin_bit_cast(uint16_t, a);
  h2 = __builtin_bit_cast(uint16_t, b);
}
__device__ inline float f16_bits_to_float(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// Range of the split: the prescaled value must stay below the fp16 maximum (|activation| < 4094 at
// ACT_PRESCALE 16).  The reference's float32 path has no such limit, so a value outside it must not
// pass silently: kernels that split activations raise the plan's sticky flag (a word of host-mapped
// memory, written only in the failing case), and the next C-ABI call on the plan returns
// TTNET_E_RANGE (plan.hip).  NaN counts as out of range.
__device__ inline bool split_out_of_range(float prescaled) { return !(__builtin_fabsf(prescaled) < 65504.0f); }

// feature (image img, k-step ks of KS, element kk of 16) into lin1's A operand
__device__ inline void store_feature(uint16_t *feat_frag, int img, int KS, int ks, int kk, float f, uint32_t *range_flag) {
  uint16_t h1, h2;
  const float fs = f * ACT_PRESCALE;
  if (split_out_of_range(fs)) *range_flag = 1u;
  split_f16x2(fs, h1, h2);
  const int ln = (img & 31) + 32 * (kk >> 3), j = kk & 7;
  const size_t base = ((size_t)(img >> 5) * KS + ks) * SPLIT_PLANES;
  feat_frag[((base + 0) * 64 + ln) * 8 + j] = h1;
  feat_frag[((base + 1) * 64 + ln) * 8 + j] = h2;
}
__device__ inline float load_feature(const uint16_t *feat_frag, int img, int KS, int ks, int kk) {
  const int ln = (img & 31) + 32 * (kk >> 3), j = kk & 7;
  const size_t base = ((size_t)(img >> 5) * KS + ks) * SPLIT_PLANES;
  return (f16_bits_to_float(feat_frag[((base + 1) * 64 + ln) * 8 + j]) + f16_bits_to_float(feat_frag[((base + 0) * 64 + ln) * 8 + j])) *
         (1.0f / ACT_PRESCALE);
}
// host: the power of two that brings max|w| into [8192, 16384) (1 for an all-zero / non-finite tensor)
float weight_prescale(const float *w, size_t n);

// stem.hip
// wfrag: the conv weights (BN scale folded in) and, as one more k-row, the folded BN shift, split into two
// fp16 planes in MFMA fragment order; init[64]: [0] = the constant that row multiplies (both from
// stem_split_weights, which fails if the shift is out of range)
// x: float32 NCHW, or (x_is_u8) uint8 NHWC with wfrag / init / norm_tab from stem_split_weights_u8
int launch_stem(const void *x, bool x_is_u8, const uint32_t *norm_tab, const void *wfrag, const float *init, uint64_t *rp,
                uint16_t *cp, int n, int p, uint32_t *range_flag, hipStream_t s, int workgroups = 0);      // 0: one per CU
// uint8 input: weights with the normalisation folded in, centres and border corrections (stem.hip)
size_t stem_u8_table_elems();
bool stem_split_weights_u8(const float *w, const double *scale, const double *shift, int p, const float mean[3], const float stdv[3],
                           uint16_t *out, float *init, uint32_t *tab /*[stem_u8_table_elems()]*/);
bool stem_split_weights(const float *w /*[p][3][7][7]*/, const double *scale, const double *shift, int p, uint16_t *out,
                        float *init /*[64]*/);
size_t stem_split_weights_elems();
// byte sizes of stem_pc_kernel's arguments, in order (plan.hip re-points the input of a captured graph node and
// must own a copy of every argument); returns their number
int stem_kernel_arg_sizes(const int **sizes);

// gate.hip
struct GateBlockArgs {
  int n;                 // images
  int C, H, W;           // input planes and size
  int Ho, Wo;            // branch output size after padding
  int off34;             // left/top zero padding of out3/out4 (1 at W=56, else 0)
  int kh1, kw1, kh2, kw2, stride, pad;
  int cf_bits;           // output bits per Block_convf group (8; 4 when convf keeps the channel count)
  const uint64_t *x_rp;  // [n][C][H]
  const uint16_t *x_cp;  // [n][C/16][H][W]
  const uint8_t *t_dw1, *t_dw2;   // [C/16][2^n/32][16] dwords: 16 channels striped per dword
  const uint16_t *t_c3;           // [C/16][65536]
  uint16_t *o1, *o2, *o3, *o4;    // [n][C/16][Ho][Wo]
};
// stage 1: Block_conv1, Block_conv2 (depthwise units) and Block_conv3 + both majority pools
int launch_gate_stage1(const GateBlockArgs &a, hipStream_t s);
// stage 2: convf of a non-last block: 4 branch tensors -> words [n][Cout/16][Ho][Wo] and rows
// [n][Cout][Ho], Cout = cf_bits * (4C/16)
int launch_gate_pf(const GateBlockArgs &a, const uint8_t *t_cf, uint16_t *out_cp, uint64_t *out_rp, hipStream_t s);
// convf of the last block through the float table, AvgPool2d(2) fused; the features are
// written pre-split for lin1: fragment-ordered fp16 planes [n/32][(g*PP+pp)][2][64][8]
// (idx != nullptr: the branch dwords of gate_fused.hip instead of the four word tensors o1..o4)
int launch_gate_last(const GateBlockArgs &a, const float *t_last, void *feat_frag, uint32_t *range_flag, hipStream_t s,
                     const uint32_t *idx = nullptr);
// gate_fused.hip: one launch per stride-2 block of TT-small, branch tensors kept on chip.
// Activations: one layout, rows [n][C][H] of uint64 (W > 32) / uint32 (W > 16) / uint16 words.
struct FusedBlockArgs {
  int n, C, H, Ho, off34, last;
  const void *x;             // block input rows
  const void *img_c3, *img_dw;   // table images built by launch_fused_images
  const uint8_t *t_cf;       // binarised Block_convf table [C/4][65536] (not read by a last block)
  void *y;                   // output rows [n][2C][Ho] (binarised blocks)
  uint32_t *idx;             // branch dwords [n][C/8][Ho*Ho]: output of a last block (read by gate_last), optional tap otherwise
};
bool fused_block_supported(int C, int H, int Ho, int stride, int pad, int kh, int kw);
int row_bytes(int W);
int launch_gate_block(const FusedBlockArgs &f, hipStream_t s);
int launch_fused_images(const void *t_dw1, const void *t_dw2, const void *t_c3, int C, void *img_dw, void *img_c3, hipStream_t s);
int launch_branch_rows(const uint32_t *idx, uint64_t *rows, int n, int C, int Ho, int branch, hipStream_t s);
int launch_widen_rows(const void *src, uint64_t *dst, size_t count, int W, hipStream_t s);
int launch_narrow_rows(const uint64_t *src, void *dst, size_t count, int W, hipStream_t s);
// x-small variant (fan-in 4, gate_xs.hip): everything on row-packed planes
int launch_xs_branches(const GateBlockArgs &a, const void *t_c3, uint64_t *const o[4], hipStream_t s);
int launch_xs_pf(int n, int C, int Ho, int Wo, int cout_g, uint64_t *const o[4], const void *t_cf, uint64_t *out_rp,
                 hipStream_t s);
int launch_xs_last(int n, int C, int Ho, int Wo, int cout_g, uint64_t *const o[4], const float *t_last, void *feat_frag,
                   uint32_t *range_flag, hipStream_t s);
// full variant (fan-in 30, gate_full.hip): direct float64 evaluation on row-packed planes
struct FullDwArgs {
  int n, C, H, W;           // input planes
  int kh, kw, stride, pad;
  int ho, wo;               // conv output size
  int Ho;                   // rows per plane of `out` (after the branch padding)
  int pad_t, pad_l;         // where the conv output sits inside the padded plane
  const uint64_t *x_rp;
  const float *w1, *w2;     // conv1.weight [8C][kh*kw], conv2.weight [C][8]
  const double *s1, *t1, *s2, *t2;
  uint64_t *out;            // [n][C][Ho]
  // fast evaluation (gate_full.hip: full_dw_fast_kernel): the outputs whose sign float32 cannot vouch for
  uint32_t *fix_list, *fix_count;   // ids of up to fix_cap outputs; one counter
  uint32_t fix_cap;
  float tau_scale;          // as FullPwArgs
  const float *gel;         // launch_full_gelu_tables' table for scale 1 (nullptr: the kernel builds its own copy)
};
struct FullPwArgs {
  int n, H, W;              // pixel grid
  int groups, cin, mid, cout, Cout;
  int Csrc;                 // channels per source tensor
  int interleaved;          // 0: src[0] planes; 1: channel J -> src[J%4] plane J/4 (the 4-branch concat)
  const uint64_t *src[4];
  const float *w1, *w2;     // conv1.weight [G*mid][cin], conv2.weight [G*cout][mid]
  const double *s1, *t1, *s2, *t2;
  uint64_t *out_rp;         // [n][Cout][H]  (binarised blocks)
  float *out_float;         // [n][Cout][H][W] relu'd (last block), or nullptr
  // fast evaluation (gate_full.hip: full_pw_fast_kernel): the (pixel, group) pairs whose sign the split-fp16
  // evaluation cannot vouch for, [groups][n*H*W] pixel ids and [64] counters; the range flag of the plan
  uint32_t *fix_list, *fix_count, *range_flag;
  float tau_scale;          // 1; TTNET_FULL_TAU_SCALE shrinks the bound to measure its margin (tests only)
  const float *gel;         // launch_full_gelu_tables' table for the activation prescale (nullptr: built by the kernel)
};
int launch_full_dw(const FullDwArgs &a, hipStream_t s);
int launch_full_pw(const FullPwArgs &a, hipStream_t s);
// the fast kernels' GELU tables (tangent lines at 4096 nodes): [2][4096][2] float32, scale 1 (depthwise) and the activation
// prescale (1x1); model-independent, built once per plan
size_t full_gelu_tables_elems();
int launch_full_gelu_tables(float *dst, hipStream_t s);
int launch_rp_majority(const uint64_t *x, uint64_t *out, int n, int C, int H, int W, int Ho, int pad_t, int pad_l,
                       hipStream_t s);
int launch_full_pool_split(const float *x, void *feat_frag, int n, int C, int H, int W, uint32_t *range_flag, hipStream_t s);
// CIFAR vAlexnet variant (gate_va.hip)
int launch_va_stem(const float *x, const float *w, const float *bias, const float *scale, const float *shift,
                   uint64_t *rp, int n, hipStream_t s);
int launch_va_block(const uint64_t *x_rp, const void *t1, const void *t2, const void *t3, uint64_t *y, int n, hipStream_t s);
int launch_va_feat(const uint64_t *y, void *feat_frag, int n, hipStream_t s);
int launch_va_frag_to_flat(const void *af, float *out, int n, hipStream_t s);
int launch_cp_to_rp(const uint16_t *cp, uint64_t *rp, int n, int C, int H, int W, hipStream_t s);
int launch_rp_to_cp(const uint64_t *rp, uint16_t *cp, int n, int C, int H, int W, hipStream_t s);
// feature planes (fragment order) -> float32 [n][(16g+k)*PP + pp] (reference Flatten order)
int launch_frag_to_reference_order(const void *af, float *out, int n, int G, int PP, hipStream_t s);
// fp16 x 2 split GEMM on fragment-ordered operands: part = split-K slabs in accumulator order (head.hip)
int gemm_f16x2_splits(int M, int N, int KS);
size_t gemm_f16x2_part_elems(int M, int N, int KS);      // float32 elements of the split-K slabs (padded tile grid)
int launch_gemm_f16x2(const void *Af, const void *Bf, float *part, int M, int N, int K, int splits, hipStream_t s);
size_t frag_elems(int rows, int K);
// src [R][ld] (ld = 0: K) -> split planes with K (a multiple of 16) columns, columns >= kvalid (0: K) zero
int launch_split_to_frag(const float *src, void *dst, int R, int K, int rows_padded, float prescale, hipStream_t s, int ld = 0,
                         int kvalid = 0);

// head.hip
// z = sum_s part; z = z*scale+shift; y = 0.47+0.5z+0.09z^2 (Classifier_scale middle), written as lin2's
// split A operand: fragment order, rows = images (allocate padded to 64), ceil(N/16) k-steps
int launch_head_mid(const float *part, int splits, const float *scale, const float *shift, void *mid_frag, int M, int N,
                    int polynomial, uint32_t *range_flag, hipStream_t s);
// logits[M][N] = mid[M][K] * W2[N][K]^T * inv + bias on split operands (w2f rows padded to 64)
int launch_lin2_f16x2(const void *mid_frag, const void *w2f, const float *bias, float inv, float *out, int M, int N, int K,
                      hipStream_t s);
// W1p[o][(g*PP+pp)*16+k] = W1[o][(16g+k)*PP+pp]
int launch_permute_lin1(const float *w1, float *w1p, int O, int G, int PP, hipStream_t s);

}  // namespace ttnet
