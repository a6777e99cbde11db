// Gate path: the binarised 4-branch blocks evaluated as truth-table lookups on packed bits.
//
// Replaces Block_resnet_multihead_general_BN_vf_imgnet_v2small.forward
// (models/TT_general_imagenet_v2_small.py:78-148) and the Block_TT.forward calls inside it
// (models/TT_FHE_SMALL.py:307-320).  All work here is integer / bitwise and bit exact.
//
// Data layout in HBM (include/ttnet.h): depthwise windows read row-packed planes
// (uint64 per image row), the grouped 1x1 blocks read channel-packed words (uint16 = the 16
// input channels of one group = the table index itself).  Tables are staged in LDS
// (128 KiB per workgroup) and read with one ds_read per lookup.
//
// Bound: HBM nominally (packed activations + tables once per batch, SURVEY §8d:
// 74,592 B/image + 14.2 MB); in practice LDS gather issue + index-forming VALU.

#include "ttnet_common.h"

namespace ttnet {

namespace {

constexpr int kGateThreads = 512;

__device__ inline void copy_to_lds(uint8_t *dst, const uint8_t *src, size_t bytes) {
  // bytes is a multiple of 4; both sides 4-byte aligned
  if ((bytes & 15) == 0 && (((uintptr_t)src) & 15) == 0) {
    const uint4 *s4 = (const uint4 *)src;
    uint4 *d4 = (uint4 *)dst;
    for (size_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d4[i] = s4[i];
  } else {
    const uint32_t *s1 = (const uint32_t *)src;
    uint32_t *d1 = (uint32_t *)dst;
    for (size_t i = threadIdx.x; i < bytes / 4; i += blockDim.x) d1[i] = s1[i];
  }
}

__device__ inline uint32_t maj4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  // at least two of four set, bitwise (act(AvgPool2d(2)(x) - 0.5), :93-94)
  return (a & b) | (c & d) | ((a | b) & (c | d));
}

// ---- depthwise Block_conv1 / Block_conv2 ---------------------------------------------------
// grid (C/16, 2 branches, slices); lane = (channel c = lane&15, output row slot = lane>>4).
// Each lane walks one output row of its channel; the ballot of 64 lanes is four
// channel-packed words (4 rows x 16 channels) of one output column.
template <int KH, int KW>
__global__ __launch_bounds__(kGateThreads) void gate_dw_kernel(GateBlockArgs a, int tb, int imgs_per_slice) {
  extern __shared__ __align__(16) uint8_t lds[];
  const int q = blockIdx.x, branch = blockIdx.y;
  const int Q = a.C / 16;
  const uint8_t *tab = (branch ? a.t_dw2 : a.t_dw1) + (size_t)q * 16 * tb;
  uint16_t *out = branch ? a.o2 : a.o1;
  copy_to_lds(lds, tab, (size_t)16 * tb);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int c = lane & 15, slot = lane >> 4;
  const int n0 = blockIdx.z * imgs_per_slice;
  const int n1 = min(a.n, n0 + imgs_per_slice);
  const int rows4 = (a.Ho + 3) / 4;
  const int tasks = (n1 - n0) * rows4;
  const uint8_t *mytab = lds + c * tb;
  for (int t = wave; t < tasks; t += nwaves) {
    const int n = n0 + t / rows4, oy = (t % rows4) * 4 + slot;
    const bool valid = oy < a.Ho;
    uint64_t row[KH];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      const int iy = oy * a.stride - a.pad + kh;
      uint64_t r = 0;
      if (valid && iy >= 0 && iy < a.H) r = a.x_rp[((size_t)n * a.C + 16 * q + c) * a.H + iy];
      row[kh] = r << a.pad;
    }
    uint64_t keep = 0;
    for (int ox = 0; ox < a.Wo; ++ox) {
      const int sh = ox * a.stride;
      uint32_t idx = 0;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) idx |= ((uint32_t)(row[kh] >> sh) & ((1u << KW) - 1u)) << (kh * KW);
      const uint32_t byte = mytab[idx >> 3];
      const bool bit = valid && ((byte >> (idx & 7)) & 1u);
      const uint64_t m = __ballot(bit);
      if (lane == ox) keep = m;
    }
    if (lane < a.Wo) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int oys = (t % rows4) * 4 + s;
        if (oys < a.Ho) out[(((size_t)n * a.Ho + oys) * a.Wo + lane) * Q + q] = (uint16_t)(keep >> (16 * s));
      }
    }
  }
}

// ---- Block_conv3 (grouped 1x1, 16 -> 16 bits) + the two 2x2 majority pools ---------------
// grid (C/16, slices); one thread per pooled pixel: four lookups, two majorities.
__global__ __launch_bounds__(kGateThreads) void gate_pw_kernel(GateBlockArgs a, int imgs_per_slice) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint16_t *tab = (uint16_t *)lds;
  const int q = blockIdx.x, Q = a.C / 16;
  copy_to_lds(lds, (const uint8_t *)(a.t_c3 + (size_t)q * 65536), 65536 * 2);
  __syncthreads();
  const int Hp = a.H / 2, Wp = a.W / 2;
  const int n0 = blockIdx.y * imgs_per_slice;
  const int n1 = min(a.n, n0 + imgs_per_slice);
  const int per = Hp * Wp;
  const int tasks = (n1 - n0) * per;
  for (int t = threadIdx.x; t < tasks; t += blockDim.x) {
    const int n = n0 + t / per, r = t % per, py = r / Wp, px = r % Wp;
    const uint16_t *src = a.x_cp + (((size_t)n * a.H + 2 * py) * a.W + 2 * px) * Q + q;
    const uint32_t w0 = src[0], w1 = src[Q], w2 = src[(size_t)a.W * Q], w3 = src[(size_t)a.W * Q + Q];
    const uint32_t r0 = tab[w0], r1 = tab[w1], r2 = tab[w2], r3 = tab[w3];
    const size_t dst = (((size_t)n * a.Ho + py + a.off34) * a.Wo + px + a.off34) * Q + q;
    a.o3[dst] = (uint16_t)maj4(r0, r1, r2, r3);
    a.o4[dst] = (uint16_t)maj4(w0, w1, w2, w3);
  }
}

// ---- Block_convf of a binarised block (grouped 1x1 over the interleaved branches) --------
// The reference interleaves to channel 4c+branch (:144-147) and groups 16 of those: group g
// reads channels 4g..4g+3 of each branch.  Internal index = nib(out1) | nib(out2)<<4 |
// nib(out3)<<8 | nib(out4)<<12, each nibble LSB = channel 4g.  Two groups (2 x 64 KiB of
// 8-bit entries) per workgroup produce one channel-packed output word.
__global__ __launch_bounds__(kGateThreads) void gate_pf_kernel(GateBlockArgs a, const uint8_t *t_cf,
                                                              uint16_t *out_cp, int imgs_per_slice) {
  extern __shared__ __align__(16) uint8_t lds[];
  const int j = blockIdx.x;              // output word; groups 2j, 2j+1
  const int Q = a.C / 16, Qout = a.C / 8;
  copy_to_lds(lds, t_cf + (size_t)(2 * j) * 65536, 2 * 65536);
  __syncthreads();
  const int wq = j >> 1, sh = 8 * (j & 1);
  const int n0 = blockIdx.y * imgs_per_slice;
  const int n1 = min(a.n, n0 + imgs_per_slice);
  const int per = a.Ho * a.Wo;
  const size_t base = (size_t)n0 * per;
  const int tasks = (n1 - n0) * per;
  for (int t = threadIdx.x; t < tasks; t += blockDim.x) {
    const size_t pix = base + t;
    const uint32_t b1 = (a.o1[pix * Q + wq] >> sh) & 0xFF, b2 = (a.o2[pix * Q + wq] >> sh) & 0xFF;
    const uint32_t b3 = (a.o3[pix * Q + wq] >> sh) & 0xFF, b4 = (a.o4[pix * Q + wq] >> sh) & 0xFF;
    const uint32_t i0 = (b1 & 15) | ((b2 & 15) << 4) | ((b3 & 15) << 8) | ((b4 & 15) << 12);
    const uint32_t i1 = (b1 >> 4) | ((b2 >> 4) << 4) | ((b3 >> 4) << 8) | ((b4 >> 4) << 12);
    out_cp[pix * Qout + j] = (uint16_t)(lds[i0] | ((uint32_t)lds[65536 + i1] << 8));
  }
}

// ---- Block_convf of the LAST block: float outputs through a 4 MiB-per-group table ---------
// relu(bn2(conv2(gelu(bn1(conv1(bits)))))) for 16 input bits -> 16 floats is a 64-byte
// row; the table for all 64 groups is 256 MiB and sits in HBM / Infinity Cache.  The
// following AvgPool2d(2) (:197) is fused: four rows are gathered and averaged.
// 16 lanes share one (image, group, pooled pixel) and read one 64-byte row together.
__global__ __launch_bounds__(256) void gate_last_kernel(GateBlockArgs a, const float *t_last, float *feat) {
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
    const size_t pix = ((size_t)n * a.Ho + 2 * py + (d >> 1)) * a.Wo + 2 * px + (d & 1);
    const uint32_t idx = ((a.o1[pix * Q + wq] >> sh) & 15) | (((a.o2[pix * Q + wq] >> sh) & 15) << 4) |
                         (((a.o3[pix * Q + wq] >> sh) & 15) << 8) | (((a.o4[pix * Q + wq] >> sh) & 15) << 12);
    v[d] = tab[(size_t)idx * 16 + k];
  }
  feat[task * 16 + k] = (((v[0] + v[1]) + v[2]) + v[3]) * 0.25f;
}

// ---- layout conversions --------------------------------------------------------------------
__global__ void cp_to_rp_kernel(const uint16_t *cp, uint64_t *rp, int n, int C, int H, int W) {
  const int Q = C / 16;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * Q * H) return;
  const int y = t % H, q = (t / H) % Q, img = t / ((size_t)H * Q);
  uint64_t rows[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) rows[k] = 0;
  for (int x = 0; x < W; ++x) {
    const uint32_t w = cp[(((size_t)img * H + y) * W + x) * Q + q];
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
    cp[(((size_t)img * H + y) * W + x) * Q + q] = (uint16_t)w;
  }
}

__global__ void feat_to_ref_kernel(const float *feat, float *out, int n, int G, int PP) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)G * PP * 16;
  if (t >= (size_t)n * per) return;
  const int k = t % 16, pp = (t / 16) % PP, g = (t / (16 * (size_t)PP)) % G;
  const size_t img = t / per;
  out[img * per + ((size_t)(16 * g + k)) * PP + pp] = feat[t];
}

int slices_for(int n, int units, int *imgs_per_slice) {
  // enough workgroups to cover the 256 CUs about twice, without slicing finer than 1 image
  int want = (512 + units - 1) / units;
  if (want < 1) want = 1;
  if (want > n) want = n;
  const int ips = (n + want - 1) / want;
  *imgs_per_slice = ips;
  return (n + ips - 1) / ips;
}

template <typename K>
int allow_big_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    TT_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return TTNET_OK;
}

}  // namespace

int launch_gate_dw(const GateBlockArgs &a, hipStream_t s) {
  if (a.C % 16 || a.W + 2 * a.pad > 64 || a.Wo > 64 || a.kh1 != a.kh2 || a.kw1 != a.kw2) {
    set_error("gate_dw: unsupported geometry C=%d W=%d pad=%d", a.C, a.W, a.pad);
    return TTNET_E_UNSUPPORTED;
  }
  const int nb = a.kh1 * a.kw1;
  const int tb = (1 << nb) >= 32 ? (1 << nb) / 8 : 4;
  const size_t lds = (size_t)16 * tb;
  int ips;
  const int slices = slices_for(a.n, (a.C / 16) * 2, &ips);
  dim3 grid(a.C / 16, 2, slices);
  if (a.kh1 == 4 && a.kw1 == 4) {
    TT_TRY(allow_big_lds(gate_dw_kernel<4, 4>, lds));
    hipLaunchKernelGGL((gate_dw_kernel<4, 4>), grid, dim3(kGateThreads), lds, s, a, tb, ips);
  } else if (a.kh1 == 2 && a.kw1 == 2) {
    hipLaunchKernelGGL((gate_dw_kernel<2, 2>), grid, dim3(kGateThreads), lds, s, a, tb, ips);
  } else {
    set_error("gate_dw: no kernel for %dx%d windows", a.kh1, a.kw1);
    return TTNET_E_UNSUPPORTED;
  }
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_gate_pw(const GateBlockArgs &a, hipStream_t s) {
  int ips;
  const int slices = slices_for(a.n, a.C / 16, &ips);
  TT_TRY(allow_big_lds(gate_pw_kernel, 131072));
  hipLaunchKernelGGL(gate_pw_kernel, dim3(a.C / 16, slices), dim3(kGateThreads), 131072, s, a, ips);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_gate_pf(const GateBlockArgs &a, const uint8_t *t_cf, uint16_t *out_cp, hipStream_t s) {
  int ips;
  const int slices = slices_for(a.n, a.C / 8, &ips);
  TT_TRY(allow_big_lds(gate_pf_kernel, 131072));
  hipLaunchKernelGGL(gate_pf_kernel, dim3(a.C / 8, slices), dim3(kGateThreads), 131072, s, a, t_cf, out_cp, ips);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_gate_last(const GateBlockArgs &a, const float *t_last, float *feat, hipStream_t s) {
  const size_t tasks = (size_t)a.n * (a.C / 4) * (a.Ho / 2) * (a.Wo / 2);
  const size_t threads = tasks * 16;
  hipLaunchKernelGGL(gate_last_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, t_last, feat);
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

int launch_feat_to_reference_order(const float *feat, float *out, int n, int G, int PP, hipStream_t s) {
  const size_t t = (size_t)n * G * PP * 16;
  hipLaunchKernelGGL(feat_to_ref_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, feat, out, n, G, PP);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

}  // namespace ttnet
