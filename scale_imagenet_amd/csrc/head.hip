// Classifier head: lin1 (no bias) -> BatchNorm1d -> 0.47 + 0.50 x + 0.09 x^2 -> lin2 (+bias)
// (Classifier_scale.forward, models/TT_general_imagenet_v2_small.py:229-236;
//  Polynome_ACT.forward :213-215).
//
// Both linears are C[M][N] = A[M][K] * B[N][K]^T with K contiguous on both sides, run on
// the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (the 1e-5 logit tolerance rules
// out bf16 operands).  M = images is small (256), so K is split across
// workgroups to fill the 256 CUs; partial slabs are summed by the fused epilogue kernels
// in a fixed order (bitwise reproducible, no float atomics).
//
// Bound: fp32 MFMA, 17.4 MMAC per image (157 TFLOP/s peak); lin1's 65.5 MB of weights are
// read once per batch.

#include "ttnet_common.h"

namespace ttnet {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32, LDT = BM + 1;

__global__ __launch_bounds__(256) void gemm_nt_splitk_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                            float *__restrict__ part, int M, int N, int K,
                                                            int kper) {
  __shared__ float As[BK][LDT];
  __shared__ float Bs[BK][LDT];
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * kper, kend = min(K, kbeg + kper);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lrow = tid >> 3, lk = (tid & 7) * 4;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = lrow + 32 * i;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      const int k = k0 + lk;
      if (k < kend) {   // K and kper are multiples of 4
        if (m0 + row < M) va = *(const float4 *)(A + (size_t)(m0 + row) * K + k);
        if (n0 + row < N) vb = *(const float4 *)(B + (size_t)(n0 + row) * K + k);
      }
      As[lk + 0][row] = va.x; As[lk + 1][row] = va.y; As[lk + 2][row] = va.z; As[lk + 3][row] = va.w;
      Bs[lk + 0][row] = vb.x; Bs[lk + 1][row] = vb.y; Bs[lk + 2][row] = vb.z; Bs[lk + 3][row] = vb.w;
    }
    __syncthreads();
    const int kk0 = lane >> 5, ri = wr * 32 + (lane & 31), ci = wc * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[kk + kk0][ri];
      const float b = Bs[kk + kk0][ci];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float *dst = part + (size_t)blockIdx.z * M * N;
  const int col = n0 + wc * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M && col < N) dst[(size_t)row * N + col] = acc[r];
  }
}

__global__ void head_mid_kernel(const float *__restrict__ part, int splits, const float *__restrict__ scale,
                                const float *__restrict__ shift, float *__restrict__ out, int M, int N) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int nidx = i % N;
  double s = 0.0;
  for (int z = 0; z < splits; ++z) s += (double)part[(size_t)z * M * N + i];
  const float zf = fmaf((float)s, scale[nidx], shift[nidx]);
  // 0.47 + 0.50 * x + 0.09 * x ** 2, evaluated left to right in fp32 like the reference
  const float t = __fadd_rn(0.47f, __fmul_rn(0.50f, zf));
  out[i] = __fadd_rn(t, __fmul_rn(0.09f, __fmul_rn(zf, zf)));
}

__global__ void head_out_kernel(const float *__restrict__ part, int splits, const float *__restrict__ bias,
                                float *__restrict__ out, int M, int N) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  double s = 0.0;
  for (int z = 0; z < splits; ++z) s += (double)part[(size_t)z * M * N + i];
  out[i] = (float)(s + (double)bias[i % N]);
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

int launch_gemm_nt_splitk(const float *A, const float *B, float *part, int M, int N, int K, int splits,
                          hipStream_t s) {
  if (K % 4) {
    set_error("gemm: K=%d must be a multiple of 4", K);
    return TTNET_E_UNSUPPORTED;
  }
  int kper = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, splits);
  hipLaunchKernelGGL(gemm_nt_splitk_kernel, grid, dim3(256), 0, s, A, B, part, M, N, K, kper);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_head_mid(const float *part, int splits, const float *scale, const float *shift, float *out, int M,
                    int N, hipStream_t s) {
  const size_t t = (size_t)M * N;
  hipLaunchKernelGGL(head_mid_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, part, splits, scale, shift,
                     out, M, N);
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}

int launch_head_out(const float *part, int splits, const float *bias, float *out, int M, int N, hipStream_t s) {
  const size_t t = (size_t)M * N;
  hipLaunchKernelGGL(head_out_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, part, splits, bias, out, M,
                     N);
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
