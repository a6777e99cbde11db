// Does VALU work of one wave overlap the MFMAs of another wave on the same SIMD (gfx950)?
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int DEP>
__global__ __launch_bounds__(512) void k(float *out, int mode, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  float r = 0.f;
  if (mf && (mode & 1)) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x = {}, y = {};
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
      if (DEP) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
      } else {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
      }
    }
    for (int i = 0; i < 16; ++i) r += a0[i] + a1[i] + a2[i] + a3[i];
  }
  if (!mf && (mode & 2)) {
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {      // 32 VALU per iteration = the issue time of 4 MFMAs (4 x 32 cycles)
        v0 = fmaf(v0, 1.0001f, 0.5f);
        v1 = fmaf(v1, 1.0001f, 0.5f);
        v2 = fmaf(v2, 1.0001f, 0.5f);
        v3 = fmaf(v3, 1.0001f, 0.5f);
      }
    }
    r = v0 + v1 + v2 + v3;
  }
  if (mf && mode == 4) {            // same wave: each MFMA followed by 8 independent VALU
    f32x16 a0 = {}, a1 = {};
    bf16x8 x = {}, y = {};
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)1.0f; }
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
        v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
    }
    for (int i = 0; i < 16; ++i) r += a0[i] + a1[i];
    r += v0 + v1 + v2 + v3;
  }
  if (mode == 5) {                  // two waves again, MFMA wave at low priority and with idle issue slots between MFMAs
    if (mf) {
      f32x16 a0 = {};
      bf16x8 x = {}, y = {};
      for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)1.0f; }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
          asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
        }
      }
      for (int i = 0; i < 16; ++i) r += a0[i];
    } else {
      __builtin_amdgcn_s_setprio(3);
      float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
        }
      }
      r = v0 + v1 + v2 + v3;
    }
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}

template <int DEP>
void run(float *out) {
  const int iters = 20000;
  for (int mode = 1; mode <= 5; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<DEP>, dim3(256), dim3(512), 0, 0, out, mode, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEP>, dim3(256), dim3(512), 0, 0, out, mode, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dep=%d mode=%d (%s): %.3f ms  -> %.1f cycles@2.4GHz per iteration (4 MFMA = 128 pipe cycles; 32 VALU = 128 issue cycles)\n",
           DEP, mode, mode == 1 ? "mfma only" : mode == 2 ? "valu only" : mode == 3 ? "both" : mode == 4 ? "same wave interleaved" : "two waves, nops+prio", ms, ms * 1e-3 * 2.4e9 / iters);
  }
}
int main() {
  float *out; hipMalloc(&out, 4096);
  run<0>(out); run<1>(out);
  return 0;
}
