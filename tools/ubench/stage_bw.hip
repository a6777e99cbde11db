// Microbenchmark: how fast can one workgroup per CU copy 128 KiB from an L2-resident buffer into
// LDS?  Variants: direct global->LDS (global_load_lds_dwordx4) vs register staging
// (global_load_dwordx4 + ds_write_b128), 256 / 512 / 1024 threads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int MODE>
__global__ void stage_kernel(const uint8_t *src, size_t src_bytes, int bytes, int reps, uint32_t *sink) {
  extern __shared__ __align__(16) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t acc = 0;
  for (int r = 0; r < reps; ++r) {
    const uint8_t *s = src + ((size_t)(blockIdx.x * 7 + r * 2053) * bytes) % (src_bytes - bytes);
    if (MODE == 0) {
      for (int c = wave; c < bytes / 1024; c += nw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s + (size_t)c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + c * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      constexpr int U = 8;
      for (int c0 = wave; c0 < bytes / 1024; c0 += nw * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = c0 + u * nw;
          if (c < bytes / 1024) v[u] = *(const uint4 *)(s + (size_t)c * 1024 + lane * 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = c0 + u * nw;
          if (c < bytes / 1024) *(uint4 *)(lds + c * 1024 + lane * 16) = v[u];
        }
      }
    }
    __syncthreads();
    acc += ((uint32_t *)lds)[(threadIdx.x * 33 + r) % (bytes / 4)];
    __syncthreads();
  }
  if (acc == 0x12345678) sink[0] = acc;
}

int main() {
  uint8_t *src; uint32_t *sink;
  hipMalloc(&src, (size_t)1 << 30); hipMalloc(&sink, 4);
  hipMemset(src, 1, (size_t)1 << 30);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int bytes = 128 * 1024, reps = 20;
  for (size_t src_bytes : {(size_t)2 << 20, (size_t)24 << 20, (size_t)120 << 20, (size_t)1 << 30})
  for (int mode = 0; mode < 1; ++mode)
    for (int threads : {256, 1024})
      for (int blocks : {256}) {
        auto k = mode == 0 ? stage_kernel<0> : stage_kernel<1>;
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), bytes, 0, src, src_bytes, bytes, reps, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), bytes, 0, src, src_bytes, bytes, reps, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_stage_us = ms * 1e3 / reps / (blocks / 256.0);
        printf("src %4zu MiB mode %s threads %4d blocks %3d: %.2f us per 128 KiB stage per CU -> %.1f GB/s per CU, %.2f TB/s chip\n",
               src_bytes >> 20, mode == 0 ? "glds" : "regs", threads, blocks, per_stage_us, bytes / per_stage_us / 1e3,
               256.0 * bytes / per_stage_us / 1e6);
      }
  return 0;
}
