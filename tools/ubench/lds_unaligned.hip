// Does ds_read_b128 / ds_read_b64 at a 4-byte-aligned (not 16 / 8-byte-aligned) LDS address return the right
// bytes on gfx950, and at what rate?  The stem's B fragments are 16 bytes at byte offset 4 * ox.
//   hipcc -O3 --offload-arch=gfx950 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>   // 0: 2 x ds_read2_b32, 1: ds_read_b128 (per-lane offset 4 * lane), 2: 2 x ds_read_b64, 3: ds_read_b128 aligned (16 * lane)
__global__ void k(uint32_t *out, long long *cycles, int iters) {
  __shared__ uint32_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x;
  uint32_t acc = 0;
  const uint32_t base = (uint32_t)(uintptr_t)(lds) + (MODE == 3 ? 16 * lane : 4 * lane);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t a = base + 256 * (it & 15);
    uint32_t x0, x1, x2, x3;
    if (MODE == 0) {
      asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(*(uint64_t *)&x0), "=v"(*(uint64_t *)&x2) : "v"(a));
      uint64_t lo = *(uint64_t *)&x0, hi = *(uint64_t *)&x2;
      x0 = (uint32_t)lo; x1 = (uint32_t)(lo >> 32); x2 = (uint32_t)hi; x3 = (uint32_t)(hi >> 32);
    } else if (MODE == 2) {
      uint64_t lo, hi;
      asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo), "=v"(hi) : "v"(a));
      x0 = (uint32_t)lo; x1 = (uint32_t)(lo >> 32); x2 = (uint32_t)hi; x3 = (uint32_t)(hi >> 32);
    } else {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      u4 v;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
      x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w;
    }
    acc += x0 ^ (x1 * 3u) ^ (x2 * 5u) ^ (x3 * 7u);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name) {
  uint32_t *out; long long *cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 4096;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<uint32_t> h(256); std::vector<long long> c(256);
  hipMemcpy(h.data(), out, 256 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  // reference for lane l
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
      const int d = (MODE == 3 ? 4 * l : l) + 64 * (it & 15);
      auto L = [&](int i) { return (uint32_t)(i * 2654435761u); };
      acc += L(d) ^ (L(d + 1) * 3u) ^ (L(d + 2) * 5u) ^ (L(d + 3) * 7u);
    }
    bad += acc != h[l];
  }
  printf("%-28s wrong lanes %d of 64, %.1f cycles per 16-byte read per wave (4 waves per CU)\n", name, bad, (double)c[0] / iters);
}

int main() {
  run<0>("2 x ds_read2_b32 (4B align)");
  run<2>("2 x ds_read_b64 (4B align)");
  run<1>("ds_read_b128 (4B align)");
  run<3>("ds_read_b128 (16B align)");
  return 0;
}
