// hipGraph capture on a side stream + per-launch pointer patching with hipGraphExecKernelNodeSetParams,
// the pattern ttnet_forward uses (plan.hip), on harmless buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
__global__ void k_first(const float *x, float *tmp, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) tmp[i] = x[i] * 2.f; }
__global__ void k_mid(float *tmp, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) tmp[i] += 1.f; }
__global__ void k_last(const float *tmp, float scale, float *out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = tmp[i] * scale; }
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("FAIL %s: %s\n", #e, hipGetErrorString(r)); return 1; } } while (0)
int main() {
  const int n = 1024;
  float *x[2], *out[2], *tmp;
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&x[i], n * 4)); CK(hipMalloc(&out[i], n * 4)); }
  CK(hipMalloc(&tmp, n * 4));
  float h[n];
  for (int i = 0; i < n; ++i) h[i] = 1.f; CK(hipMemcpy(x[0], h, n * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < n; ++i) h[i] = 10.f; CK(hipMemcpy(x[1], h, n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(out[0], 0, n * 4)); CK(hipMemset(out[1], 0, n * 4));
  hipStream_t cap; CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_first, dim3(4), dim3(256), 0, cap, x[0], tmp, n);
  hipLaunchKernelGGL(k_mid, dim3(4), dim3(256), 0, cap, tmp, n);
  hipLaunchKernelGGL(k_last, dim3(4), dim3(256), 0, cap, tmp, 3.f, out[0], n);
  hipGraph_t g; CK(hipStreamEndCapture(cap, &g));
  hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  hipGraphNode_t node; size_t cnt = 1; CK(hipGraphGetRootNodes(g, &node, &cnt));
  printf("roots %zu\n", cnt);
  hipGraphNode_t first = node;
  for (;;) { size_t nd = 0; CK(hipGraphNodeGetDependentNodes(node, nullptr, &nd)); if (!nd) break; hipGraphNode_t nx; CK(hipGraphNodeGetDependentNodes(node, &nx, &nd)); node = nx; }
  hipGraphNode_t last = node;
  hipKernelNodeParams pf{}, pl{};
  CK(hipGraphKernelNodeGetParams(first, &pf)); CK(hipGraphKernelNodeGetParams(last, &pl));
  printf("first: kernelParams=%p extra=%p; last: kernelParams=%p extra=%p\n", (void *)pf.kernelParams, (void *)pf.extra, (void *)pl.kernelParams, (void *)pl.extra);
  if (!pf.kernelParams || !pl.kernelParams) { printf("no kernelParams: patching unsupported\n"); return 2; }
  uint64_t fa[3] = {0, 0, 0}, la[4] = {0, 0, 0, 0}; void *fp[3], *lp[4];
  const int fs[3] = {8, 8, 4}, ls[4] = {8, 4, 8, 4};
  for (int i = 0; i < 3; ++i) { memcpy(&fa[i], pf.kernelParams[i], fs[i]); fp[i] = &fa[i]; }
  for (int i = 0; i < 4; ++i) { memcpy(&la[i], pl.kernelParams[i], ls[i]); lp[i] = &la[i]; }
  printf("arg check: x %d out %d n %d\n", fa[0] == (uint64_t)(uintptr_t)x[0], la[2] == (uint64_t)(uintptr_t)out[0], (int)la[3] == n);
  if (fa[0] != (uint64_t)(uintptr_t)x[0] || la[2] != (uint64_t)(uintptr_t)out[0]) { printf("layout differs: stop\n"); return 3; }
  pf.kernelParams = fp; pf.extra = nullptr; pl.kernelParams = lp; pl.extra = nullptr;
  CK(hipGraphLaunch(ex, 0)); CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, out[0], n * 4, hipMemcpyDeviceToHost)); printf("replay 1: out0[5] = %g (want 9)\n", h[5]);
  fa[0] = (uint64_t)(uintptr_t)x[1]; la[2] = (uint64_t)(uintptr_t)out[1];
  CK(hipGraphExecKernelNodeSetParams(ex, first, &pf)); CK(hipGraphExecKernelNodeSetParams(ex, last, &pl));
  CK(hipGraphLaunch(ex, 0)); CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, out[1], n * 4, hipMemcpyDeviceToHost)); printf("replay 2 (patched): out1[5] = %g (want 63)\n", h[5]);
  fa[0] = (uint64_t)(uintptr_t)x[0];
  CK(hipGraphExecKernelNodeSetParams(ex, first, &pf));
  CK(hipGraphLaunch(ex, 0)); CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, out[1], n * 4, hipMemcpyDeviceToHost)); printf("replay 3 (x patched back): out1[5] = %g (want 9)\n", h[5]);
  return 0;
}
