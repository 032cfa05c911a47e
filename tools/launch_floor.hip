// tools/launch_floor.hip -- measures the dependent-kernel boundary cost on this box (eager vs
// hipGraph, block size, grid size, kernarg size, dependent-load round trips).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

struct BigArgs { float *p; int n; int pad[26]; };
__global__ void k_empty() {}
__global__ void k_args(BigArgs a) { if (a.n < 0) a.p[0] = 1.f; }
// each block reads what the previous kernel's block (another XCD, shifted) wrote: one dependent round trip
__global__ void k_dep(const float *__restrict__ src, float *__restrict__ dst, int n, int hops) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = (i + 4096 + blockDim.x) % n;
  float v = src[j];
  for (int h = 1; h < hops; h++) { j = (j + 8192 + (int)(v * 1e-30f)) % n; v += src[j]; }
  dst[i % n] = v + 1.f;
}

template <class F> float time_chain(hipStream_t st, int n, bool graph, F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraphExec_t exec = nullptr;
  if (graph) {
    hipGraph_t g; CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; i++) launch(i);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
  }
  float best = 1e9;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0, st));
    if (graph) CK(hipGraphLaunch(exec, st)); else for (int i = 0; i < n; i++) launch(i);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  if (exec) CK(hipGraphExecDestroy(exec));
  return best * 1e3f / n;
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int N = 200;
  float *a, *b; const int n = 1 << 20; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  printf("HIP_FORCE_DEV_KERNARG=%s\n", getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "(unset)");
  for (int graph = 0; graph < 2; graph++) {
    for (int grid : {1, 64, 256, 1024}) for (int block : {64, 256, 512}) {
      float t = time_chain(st, N, graph, [&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, st); });
      printf("%s empty      grid=%4d block=%3d : %.2f us/kernel\n", graph ? "graph" : "eager", grid, block, t);
    }
    BigArgs ba; ba.p = a; ba.n = 1;
    for (int grid : {1, 256}) {
      float t = time_chain(st, N, graph, [&](int) { hipLaunchKernelGGL(k_args, dim3(grid), dim3(256), 0, st, ba); });
      printf("%s args128B   grid=%4d block=256 : %.2f us/kernel\n", graph ? "graph" : "eager", grid, t);
    }
    for (int hops : {1, 2, 3, 8}) for (int grid : {64, 256}) {
      float t = time_chain(st, N, graph, [&](int i) {
        hipLaunchKernelGGL(k_dep, dim3(grid), dim3(256), 0, st, (const float *)(i & 1 ? b : a), i & 1 ? a : b, n, hops); });
      printf("%s dep-load hops=%d grid=%4d block=256 : %.2f us/kernel\n", graph ? "graph" : "eager", hops, grid, t);
    }
  }
  return 0;
}
