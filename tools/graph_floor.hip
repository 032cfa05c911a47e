// tools/graph_floor.hip -- device-side cost of one hipGraphLaunch as a function of node count. Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_small(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int n : {1, 2, 10, 40, 80}) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, p);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int reps : {1, 50}) {
      float best = 1e9;
      for (int trial = 0; trial < 5; trial++) {
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("graph of %2d kernels, %2d back-to-back replays: %.2f us per replay = %.2f us per kernel\n", n, reps, best * 1e3 / reps, best * 1e3 / reps / n);
    }
    // same kernels eagerly
    float best = 1e9;
    for (int trial = 0; trial < 5; trial++) {
      CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st));
      for (int r = 0; r < 50; r++) for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, p);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("eager  %2d kernels x 50: %.2f us per group = %.2f us per kernel\n", n, best * 1e3 / 50, best * 1e3 / 50 / n);
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
  }
  return 0;
}
