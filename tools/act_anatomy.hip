// tools/act_anatomy.hip -- why is fetching the small shared activation vector slow? Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

// VAR 0: no act loads. 1: broadcast pattern (16 distinct 32B per wave instr, every wave the same 8KB).
// 2: coalesced per-lane-distinct loads from the same 8KB (lane*8 floats, wave offset). 3: one wave stages 8KB into LDS
// (coalesced), barrier, all waves read LDS. 4: broadcast pattern but only wave 0 loads. 5: all 8 waves cooperatively stage 8KB
// to LDS (1KB each), barrier, read LDS. 6: like 1 but from per-WG private copy (no sharing across WGs).
template <int VAR>
__global__ __launch_bounds__(512) void k_act(const float *__restrict__ act, float *__restrict__ sink, const float *__restrict__ priv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float lds[2048];
  __shared__ float red[8][64];
  float s = 0.f;
  if (VAR == 1 || VAR == 4 || VAR == 6) {
    if (VAR != 4 || wave == 0) {
      const float *base = VAR == 6 ? priv + (size_t)blockIdx.x * 2048 : act;
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const float *ap = base + ((lane & 3) * 512 + (wave * 2 + c) * 32 + (lane >> 4) * 8);
        const float4 a = *reinterpret_cast<const float4 *>(ap), b = *reinterpret_cast<const float4 *>(ap + 4);
        s += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
      }
    }
  } else if (VAR == 2) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const float *ap = act + ((wave * 2 + c) * 512 + lane * 8) % 2048;
      const float4 a = *reinterpret_cast<const float4 *>(ap), b = *reinterpret_cast<const float4 *>(ap + 4);
      s += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    }
  } else if (VAR == 3 || VAR == 5) {
    if (VAR == 3) {
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 8; c++) *reinterpret_cast<float4 *>(&lds[c * 256 + lane * 4]) = *reinterpret_cast<const float4 *>(act + c * 256 + lane * 4);
      }
    } else {
      *reinterpret_cast<float4 *>(&lds[wave * 256 + lane * 4]) = *reinterpret_cast<const float4 *>(act + wave * 256 + lane * 4);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const float *ap = lds + ((lane & 3) * 512 + (wave * 2 + c) * 32 + (lane >> 4) * 8);
      const float4 a = *reinterpret_cast<const float4 *>(ap), b = *reinterpret_cast<const float4 *>(ap + 4);
      s += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    }
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) { for (int w = 1; w < 8; w++) s += red[w][lane]; if (lane == 0) sink[blockIdx.x] = s; }
}

template <class F> float time_chain(hipStream_t st, int n, F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t g; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; i++) launch(i);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
  float best = 1e9;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(exec));
  return best * 1e3f / n;
}
template <int VAR> void run(hipStream_t st, const char *name, float *act, float *sink, float *priv, int grid) {
  float t = time_chain(st, 200, [&](int) { hipLaunchKernelGGL(k_act<VAR>, dim3(grid), dim3(512), 0, st, (const float *)act, sink, (const float *)priv); });
  printf("%-70s grid=%3d : %.2f us/kernel\n", name, grid, t);
}
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *act, *sink, *priv;
  CK(hipMalloc(&act, 2048 * 4)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&priv, (size_t)256 * 2048 * 4));
  CK(hipMemset(act, 0, 2048 * 4)); CK(hipMemset(priv, 0, (size_t)256 * 2048 * 4));
  for (int grid : {200, 32}) {
    run<0>(st, "no act loads", act, sink, priv, grid);
    run<1>(st, "broadcast pattern, every wave (as the step kernels do)", act, sink, priv, grid);
    run<2>(st, "coalesced per-lane-distinct loads of the same 8 KB", act, sink, priv, grid);
    run<4>(st, "broadcast pattern, wave 0 only", act, sink, priv, grid);
    run<6>(st, "broadcast pattern from a per-WG private copy", act, sink, priv, grid);
    run<3>(st, "wave 0 stages 8 KB to LDS, all waves read LDS", act, sink, priv, grid);
    run<5>(st, "8 waves stage 1 KB each to LDS, all waves read LDS", act, sink, priv, grid);
  }
  return 0;
}
