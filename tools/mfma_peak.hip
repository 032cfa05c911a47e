// tools/mfma_peak.hip -- what does v_mfma_f32_16x16x4_f32 sustain on this chip (fp32 MFMA ceiling used in bench.py's roofline),
// and at which shader clock?  One wave per SIMD (grid = 256 CUs x 4 waves) or two; NB independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); return 1;} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NB>
__global__ __launch_bounds__(256) void k_peak(float *out, long long *clk, int iters, float a0, float b0) {
  f32x4 acc[NB];
  for (int i = 0; i < NB; i++) acc[i] = (f32x4){0, 0, 0, 0};
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int i = 0; i < NB; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < NB; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

// same, with 2 + 5 operand registers per k-step loaded from memory (random data), 10 accumulators: the register pattern of the fold kernel
__global__ __launch_bounds__(256) void k_peak_data(float *out, long long *clk, int iters, const float *src) {
  f32x4 acc[10];
  for (int i = 0; i < 10; i++) acc[i] = (f32x4){0, 0, 0, 0};
  float a[2][8], b[5][8];
  for (int i = 0; i < 2; i++) for (int e = 0; e < 8; e++) a[i][e] = src[(threadIdx.x * 56 + i * 8 + e) % 65536];
  for (int i = 0; i < 5; i++) for (int e = 0; e < 8; e++) b[i][e] = src[(threadIdx.x * 56 + 16 + i * 8 + e) % 65536];
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int e = 0; e < 8; e++)
#pragma unroll
      for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 5; ni++) acc[mi * 5 + ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][e], b[ni][e], acc[mi * 5 + ni], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 10; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

// accumulators forced into AGPRs ("+a") vs VGPRs ("+v"), operands in VGPRs
template <bool AG>
__global__ __launch_bounds__(256) void k_peak_regs(float *out, long long *clk, int iters, const float *src) {
  f32x4 acc[10];
  for (int i = 0; i < 10; i++) acc[i] = (f32x4){0, 0, 0, 0};
  float a[2][8], b[5][8];
  for (int i = 0; i < 2; i++) for (int e = 0; e < 8; e++) a[i][e] = src[(threadIdx.x * 56 + i * 8 + e) % 65536];
  for (int i = 0; i < 5; i++) for (int e = 0; e < 8; e++) b[i][e] = src[(threadIdx.x * 56 + 16 + i * 8 + e) % 65536];
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int e = 0; e < 8; e++)
#pragma unroll
      for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 5; ni++) {
          if (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[mi * 5 + ni]) : "v"(a[mi][e]), "v"(b[ni][e]));
          else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[mi * 5 + ni]) : "v"(a[mi][e]), "v"(b[ni][e]));
        }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 10; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
  float *out; long long *clk;
  CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&clk, 1024 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {256, 512}) for (int iters : {1280, 12800}) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_peak<10>, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.0f, 0.5f);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
      const double flop = (double)grid * 4 * iters * 40 * 2048;
      printf("grid %d x 4 waves, %d x 40 MFMA per wave: %.1f us, %.1f TFLOP/s; wave 0: %lld shader clocks in %.2f us = %.0f MHz, %.1f clocks per MFMA\n",
             grid, iters, ms * 1e3, flop / (ms * 1e-3) / 1e12, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0), (double)h[0] / (iters * 40.0));
    }
  }
  float *src; CK(hipMalloc(&src, 65536 * 4));
  for (int mode = 0; mode < 3; mode++) {
    static float h[65536];
    for (int i = 0; i < 65536; i++) h[i] = mode == 0 ? 0.f : mode == 1 ? 1.0f : (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    CK(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
      const int iters = 640;
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_peak_data, dim3(256), dim3(256), 0, 0, out, clk, iters, src);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
      const double flop = 256.0 * 4 * iters * 80 * 2048;
      printf("operands from memory (%s): %.1f us, %.1f TFLOP/s; %.0f MHz, %.1f clocks per MFMA\n", mode == 0 ? "zeros" : mode == 1 ? "ones" : "random",
             ms * 1e3, flop / (ms * 1e-3) / 1e12, hc[0] / (hc[1] / 100.0), (double)hc[0] / (iters * 80.0));
    }
  }
  for (int ag = 0; ag < 2; ag++) for (int rep = 0; rep < 2; rep++) {
    const int iters = 640;
    CK(hipEventRecord(e0, 0));
    if (ag) hipLaunchKernelGGL(k_peak_regs<true>, dim3(256), dim3(256), 0, 0, out, clk, iters, src);
    else hipLaunchKernelGGL(k_peak_regs<false>, dim3(256), dim3(256), 0, 0, out, clk, iters, src);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
    printf("accumulators in %s: %.1f us; %.0f MHz, %.1f clocks per MFMA\n", ag ? "AGPRs" : "VGPRs", ms * 1e3, hc[0] / (hc[1] / 100.0), (double)hc[0] / (iters * 80.0));
  }
  return 0;
}
