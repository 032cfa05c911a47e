// tools/mfma_bf16_peak.hip -- what do v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 sustain on this chip (random data,
// one wave per SIMD, independent accumulators)?  The ceiling of the bf16x3 fold product (klstm_fold3.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); return 1;} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NA, int NBB>
__global__ __launch_bounds__(256) void k16(float *out, long long *clk, int iters, const bf16x8 *src) {
  f32x4 acc[NA][NBB];
  for (int i = 0; i < NA; i++) for (int j = 0; j < NBB; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
  bf16x8 a[NA], b[NBB];
  for (int i = 0; i < NA; i++) a[i] = src[(threadIdx.x * 16 + i) % 4096];
  for (int i = 0; i < NBB; i++) b[i] = src[(threadIdx.x * 16 + 8 + i) % 4096];
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j < NBB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < NA; i++) for (int j = 0; j < NBB; j++) s += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int NA, int NBB>
__global__ __launch_bounds__(256) void k32(float *out, long long *clk, int iters, const bf16x8 *src) {
  f32x16 acc[NA][NBB];
  for (int i = 0; i < NA; i++) for (int j = 0; j < NBB; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0;
  bf16x8 a[NA], b[NBB];
  for (int i = 0; i < NA; i++) a[i] = src[(threadIdx.x * 16 + i) % 4096];
  for (int i = 0; i < NBB; i++) b[i] = src[(threadIdx.x * 16 + 8 + i) % 4096];
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j < NBB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < NA; i++) for (int j = 0; j < NBB; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
  float *out; long long *clk; bf16x8 *src;
  CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&clk, 1024 * 16)); CK(hipMalloc(&src, 4096 * 16));
  std::vector<unsigned short> h(4096 * 8);
  for (auto &v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto kern, int grid, int nm, double flop_per) {
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, clk, 10, src);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, clk, iters, src);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * nm;
    printf("%-44s grid %4d: %.1f clk per MFMA, %.2f GHz, %.0f TFLOP/s\n", name, grid, c[0] / n, c[0] / (c[1] * 10.0) , n * flop_per * grid * 4 / (ms * 1e-3) / 1e12);
    return 0;
  };
  for (int grid : {256, 512}) {
    run("16x16x32 bf16, 4x3 accumulators", k16<4, 3>, grid, 12, 16384.0);
    run("16x16x32 bf16, 2x2 accumulators", k16<2, 2>, grid, 4, 16384.0);
    run("32x32x16 bf16, 2x2 accumulators", k32<2, 2>, grid, 4, 32768.0);
    run("32x32x16 bf16, 2x1 accumulators", k32<2, 1>, grid, 2, 32768.0);
  }
  return 0;
}
