// tools/residency_probe.hip -- how many 12-wave / 168-register workgroups (the shape of k_fwd_persist: one per CU) become
// resident AT ONCE next to a foreign kernel that holds N compute units?  Every census workgroup records when it started
// (wall clock) and where (XCC, SE, CU), then spins 2 ms.  "late" = started more than 1 ms after the first one, i.e. had to
// wait for a compute unit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ __launch_bounds__(1024) void k_hold(long long ticks, unsigned *where) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 1.f;
  if (threadIdx.x == 0) { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); where[blockIdx.x] = x + 1; }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
template <int NT>
__global__ __launch_bounds__(NT) void k_census(long long ticks, long long *start, unsigned *hw) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 1.f;
  asm volatile("" ::: "v160");                       // (register footprint of the persistent kernels: nothing else fits on the SIMD)
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    unsigned x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    start[blockIdx.x] = t0; hw[2 * blockIdx.x] = x; hw[2 * blockIdx.x + 1] = h;
  }
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_small(float *p) { p[threadIdx.x] += 1.f; }

int main() {
  hipStream_t sh, sc, s3; CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamDefault));
  CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  long long *start; unsigned *hw, *where; float *scratch;
  CK(hipMalloc(&start, 256 * 8)); CK(hipMalloc(&hw, 512 * 4)); CK(hipHostMalloc(&where, 256 * 4, hipHostMallocMapped)); CK(hipMalloc(&scratch, 4096));
  CK(hipFuncSetAttribute((const void *)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  for (int hold : {0, 40, 48, 56})
    for (int gap : {0, 1, 2})                        // 0: census right away; 1: a host-side sync gap first; 2: a small kernel on another stream + gap
      for (int grid : {200}) {
        for (int i = 0; i < 256; i++) where[i] = 0;
        CK(hipDeviceSynchronize());
        if (hold) {
          hipLaunchKernelGGL(k_hold, dim3(hold), dim3(1024), 96 * 1024, sh, 3000000LL, where);   // 30 ms
          bool all = false;
          while (!all) { all = true; for (int i = 0; i < hold; i++) all &= ((volatile unsigned *)where)[i] != 0; }
        }
        // a first census (like the first persistent launch), then optionally a gap, then the one that is measured
        hipLaunchKernelGGL(k_census<768>, dim3(grid), dim3(768), 25 * 1024, sc, 20000LL, start, hw);
        if (gap >= 1) CK(hipStreamSynchronize(sc));
        if (gap == 2) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, 0, scratch); CK(hipStreamSynchronize(0)); }
        hipLaunchKernelGGL(k_census<768>, dim3(grid), dim3(768), 25 * 1024, sc, 200000LL, start, hw);   // 2 ms
        CK(hipStreamSynchronize(sc));
        std::vector<long long> st(grid); std::vector<unsigned> h(2 * grid);
        CK(hipMemcpy(st.data(), start, grid * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h.data(), hw, 2 * grid * 4, hipMemcpyDeviceToHost));
        const long long t0 = *std::min_element(st.begin(), st.end());
        int late = 0; std::map<int, int> late_xcc;
        for (int i = 0; i < grid; i++) if (st[i] - t0 > 100000) { late++; late_xcc[h[2 * i] & 0xf]++; }
        printf("held %2d, gap mode %d: %3d of %d census workgroups resident at once, %d late", hold, gap, grid - late, grid, late);
        if (late) { printf(" (by XCC:"); for (auto &kv : late_xcc) printf(" %d:%d", kv.first, kv.second); printf(")"); }
        printf("\n");
        CK(hipDeviceSynchronize());
      }
  return 0;
}
