// tools/gridbar_probe.hip -- what an in-kernel grid barrier costs on this box: G co-resident workgroups of NT threads run STEPS
// barrier phases; in each phase every workgroup writes WORDS floats that depend on what it read in the previous phase (sc1 stores),
// arrives -- on ONE monotonic counter (a device-scope atomic per workgroup) or on its OWN phase-tagged flag that every other workgroup
// watches (no shared address, no atomic; thread t polls flag t) --, spins (bounded) until all G have arrived, then reads WORDS floats another workgroup wrote
// (sc1 loads: no cache-wide invalidate).  Prices the "launch-free chain" of docs/DESIGN_rounds_1-4.md 9 item 1 against the 4.2-5.0 us a dependent
// launch costs.  Diagnostic only; not part of libklstm.
//   usage: gridbar_probe [steps=200] [threads=256] [words_per_thread=4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ __forceinline__ void st_sc1(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ float ld_sc1(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool FLAGS>
__global__ void k_gridbar(float *buf, unsigned *counter, unsigned *flags, int steps, int words, unsigned *tmo, float *out, long long *clk) {
  const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int per = NT * words;                                 // floats per workgroup and phase
  float v = 1.f + 1e-3f * (float)(b * NT + tid);
  const long long t0 = wall_clock64();
  for (int s = 0; s < steps; s++) {
    float *mine = buf + ((size_t)(s & 1) * G + b) * per;
    for (int w = 0; w < words; w++) __hip_atomic_store(mine + w * NT + tid, v + (float)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);                            // own stores issued and acknowledged
    __syncthreads();
    if (FLAGS) {
      // one flag per workgroup, tagged with the phase; thread t watches the flags t, t + NT, ... (no shared address, no atomic)
      if (tid == 0) __hip_atomic_store(flags + b, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int f = tid; f < G; f += NT) {
        long long spins = 0;
        while (__hip_atomic_load(flags + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(s + 1)) {
          if (++spins > 4000000) { atomicAdd(tmo, 1u); break; }
        }
      }
    } else if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(s + 1) * (unsigned)G;
      long long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > 4000000) { atomicAdd(tmo, 1u); break; }
      }
    }
    __syncthreads();
    const float *theirs = buf + ((size_t)(s & 1) * G + (b + 1 + s) % G) * per;
    float acc = 0.f;
    for (int w = 0; w < words; w++) acc += ld_sc1(theirs + w * NT + tid);
    v = 0.5f * v + 1e-6f * acc;
  }
  if (tid == 0) clk[b] = wall_clock64() - t0;
  out[b * NT + tid] = v;
}

int main(int argc, char **argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 200, NT = argc > 2 ? atoi(argv[2]) : 256, words = argc > 3 ? atoi(argv[3]) : 4;
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_gridbar<false>, NT, 0));
  printf("%s: %d CUs, %d workgroups of %d threads resident per CU\n", pr.name, pr.multiProcessorCount, occ, NT);
  for (int flagsmode = 0; flagsmode < 2; flagsmode++)
  for (int mult = 1; mult <= (occ < 4 ? occ : 4); mult *= 2) {
    const int G = pr.multiProcessorCount * mult;
    float *buf, *out; unsigned *counter, *tmo, *flags; long long *clk;
    CK(hipMalloc(&buf, (size_t)2 * G * NT * words * sizeof(float))); CK(hipMemset(buf, 0, (size_t)2 * G * NT * words * sizeof(float)));
    CK(hipMalloc(&out, (size_t)G * NT * sizeof(float)));
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&flags, (size_t)G * 4)); CK(hipMalloc(&clk, (size_t)G * 8));
    float best = 1e9f;
    unsigned htmo = 0;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipMemset(counter, 0, 4)); CK(hipMemset(tmo, 0, 4)); CK(hipMemset(flags, 0, (size_t)G * 4));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (flagsmode) hipLaunchKernelGGL(k_gridbar<true>, dim3(G), dim3(NT), 0, 0, buf, counter, flags, steps, words, tmo, out, clk);
      else hipLaunchKernelGGL(k_gridbar<false>, dim3(G), dim3(NT), 0, 0, buf, counter, flags, steps, words, tmo, out, clk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
      if (ms < best) best = ms;
    }
    printf("%s  G = %4d workgroups (%d per CU), %d steps, %d KB exchanged per step: %.2f us per barrier phase%s\n",
           flagsmode ? "one tagged flag per workgroup" : "one shared arrival counter   ", G, mult, steps,
           (int)((size_t)G * NT * words * 4 / 1024), best * 1e3f / steps, htmo ? "  [TIMEOUT: not co-resident]" : "");
    CK(hipFree(buf)); CK(hipFree(out)); CK(hipFree(counter)); CK(hipFree(tmo)); CK(hipFree(clk)); CK(hipFree(flags));
  }
  return 0;
}
