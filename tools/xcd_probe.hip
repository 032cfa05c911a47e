// tools/xcd_probe.hip -- can a chain be confined to ONE XCD, and what does an all-gather step cost there?
// (1) Which XCDs does a CU-masked stream (hipExtStreamCreateWithCUMask) land on, per mask pattern: every workgroup reports
//     HW_REG_XCC_ID.  (2) The xchg_probe exchange (data-tagged 8-byte granules, sc1 stores, sc1 16-byte sweeps) with all
//     workgroups on one XCD, against the same workgroup count spread over the chip; also with PLAIN stores (the line stays
//     in the XCD's L2) + sc1 loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_where(unsigned *xcc) {
  if (threadIdx.x == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;   // HW_REG_XCC_ID
  // stay resident long enough for the whole grid to be co-resident
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) {}
}

template <int PLAIN, int ONEX = 0>
__global__ __launch_bounds__(512) void k_xchg(unsigned long long *gran, int N, int steps, unsigned *bad, unsigned *xcc) {
  extern __shared__ unsigned lds[];
  if (ONEX && (blockIdx.x & 7) != 0) return;          // (workgroup w is dispatched to XCC w % 8: keep those of XCC 0)
  const int tid = threadIdx.x, G = ONEX ? gridDim.x / 8 : gridDim.x, b = ONEX ? blockIdx.x / 8 : blockIdx.x;
  if (tid == 0) xcc[b] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
  const int own0 = (int)((long)N * b / G), own1 = (int)((long)N * (b + 1) / G);
  unsigned sum = 0;
  const long long t_start = wall_clock64();
  for (int e = 1; e <= steps; e++) {
    unsigned long long *slot = gran + (size_t)(e & 1) * N;
    for (int i = own0 + tid; i < own1; i += 512) {
      const unsigned long long v = ((unsigned long long)e << 32) | (sum * 2654435761u + i);
      if (PLAIN == 2) slot[i] = v;
      else if (PLAIN) __builtin_nontemporal_store(v, slot + i), (void)0;
      else __hip_atomic_store(slot + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)slot, 0, N * 8, 0x00020000);
    bool done = false;
    for (unsigned spins = 0; !done; spins++) {
      bool ok = true;
      u32x4 q[4];
#pragma unroll
      for (int l = 0; l < 4; l++) { const int i = 2 * (tid + l * 512); if (i < N) q[l] = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 8, 0, 16); }
#pragma unroll
      for (int l = 0; l < 4; l++) { const int i = 2 * (tid + l * 512); if (i < N) { ok &= q[l].y == (unsigned)e && q[l].w == (unsigned)e; lds[i] = q[l].x; lds[i + 1] = q[l].z; } }
      done = __all(ok);
      if (!done && (spins & 63) == 63 && wall_clock64() - t_start > 5000000LL) { if (tid == 0) atomicExch(bad, (unsigned)e); return; }
    }
    __syncthreads();
    unsigned part = 0;
    for (int i = tid; i < N; i += 512) part += lds[i];
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    __syncthreads();
    if ((tid & 63) == 0) lds[N + (tid >> 6)] = part;
    __syncthreads();
    sum = 0;
    for (int w = 0; w < 8; w++) sum += lds[N + w];
    __syncthreads();
  }
}

int main() {
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  printf("%d CUs\n", ncu);
  unsigned *xcc; CK(hipMalloc(&xcc, 1024 * 4));
  auto masked_stream = [&](const std::vector<unsigned> &mask) { hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data())); return s; };
  auto where = [&](hipStream_t s, int G, const char *name) {
    CK(hipMemset(xcc, 0xff, 1024 * 4));
    hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, s, xcc); CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(G); CK(hipMemcpy(h.data(), xcc, G * 4, hipMemcpyDeviceToHost));
    int cnt[16] = {0}; for (auto v : h) if (v < 16) cnt[v]++;
    printf("%-44s %3d workgroups -> per XCC:", name, G); for (int i = 0; i < 8; i++) printf(" %d", cnt[i]); printf("\n");
  };
  std::vector<unsigned> all(8, 0xffffffffu), first32(8, 0), every8(8, 0), low4of32(8, 0);
  first32[0] = 0xffffffffu;
  for (int i = 0; i < 256; i += 8) every8[i / 32] |= 1u << (i % 32);
  for (int w = 0; w < 8; w++) low4of32[w] = 0xfu;
  hipStream_t s_all = masked_stream(all), s_first = masked_stream(first32), s_e8 = masked_stream(every8), s_l4 = masked_stream(low4of32);
  where(s_all, 256, "mask: all CUs");
  where(s_first, 32, "mask: bits 0..31");
  where(s_e8, 32, "mask: every 8th bit");
  where(s_l4, 32, "mask: low 4 bits of every word");
  // exchange: N = 3200 granules (4 streams x 800 cells), 32 workgroups
  unsigned long long *gran; unsigned *bad; CK(hipMalloc(&gran, 2 * 3200 * 8)); CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto xchg = [&](hipStream_t s, int G, int plain, const char *name) {
    const int N = 3200, steps = 400;
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemsetAsync(gran, 0, 2 * N * 8, s)); CK(hipMemsetAsync(bad, 0, 4, s));
      CK(hipEventRecord(e0, s));
      if (plain) hipLaunchKernelGGL(k_xchg<1>, dim3(G), dim3(512), (N + 16) * 4, s, gran, N, steps, bad, xcc);
      else hipLaunchKernelGGL(k_xchg<0>, dim3(G), dim3(512), (N + 16) * 4, s, gran, N, steps, bad, xcc);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    unsigned hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned> h(G); CK(hipMemcpy(h.data(), xcc, G * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> xs(h.begin(), h.end());
    printf("%-44s G=%3d %s stores: %.3f us per step over %zu XCC(s)%s\n", name, G, plain ? "nt   " : "sc1  ", best * 1e3f / steps, xs.size(), hb ? "  TIMEOUT (stale reads)" : "");
  };
  auto xchg1 = [&](int G, int plain, const char *name) {
    const int N = 3200, steps = 400;
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemsetAsync(gran, 0, 2 * N * 8, s_all)); CK(hipMemsetAsync(bad, 0, 4, s_all));
      CK(hipEventRecord(e0, s_all));
      if (plain == 2) hipLaunchKernelGGL((k_xchg<2, 1>), dim3(G * 8), dim3(512), (N + 16) * 4, s_all, gran, N, steps, bad, xcc);
      else if (plain) hipLaunchKernelGGL((k_xchg<1, 1>), dim3(G * 8), dim3(512), (N + 16) * 4, s_all, gran, N, steps, bad, xcc);
      else hipLaunchKernelGGL((k_xchg<0, 1>), dim3(G * 8), dim3(512), (N + 16) * 4, s_all, gran, N, steps, bad, xcc);
      CK(hipEventRecord(e1, s_all)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    unsigned hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned> h(G); CK(hipMemcpy(h.data(), xcc, G * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> xs(h.begin(), h.end());
    printf("%-44s G=%3d %s stores: %.3f us per step over %zu XCC(s)%s\n", name, G, plain == 2 ? "plain" : plain ? "nt   " : "sc1  ", best * 1e3f / steps, xs.size(), hb ? "  TIMEOUT (stale reads)" : "");
  };
  xchg1(32, 0, "exchange, workgroups of ONE XCC");
  xchg1(16, 0, "exchange, workgroups of ONE XCC");
  xchg1(32, 1, "exchange, workgroups of ONE XCC");
  xchg1(32, 2, "exchange, workgroups of ONE XCC");
  xchg1(16, 2, "exchange, workgroups of ONE XCC");
  xchg(s_all, 32, 0, "exchange, all CUs");
  xchg(s_all, 200, 0, "exchange, all CUs");
  xchg(s_first, 32, 0, "exchange, mask bits 0..31");
  xchg(s_e8, 32, 0, "exchange, mask every 8th bit");
  xchg(s_l4, 32, 0, "exchange, mask low 4 bits of every word");
  xchg(s_first, 32, 1, "exchange, mask bits 0..31");
  xchg(s_e8, 32, 1, "exchange, mask every 8th bit");
  return 0;
}
