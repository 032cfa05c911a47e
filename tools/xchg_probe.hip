// tools/xchg_probe.hip -- prices ONE all-to-all exchange step of a weights-resident (persistent) recurrence chain on
// this box: G co-resident workgroups, each publishes its share of N data-tagged 8-byte granules {tag = step, value}
// (one sc1 store each) and then sweeps ALL N granules until every tag matches (MI355X_MICROARCH.md "allgather" row,
// cdna_hip_programming.md Guideline 16 recipe R2).  The value published at step e+1 depends on a checksum over every
// value of step e, so the chain is a true dependency chain and a stale read is detected by the host-side replay.
// Diagnostic only (decides docs/DESIGN_rounds_1-4.md section 4's persistent-chain question); not part of libklstm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline unsigned mix(unsigned step, unsigned idx, unsigned sum) {
  unsigned x = sum * 2654435761u + idx * 40503u + step * 97u;
  x ^= x >> 13;
  return x * 1274126177u + 1u;
}

// NT threads per workgroup, SW sweeping waves, V granules per load (1: dwordx2, 2: dwordx4)
template <int NT, int V>
__global__ __launch_bounds__(NT) void k_xchg(unsigned long long *gran, int N, int steps, int sweep_waves, int extra_work,
                                               unsigned *tmo, unsigned *final_vals, unsigned long long *cycles) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];   // N values + NT/64 partial sums
  const int tid = threadIdx.x, G = gridDim.x, b = blockIdx.x;
  const int own0 = (int)((long)N * b / G), own1 = (int)((long)N * (b + 1) / G);
  const int nsweep = sweep_waves * 64;
  unsigned sum = 0;
  if (tid == 0) lds[N + 32] = 0u;
  __syncthreads();
  const long long t_start = wall_clock64();
  for (int e = 1; e <= steps; e++) {
    gu64 *slot = (gu64 *)(gran + (size_t)(e & 1) * N);
    // publish own granules
    for (int i = own0 + tid; i < own1; i += NT) {
      const unsigned v = mix(e, i, sum);
      __hip_atomic_store(slot + i, ((unsigned long long)e << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // sweep everything
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(gran + (size_t)(e & 1) * N), 0, N * 8, 0x00020000);
    if (tid < nsweep) {
      bool done = false;
      for (unsigned spins = 0; !done; spins++) {
        bool ok = true;
        // all loads of a pass in flight (buffer loads with the sc1 bit: the compiler tracks vmcnt for the builtins), then
        // the tag checks.  V = 1: one 8-byte granule per load, V = 2: two granules per 16-byte load.
        constexpr int MAXL = 16;
        if (V == 1) {
          u32x2 q[MAXL];
#pragma unroll
          for (int l = 0; l < MAXL; l++) {
            const int i = tid + l * nsweep;
            if (i < N) q[l] = __builtin_amdgcn_raw_buffer_load_b64(rs, i * 8, 0, 16);
          }
#pragma unroll
          for (int l = 0; l < MAXL; l++) {
            const int i = tid + l * nsweep;
            if (i < N) { ok &= q[l].y == (unsigned)e; lds[i] = q[l].x; }
          }
        } else {
          u32x4 q[MAXL];
#pragma unroll
          for (int l = 0; l < MAXL; l++) {
            const int i = 2 * (tid + l * nsweep);
            if (i < N) q[l] = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 8, 0, 16);
          }
#pragma unroll
          for (int l = 0; l < MAXL; l++) {
            const int i = 2 * (tid + l * nsweep);
            if (i < N) { ok &= q[l].y == (unsigned)e && q[l].w == (unsigned)e; lds[i] = q[l].x; lds[i + 1] = q[l].z; }
          }
        }
        done = __all(ok);
        if (!done && (spins & 63) == 63 && wall_clock64() - t_start > 100000000LL / 20) {   // 50 ms at 100 MHz: give up
          if ((tid & 63) == 0) { atomicExch(tmo, 0x80000000u | (unsigned)e); lds[N + 32] = 1u; }
          done = true;
        }
      }
    }
    __syncthreads();
    if (lds[N + 32]) return;                        // a sweeping wave of this workgroup gave up (every other workgroup times out by itself)
    // "compute": checksum over all N values (stands in for B-operand reads + MFMA + cross-wave combine)
    unsigned part = 0;
    for (int r = 0; r <= extra_work; r++)
      for (int i = tid; i < N; i += NT) part += lds[i] + r;
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    __syncthreads();
    if ((tid & 63) == 0) lds[N + (tid >> 6)] = part;
    __syncthreads();
    sum = 0;
    for (int w = 0; w < NT / 64; w++) sum += lds[N + w];
    __syncthreads();
  }
  if (tid == 0) {
    final_vals[b] = sum;
    cycles[b] = (unsigned long long)(wall_clock64() - t_start);
  }
}

static unsigned host_chain(int N, int steps, int extra_work, int NT) {
  // replay: sum_e = sum over i of (mix(e, i, sum_{e-1}) + r) for r in 0..extra_work  (the per-thread "+ r" adds r once per element)
  unsigned sum = 0;
  for (int e = 1; e <= steps; e++) {
    unsigned s = 0;
    for (int r = 0; r <= extra_work; r++)
      for (int i = 0; i < N; i++) s += mix(e, i, sum) + r;
    sum = s;
  }
  (void)NT;
  return sum;
}

template <int NT, int V = 1>
static void run(int G, int N, int steps, int sweep_waves, int extra_work, hipStream_t st) {
  unsigned long long *gran, *cyc; unsigned *tmo, *fin;
  CK(hipMalloc(&gran, (size_t)2 * N * 8)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&fin, G * 4)); CK(hipMalloc(&cyc, G * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f; bool ok = true; unsigned tm = 0;
  const size_t shm = (size_t)(N + 40) * 4;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipMemsetAsync(gran, 0, (size_t)2 * N * 8, st)); CK(hipMemsetAsync(tmo, 0, 4, st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((k_xchg<NT, V>), dim3(G), dim3(NT), shm, st, gran, N, steps, sweep_waves, extra_work, tmo, fin, cyc);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
    CK(hipMemcpy(&tm, tmo, 4, hipMemcpyDeviceToHost));
    if (tm) { ok = false; break; }
    std::vector<unsigned> f(G); CK(hipMemcpy(f.data(), fin, G * 4, hipMemcpyDeviceToHost));
    const unsigned want = host_chain(N, steps, extra_work, NT);
    for (int b = 0; b < G; b++) if (f[b] != want) ok = false;
  }
  std::vector<unsigned long long> c(G); CK(hipMemcpy(c.data(), cyc, G * 8, hipMemcpyDeviceToHost));
  unsigned long long cmax = 0; for (auto x : c) if (x > cmax) cmax = x;
  printf("V=%d G=%3d NT=%4d N=%5d (%.1f KB granules) sweepwaves=%2d work=%d : %.3f us/step (event, %d steps)  in-kernel %.3f us/step  %s%s\n",
         V, G, NT, N, N * 8 / 1024.0, sweep_waves, extra_work, best * 1e3f / steps, steps, cmax * 0.01 / steps, ok ? "OK" : "MISMATCH",
         tm ? " TIMEOUT" : "");
  fflush(stdout);
  CK(hipFree(gran)); CK(hipFree(tmo)); CK(hipFree(fin)); CK(hipFree(cyc));
}


// 16-byte granules {tag, 3 payload words}: the transport of a bf16 chain (6 bf16 operands behind one 32-bit tag).  NG granules
// in all, one 16-byte sc1 store each, swept with 16-byte sc1 loads by `sweep_waves` waves.
template <int NT>
__global__ __launch_bounds__(NT) void k_xchg16(u32x4 *gran, int NG, int steps, int sweep_waves, unsigned *tmo, unsigned *final_vals,
                                               unsigned long long *cycles) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];   // 3*NG payload words + NT/64 partial sums
  const int tid = threadIdx.x, G = gridDim.x, b = blockIdx.x;
  const int own0 = (int)((long)NG * b / G), own1 = (int)((long)NG * (b + 1) / G);
  const int nsweep = sweep_waves * 64, NW = 3 * NG;
  unsigned sum = 0;
  if (tid == 0) lds[NW + 32] = 0u;
  __syncthreads();
  const long long t_start = wall_clock64();
  for (int e = 1; e <= steps; e++) {
    u32x4 *slot = gran + (size_t)(e & 1) * NG;
    for (int i = own0 + tid; i < own1; i += NT) {
      u32x4 v = {(unsigned)e, mix(e, 3 * i, sum), mix(e, 3 * i + 1, sum), mix(e, 3 * i + 2, sum)};
      const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void *)slot, 0, NG * 16, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(v, ws, i * 16, 0, 16);   // one 16-byte sc1 store
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)slot, 0, NG * 16, 0x00020000);
    if (tid < nsweep) {
      bool done = false;
      for (unsigned spins = 0; !done; spins++) {
        bool ok = true;
        constexpr int MAXL = 8;
        u32x4 q[MAXL];
#pragma unroll
        for (int l = 0; l < MAXL; l++) {
          const int i = tid + l * nsweep;
          if (i < NG) q[l] = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16);
        }
#pragma unroll
        for (int l = 0; l < MAXL; l++) {
          const int i = tid + l * nsweep;
          if (i < NG) { ok &= q[l].x == (unsigned)e; lds[3 * i] = q[l].y; lds[3 * i + 1] = q[l].z; lds[3 * i + 2] = q[l].w; }
        }
        done = __all(ok);
        if (!done && (spins & 63) == 63 && wall_clock64() - t_start > 100000000LL / 20) {
          if ((tid & 63) == 0) { atomicExch(tmo, 0x80000000u | (unsigned)e); lds[NW + 32] = 1u; }
          done = true;
        }
      }
    }
    __syncthreads();
    if (lds[NW + 32]) return;
    unsigned part = 0;
    for (int i = tid; i < NW; i += NT) part += lds[i];
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    __syncthreads();
    if ((tid & 63) == 0) lds[NW + (tid >> 6)] = part;
    __syncthreads();
    sum = 0;
    for (int w = 0; w < NT / 64; w++) sum += lds[NW + w];
    __syncthreads();
  }
  if (tid == 0) { final_vals[b] = sum; cycles[b] = (unsigned long long)(wall_clock64() - t_start); }
}

template <int NT>
static void run16(int G, int NG, int steps, int sweep_waves, hipStream_t st) {
  u32x4 *gran; unsigned long long *cyc; unsigned *tmo, *fin;
  CK(hipMalloc(&gran, (size_t)2 * NG * 16)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&fin, G * 4)); CK(hipMalloc(&cyc, G * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f; bool ok = true; unsigned tm = 0;
  const size_t shm = (size_t)(3 * NG + 40) * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xchg16<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  for (int rep = 0; rep < 4; rep++) {
    CK(hipMemsetAsync(gran, 0, (size_t)2 * NG * 16, st)); CK(hipMemsetAsync(tmo, 0, 4, st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((k_xchg16<NT>), dim3(G), dim3(NT), shm, st, gran, NG, steps, sweep_waves, tmo, fin, cyc);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
    CK(hipMemcpy(&tm, tmo, 4, hipMemcpyDeviceToHost));
    if (tm) { ok = false; break; }
    std::vector<unsigned> f(G); CK(hipMemcpy(f.data(), fin, G * 4, hipMemcpyDeviceToHost));
    unsigned sum = 0;
    for (int e = 1; e <= steps; e++) { unsigned s2 = 0; for (int i = 0; i < 3 * NG; i++) s2 += mix(e, i, sum); sum = s2; }
    for (int b = 0; b < G; b++) if (f[b] != sum) ok = false;
  }
  printf("16-byte granules G=%3d NT=%4d NG=%5d (%.1f KB, %d bf16 values) sweepwaves=%2d : %.3f us/step (event, %d steps)  %s%s\n",
         G, NT, NG, NG * 16 / 1024.0, 6 * NG, sweep_waves, best * 1e3f / steps, steps, ok ? "OK" : "MISMATCH", tm ? " TIMEOUT" : "");
  fflush(stdout);
  CK(hipFree(gran)); CK(hipFree(tmo)); CK(hipFree(fin)); CK(hipFree(cyc));
}

int main(int argc, char **argv) {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int steps = argc > 1 ? atoi(argv[1]) : 200;
  if (argc > 2 && !strcmp(argv[2], "wide")) {          // the exchange of a 16..32-stream bf16 chain (C = 1024 cells)
    for (int S : {8, 16, 32}) {
      const int NG = (1024 * S + 5) / 6;
      for (int G : {128, 256}) {
        if (NG <= 8 * 512) run16<512>(G, NG, steps, 8, st);
        if (NG <= 8 * 1024) run16<1024>(G, NG, steps, 16, st);
        if (NG <= 8 * 768) run16<768>(G, NG, steps, 12, st);
      }
    }
    return 0;
  }
  for (int N : {2048, 3200, 6400, 12800}) {
    for (int G : {50, 100, 200, 256}) {
      if (N <= 16 * 256) run<256>(G, N, steps, 4, 0, st);
      if (N <= 16 * 512) run<512>(G, N, steps, 8, 0, st);
      run<1024>(G, N, steps, 16, 0, st);
      if (N <= 16 * 1024) run<1024>(G, N, steps, 8, 0, st);
      if (N <= 32 * 512) run<512, 2>(G, N, steps, 8, 0, st);
      run<1024, 2>(G, N, steps, 16, 0, st);
    }
  }
  // heavier per-step work between exchanges (arrival skew)
  for (int G : {100, 200}) run<256>(G, 3200, steps, 4, 8, st);
  return 0;
}
