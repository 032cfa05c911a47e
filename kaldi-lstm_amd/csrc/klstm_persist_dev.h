// kaldi-lstm_amd/csrc/klstm_persist_dev.h -- device helpers shared by the two persistent (weights-resident) chain kernels,
// klstm_persist.hip (forward) and klstm_persist_bwd.hip (backward): the granule transport, buffer-descriptor plane I/O,
// the k-group sums of the 4-row MFMA geometry, bounded waits and the end-of-launch epoch hand-over; the reduction of the tail
// workgroups' partial rows (k_tail_reduce, and the first workgroups of k_grads).
#pragma once
#include <hip/hip_runtime.h>
#include "klstm_math.h"

namespace klstm {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr long long SPIN_LIMIT_DEFAULT = 5000000;   // wall_clock64 ticks (100 MHz): 50 ms per wait

#ifdef KLSTM_PERSIST_TIMING
#define PT_N 10
#define PT_DECL() long long pt_prev = clock64(), pt_acc[PT_N] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PT_MARK(i) do { const long long pt_now = clock64(); pt_acc[i] += pt_now - pt_prev; pt_prev = pt_now; } while (0)
#define PT_FLUSH(base) do { if (lane == 0) for (int i_ = 0; i_ < PT_N; i_++) a.dbg[((size_t)blockIdx.x * 16 + wave) * PT_N + i_] = pt_acc[i_]; } while (0)
#else
#define PT_DECL() do {} while (0)
#define PT_MARK(i) do {} while (0)
#define PT_FLUSH(base) do {} while (0)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ void buf_store_f32(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 0);
}
__device__ __forceinline__ void publish(unsigned long long *slot, int idx, unsigned tag, float v) {
  __hip_atomic_store(slot + idx, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);                     // one 8-byte sc1 store: tag and value cannot tear
}

// Sum over the 16 k-groups of the 4-row geometry (lanes with equal lane & 3): two DPP row shifts inside each row of 16 lanes
// (lanes 12..15 then hold their row's sums), then the four rows.  The totals of streams 0..3 end up in lanes 12..15 (of
// every row): those are the epilogue lanes.
//   kgroup_sum    : rows combined by two ds_bpermute rounds (an LDS round trip each)
//   kgroup_sum_pl : rows combined by v_permlane16_swap / v_permlane32_swap (gfx950; register-only): with x = y = c,
//                   permlane16_swap leaves x = [c.r0, c.r0, c.r2, c.r2], y = [c.r1, c.r1, c.r3, c.r3] (odd rows of x swapped with
//                   even rows of y), permlane32_swap x = [s.lo, s.lo], y = [s.hi, s.hi]
// Both are fixed instruction sequences: deterministic, identical in every wave (the two orders differ in the last bits).
__device__ __forceinline__ f32x4 kgroup_sum(f32x4 v) {
  // (scalar copies: __builtin_bit_cast applied directly to a vector-element expression reads element 0)
  float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; e++) {
    c[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c[e]), 0x114, 0xf, 0xf, true));   // row_shr:4: lane i += lane i-4
    c[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c[e]), 0x118, 0xf, 0xf, true));   // row_shr:8: lanes 12..15 = row sums
  }
#pragma unroll
  for (int m = 16; m < 64; m <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; e++) c[e] += __shfl_xor(c[e], m);
  }
  return f32x4{c[0], c[1], c[2], c[3]};
}
__device__ __forceinline__ f32x4 kgroup_sum_pl(f32x4 v) {
  float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; e++) {
    c[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c[e]), 0x114, 0xf, 0xf, true));
    c[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c[e]), 0x118, 0xf, 0xf, true));
  }
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const unsigned u = __float_as_uint(c[e]);
    const u32x2 r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);           // rows 0,1: r0 + r1; rows 2,3: r2 + r3
    const unsigned us = __float_as_uint(s);
    const u32x2 r32 = __builtin_amdgcn_permlane32_swap(us, us, false, false);
    c[e] = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);                    // every row: (r0 + r1) + (r2 + r3)
  }
  return f32x4{c[0], c[1], c[2], c[3]};
}

// d_r / in_diff from the tail workgroups' partial rows: output idx = (frame row r = (t - 1) S + s, column quad cq), 8 lanes per output --
// lane l adds slots l, l + 8, l + 16, ... in that order, then three butterfly stages: ONE fixed tree per output whoever runs it
// (deterministic).  out_diff is added to the d_r columns (:391), d_r(T) = out_diff(T) (:351).
// Every load of an output -- the guard words, up to four partial rows per lane, the out_diff piece -- is requested before the first
// use (one memory round trip per output instead of three in a row: the launch is nothing but latency, 4.3 -> ... us); a raised guard only
// withholds the stores (the partial rows of a launch that gave up are garbage, reading them is harmless).
// WT: the d_r rows leave as 16-byte write-through (sc1) stores -- a consumer in the SAME launch (the W_r_m gradient tiles of k_grads, which
// read them with sc1 loads behind an arrival counter) finds them beyond its own XCD's L2; in_diff has no reader inside the launch.
template <bool WT>
__device__ __forceinline__ void tail_reduce_outputs(const float *tws, int nslots, int T, int S, int R, int ncols, const float *od, int od_stride,
                                                    float *dr, float *in_diff, int id_stride, int first, int step, int l,
                                                    const unsigned *guard = nullptr) {
  const int nqc = ncols >> 2, nout = T * S * nqc;
  const size_t stride = (size_t)T * S * ncols;
  unsigned bad = 0u;
  if (guard) bad = __hip_atomic_load(guard + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | __hip_atomic_load(guard + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int idx = first;; idx += step) {              // (idx grows with the lane: a wave leaves when its first output is past the end -- the butterflies need every lane)
    const bool on = idx < nout;
    if (!__any(on)) break;
    const int r = on ? idx / nqc : 0, cq = on ? idx - r * nqc : 0;
    const int t = r / S + 1, s = r - (t - 1) * S;
    const float *p = tws + (size_t)r * ncols + 4 * cq;
    const bool is_r = 4 * cq < R, want_od = on && l == 0 && is_r && t >= 2;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f), oT = o;
    if (want_od) o = *reinterpret_cast<const float4 *>(od + (size_t)((t - 2) * S + s) * od_stride + 4 * cq);
    if (want_od && t == T) oT = *reinterpret_cast<const float4 *>(od + (size_t)((T - 1) * S + s) * od_stride + 4 * cq);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nslots <= 32) {                              // (C <= 1024: at most four slots per lane, all in flight at once; same order of additions)
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int j = l + 8 * q;
        v[q] = *reinterpret_cast<const float4 *>(p + (size_t)(j < nslots ? j : 0) * stride);
      }
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (l + 8 * q < nslots) { sum.x += v[q].x; sum.y += v[q].y; sum.z += v[q].z; sum.w += v[q].w; }
    } else {
      for (int j = l; j < nslots; j += 8) {
        const float4 v = *reinterpret_cast<const float4 *>(p + j * stride);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      sum.x += __shfl_xor(sum.x, m); sum.y += __shfl_xor(sum.y, m); sum.z += __shfl_xor(sum.z, m); sum.w += __shfl_xor(sum.w, m);
    }
    if (!on || l != 0 || bad) continue;
    if (is_r) {
      if (t < 2) continue;                           // (frame 1 feeds no d_r row; d_r(T) = out_diff(T) goes out with the row of frame T; T >= 3 here)
      const float4 v = make_float4(o.x + sum.x, o.y + sum.y, o.z + sum.z, o.w + sum.w);   // :391
      if constexpr (WT) {
        const __amdgpu_buffer_rsrc_t rs = buf_rsrc(dr, (T + 1) * S * R * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (((t - 1) * S + s) * R + 4 * cq) * 4, 0, 16);   // aux 16 = sc1
        if (t == T) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, oT), rs, ((T * S + s) * R + 4 * cq) * 4, 0, 16);
      } else {
        *reinterpret_cast<float4 *>(dr + ((size_t)(t - 1) * S + s) * R + 4 * cq) = v;
        if (t == T) *reinterpret_cast<float4 *>(dr + ((size_t)T * S + s) * R + 4 * cq) = oT;
      }
    } else if (in_diff) {
      *reinterpret_cast<float4 *>(in_diff + (size_t)((t - 1) * S + s) * id_stride + 4 * cq - R) = sum;   // :457
    }
  }
}


// workgroup barrier that orders LDS traffic only
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Bounded wait on an LDS counter of this workgroup: until *ctr >= target (returns true), the abort flag is up or
// `limit` wall-clock ticks have passed since the wait began (returns false; on expiry the abort flag is raised).
__device__ __forceinline__ bool lds_wait_ge(int *ctr, int target, unsigned *abortf, long long limit) {
  if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) { asm volatile("" ::: "memory"); return true; }
  const long long t0 = wall_clock64();
  for (unsigned spins = 1;; spins++) {
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) break;
    if ((spins & 15) == 0) {
      if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
      if ((spins & 1023) == 0 && wall_clock64() - t0 > limit) {
        __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return false;
      }
    }
  }
  asm volatile("" ::: "memory");                     // (the reads that follow are not hoisted above the poll)
  return true;
}

// end of launch: the last workgroup to arrive advances the epoch for the next call (a later launch cannot start before
// every workgroup of this one has exited, so nobody reads ctrl[0] concurrently)
// count (or null): the engine's launch counter (persistent launches that have run): the LAST workgroup moves it on by one --
// nobody else writes it during a launch, so launch_ordinal() at a give-up (always in front of that workgroup's own arrival
// here) and the last workgroup itself both see the value the launch started with.
__device__ __forceinline__ unsigned launch_ordinal(const unsigned *guard) {
#ifdef KLSTM_NO_LAUNCH_COUNT
  return 0u;
#else
  return guard ? __hip_atomic_load(guard + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : 0u;
#endif
}
// hdone (or null): a HOST-MAPPED word the last workgroup writes when the launch is over: the new launch count in bits 0..30, bit 31 set when
// a status word of the chain is up (this launch or one in front of it gave up).  A host that waits for a persistent launch
// ("persist_verify") spins on this word instead of synchronising the stream: it hears of the end of the launch one PCIe write after
// the last workgroup left, not one completion-signal round trip later.  One word carries both facts: nothing to order.
__device__ __forceinline__ void finish(unsigned *ctrl, unsigned epoch, int ntags, unsigned *count = nullptr, const unsigned *guard = nullptr,
                                       unsigned *hdone = nullptr) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = atomicAdd(&ctrl[1], 1u);
    if (old == gridDim.x - 1) {
      __hip_atomic_store(&ctrl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctrl[0], epoch + (unsigned)ntags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef KLSTM_NO_LAUNCH_COUNT
      unsigned n = 0u;
      if (count) n = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      if (hdone && guard) {
        const unsigned bad = __hip_atomic_load(guard + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                             __hip_atomic_load(guard + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hdone, (n & 0x7fffffffu) | (bad ? 0x80000000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
#endif
    }
  }
}

}  // namespace klstm
