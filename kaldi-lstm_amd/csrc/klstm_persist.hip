// kaldi-lstm_amd/csrc/klstm_persist.hip -- weights-RESIDENT recurrence chain for NumStream <= 4 (engine option "persist").
//
// The launch-per-step chain (klstm_kernels.hip) re-fetches its whole weight operand (~10.5 MB at 40/800/512) in every one
// of the 2T step kernels because nothing on-chip survives a kernel boundary: 12x the algorithmic HBM traffic of a
// minibatch.  Here ONE kernel per direction runs all steps of the folded recurrence
//     forward   a(t)   = W_x x(t) + b + W_rm m(t-1)            (...streams.h:275 with r(t-1) = W_r_m m(t-1), :312)
//     backward  d_m(t) = P(t) + dgifo(t+1) W_rm                  (:391 substituted into :408),  P = out_diff W_r_m
// with each workgroup's slice of the packed operand ([W_rm | W_x] rows of its cells / W_rm^T rows of its cells) held in
// VGPRs for the whole minibatch, and the per-step all-to-all (every workgroup needs all of m(t-1) / d_m(t+1):
// S x C floats, 12.8 KB at 4 x 800) done INSIDE the launch:
//   * transport = data-tagged 8-byte granules {tag, fp32 value}, one sc1 (write-through, agent-scope relaxed atomic)
//     store per (cell, stream) by the owning lane, swept with 16-byte sc1 buffer loads by every workgroup until every tag
//     matches (cdna_hip_programming.md Guideline 16 recipe R2: the data IS the flag, no fence, no separate flag;
//     MI355X_MICROARCH.md "allgather" row).  Placement-independent: no dependence on dispatch order or XCD.
//   * two granule slots (parity of t): a workgroup can publish step t+1 only after it has seen ALL of step t, and every
//     workgroup publishes step t only after its sweep of step t-1 has finished, so a slot is never rewritten while
//     somebody still sweeps it.
//   * tags = epoch + t with a device-resident epoch that the last workgroup to finish advances by T + 2: no per-call
//     memset, and a hipGraph replay (frozen kernel arguments) still sees fresh tags.
//   * every spin is bounded (wall clock, ~50 ms): on expiry the workgroup records the step in status[0] and leaves; the
//     engine reports it at the next synchronising call.  All workgroups must be co-resident: grid <= 128 workgroups of
//     512 threads on 256 CUs.
// Backward: only d_m travels.  dgifo(t+1) -- the 4C-wide operand of the contraction -- is recomputed by EVERY workgroup
// for all cells from d_m(t+1), its own replica of the d_c / d_i / d_f carry and the forward planes (L2-resident, requested
// before the sweep): S x C granules per step instead of S x 4C, and the replicas are bit-identical (same instruction
// sequence on the same inputs).  The owner of a cell writes its dgifo / dc rows for the gradient products.
//
// Geometry = the 4x4x1_16b forms of klstm_kernels.hip (same packed operands, written by the fold product):
//   forward : tile = 4 cells x 4 gates (16 rows), chunk = 32 k, lane l feeds A row l&15 / k-group l>>4, B stream l&3
//   backward: tile = 4 cells (4 rows), chunk = 128 k, block b = k-group, A lane 4b+i = row i, B lane 4b+j = stream j
// A workgroup owns TPW tiles; its 8 waves are split TPW ways, the waves of a tile split K.
#include "klstm_kernels.h"
#include "klstm_math.h"

#include <hip/hip_ext.h>

namespace klstm {

#pragma clang fp contract(off)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int PCMAX = 1024;            // largest cell dim: cells per thread in the sweep / elementwise passes = PCMAX / threads
constexpr long long SPIN_LIMIT = 5000000;   // wall_clock64 ticks (100 MHz): 50 ms

struct PersistFwdArgs {
  int C, I, S, T;
  int nchm, nch;                  // 32-wide chunks over C (the m part) and in total (m + x)
  const float4 *wpk;              // packed [W_rm | W_x], gates order: [C/4 tiles][nch][2][64]
  const float *bias, *pi, *pf, *po;
  float *gifo, *cc, *hh, *mm;     // activation planes, time-major row blocks of S
  const float *x; int x_stride;   // input rows [T*S x I]
  float *c_save;                  // prev_c [S x C]
  unsigned long long *gran;       // [2][C*4] granules, cell-major (4 stream slots per cell)
  unsigned *ctrl;                 // [0] epoch, [1] finished workgroups, [2] status (0 = ok)
};

struct PersistBwdArgs {
  int C, S, T;
  int nch;                        // 128-wide chunks over 4C
  const float4 *wpk;              // packed W_rm^T, 4-row geometry: [C/4 tiles][nch][2][64]
  const float *pi, *pf, *po;
  const float *gifo, *cc, *hh;
  float *dgifo, *dc;
  const float *P;                 // out_diff * W_r_m [T*S x C]
  unsigned long long *gran;
  unsigned *ctrl;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gran_rsrc(const unsigned long long *p, int n) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long *>(p), 0, n * 8, 0x00020000);
}
__device__ __forceinline__ void publish(unsigned long long *slot, int idx, unsigned tag, float v) {
  __hip_atomic_store(slot + idx, ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);                     // one 8-byte sc1 store: tag and value cannot tear
}

// Sweep the 4 granules of each of this thread's cells until every tag of a live stream equals `tag`; returns false on
// timeout.  Two 16-byte sc1 loads per cell, all in flight before the first check.
template <int PCELL>
__device__ __forceinline__ bool sweep_cells(const unsigned long long *slot, int C, int S, unsigned tag, const int (&cell)[PCELL],
                                            float (&v)[PCELL][4], long long t_start) {
  const __amdgpu_buffer_rsrc_t rs = gran_rsrc(slot, C * 4);
  for (unsigned spins = 0;; spins++) {
    u32x4 q[PCELL][2];
#pragma unroll
    for (int j = 0; j < PCELL; j++)
      if (cell[j] < C) {
        q[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, cell[j] * 32, 0, 16);        // aux 16 = sc1
        q[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, cell[j] * 32 + 16, 0, 16);
      }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < PCELL; j++)
      if (cell[j] < C) {
        ok &= q[j][0].y == tag && (S < 2 || q[j][0].w == tag) && (S < 3 || q[j][1].y == tag) && (S < 4 || q[j][1].w == tag);
        v[j][0] = __builtin_bit_cast(float, q[j][0].x); v[j][1] = __builtin_bit_cast(float, q[j][0].z);
        v[j][2] = __builtin_bit_cast(float, q[j][1].x); v[j][3] = __builtin_bit_cast(float, q[j][1].z);
      }
    if (ok) return true;
    if ((spins & 31) == 31 && wall_clock64() - t_start > SPIN_LIMIT) return false;
  }
}

// end of launch: the last workgroup to arrive advances the epoch for the next call (a later launch cannot start before
// every workgroup of this one has exited, so nobody reads ctrl[0] concurrently)
__device__ __forceinline__ void finish(unsigned *ctrl, unsigned epoch, int T) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = atomicAdd(&ctrl[1], 1u);
    if (old == gridDim.x - 1) {
      __hip_atomic_store(&ctrl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctrl[0], epoch + (unsigned)T + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------
// forward: steps 2..T (step 1 closes over the carried r under possibly older weights and stays with k_gates_v)
// -------------------------------------------------------------------------------------------------------------------
template <int TPW, int MAXC, int PNW>
__global__ __launch_bounds__(PNW * 64) void k_fwd_persist(PersistFwdArgs a) {
  constexpr int WPT = PNW / TPW;                     // waves per tile (K split)
  constexpr int PNT = PNW * 64, PCELL = PCMAX / PNT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int C = a.C, S = a.S, T = a.T, I = a.I, nch = a.nch;
  const int LDB = nch * KCH + 16;                    // ds_read_b128 of the B operand conflict-free (klstm_kernels.hip VGeo)
  float *ldsB = lds;                                 // [4][LDB]: row s = [ m(t-1)[s][0..C) | pad | x(t)[s][0..I) | pad ]
  f32x4 *red = reinterpret_cast<f32x4 *>(lds + 4 * LDB);      // [PNW][16]
  unsigned *abortf = reinterpret_cast<unsigned *>(red + PNW * 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = wave / WPT, kw = wave % WPT;
  const int tile = blockIdx.x * TPW + tl;
  const int bs = lane & 3, kg = lane >> 4, q = (lane >> 2) & 3;
  const long long t_start = wall_clock64();
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- resident weights: this wave's chunks of its tile ----
  float4 a0[MAXC], a1[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; i++) {
    const int ch = kw + i * WPT;
    const float4 *ap = a.wpk + ((size_t)tile * nch + (ch < nch ? ch : 0)) * 128 + lane;
    a0[i] = ap[0]; a1[i] = ap[64];
  }
  // ---- owner lanes: wave kw == 0 of a tile, lanes 0..15 = (cell 4*tile + q, stream bs) ----
  const int e_cell = tile * 4 + q;
  const bool e_on = kw == 0 && lane < 16 && bs < S && e_cell < C;
  const int lc = e_on ? e_cell : 0, ls = e_on ? bs : 0;
  float pre[4];
#pragma unroll
  for (int g = 0; g < 4; g++) pre[g] = a.bias[g * C + lc];
  const float wpi = a.pi[lc], wpf = a.pf[lc], wpo = a.po[lc];
  float cp = a.cc[((size_t)1 * S + ls) * C + lc];                    // c(1), written by the step-1 kernel
  // ---- this thread's cells in the sweep ----
  int cell[PCELL];
#pragma unroll
  for (int j = 0; j < PCELL; j++) cell[j] = tid + j * PNT;
  // zero the B slab once: pad columns and rows of absent streams stay zero for the whole launch
  for (int i = tid; i < 4 * LDB; i += PNT) ldsB[i] = 0.f;
  if (tid == 0) *abortf = 0u;
  __syncthreads();

  const int nx4 = I / 4;                             // float4 per x row
  for (int t = 2; t <= T; t++) {
    // x(t): requested before the sweep, stored after it
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool x_on = tid < S * nx4;
    const int xs = x_on ? tid / nx4 : 0, xk = x_on ? (tid % nx4) * 4 : 0;
    if (x_on) xv = *reinterpret_cast<const float4 *>(a.x + ((size_t)(t - 1) * S + xs) * a.x_stride + xk);
    // m(t-1) of every cell
    float mv[PCELL][4];
    if (t == 2) {
#pragma unroll
      for (int j = 0; j < PCELL; j++)
#pragma unroll
        for (int s = 0; s < 4; s++) mv[j][s] = (cell[j] < C && s < S) ? a.mm[((size_t)1 * S + s) * C + cell[j]] : 0.f;
    } else {
      if (!sweep_cells(a.gran + (size_t)((t - 1) & 1) * C * 4, C, S, epoch + (unsigned)(t - 1), cell, mv, t_start)) {
        *abortf = 1u;
        if (lane == 0) atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t);
      }
    }
#pragma unroll
    for (int j = 0; j < PCELL; j++)
      if (cell[j] < C) {
#pragma unroll
        for (int s = 0; s < 4; s++) if (s < S) ldsB[s * LDB + cell[j]] = mv[j][s];
      }
    if (x_on) *reinterpret_cast<float4 *>(ldsB + xs * LDB + a.nchm * KCH + xk) = xv;
    __syncthreads();
    if (*reinterpret_cast<volatile unsigned *>(abortf)) break;
    // ---- contraction over [m(t-1) | x(t)] with the resident weights ----
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MAXC; i++) {
      const int ch = kw + i * WPT;
      if (ch < nch) {
        const float *bp = ldsB + bs * LDB + ch * KCH + kg * 8;
        const float4 b0 = *reinterpret_cast<const float4 *>(bp), b1 = *reinterpret_cast<const float4 *>(bp + 4);
        const float av[8] = {a0[i].x, a0[i].y, a0[i].z, a0[i].w, a1[i].x, a1[i].y, a1[i].z, a1[i].w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], bv[j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j + 1], bv[j + 1], acc1, 0, 0, 0);
        }
      }
    }
    f32x4 v = acc0 + acc1;
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) {              // the 4 k-groups of a (row, stream) pair sit 16 lanes apart
      v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
    }
    if (lane < 16) red[wave * 16 + lane] = v;
    __syncthreads();
    if (e_on) {
      f32x4 s4 = red[(tl * WPT) * 16 + lane];
#pragma unroll
      for (int w = 1; w < WPT; w++) s4 += red[(tl * WPT + w) * 16 + lane];
      const size_t e_row = (size_t)t * S + bs;
      float ag = s4.x + pre[0];
      float ai = s4.y + pre[1];
      float af = s4.z + pre[2];
      float ao = s4.w + pre[3];
      ai += wpi * cp;                                // :278
      af += wpf * cp;                                // :281
      const float gi = k_sigmoid(ai), gf = k_sigmoid(af), gg = k_tanh(ag);   // :284-288
      float c = gg * gi;                             // :291
      c = c + cp * gf;                               // :294
      c = c < -50.f ? -50.f : c;                     // :296
      c = c > 50.f ? 50.f : c;                       // :297
      const float h = k_tanh(c);                     // :300
      ao += wpo * c;                                 // :303
      const float go = k_sigmoid(ao);                // :306
      const float m = h * go;                        // :309
      if (t < T) publish(a.gran + (size_t)(t & 1) * C * 4, e_cell * 4 + bs, epoch + (unsigned)t, m);
      float *gp = a.gifo + e_row * 4 * C + e_cell;
      gp[0] = gg; gp[C] = gi; gp[2 * C] = gf; gp[3 * C] = go;
      a.cc[e_row * C + e_cell] = c;
      a.hh[e_row * C + e_cell] = h;
      a.mm[e_row * C + e_cell] = m;
      if (t == T) a.c_save[(size_t)bs * C + e_cell] = c;       // :331 (c columns)
      cp = c;
    }
  }
  finish(a.ctrl, epoch, T);
}

// -------------------------------------------------------------------------------------------------------------------
// backward: steps T..1
// -------------------------------------------------------------------------------------------------------------------
template <int TPW, int MAXC, int PNW>
__global__ __launch_bounds__(PNW * 64) void k_bwd_persist(PersistBwdArgs a) {
  constexpr int WPT = PNW / TPW;
  constexpr int PNT = PNW * 64, PCELL = PCMAX / PNT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int C = a.C, S = a.S, T = a.T, nch = a.nch, K = 4 * a.C;
  const int LDD = nch * 128 + 16;                    // (LDD mod 64 == 16: the 16-lane groups of ds_read_b128 hit 16 distinct slots)
  float *ldsD = lds;                                 // [4][LDD]: dgifo(t+1) rows, natural g|i|f|o order
  f32x4 *red = reinterpret_cast<f32x4 *>(lds + 4 * LDD);      // [PNW][4]
  unsigned *abortf = reinterpret_cast<unsigned *>(red + PNW * 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = wave / WPT, kw = wave % WPT;
  const int tile = blockIdx.x * TPW + tl;
  const int kg = lane >> 2, bj = lane & 3;
  const long long t_start = wall_clock64();
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  float4 a0[MAXC], a1[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; i++) {
    const int ch = kw + i * WPT;
    const float4 *ap = a.wpk + ((size_t)tile * nch + (ch < nch ? ch : 0)) * 128 + lane;
    a0[i] = ap[0]; a1[i] = ap[64];
  }
  // owner lanes of the d_m epilogue: wave kw == 0 of a tile, lanes 0..15 = (cell 4*tile + lane/4, stream lane%4)
  const int e_i = (lane >> 2) & 3, e_j = lane & 3;
  const int e_cell = tile * 4 + e_i;
  const bool e_on = kw == 0 && lane < 16 && e_j < S && e_cell < C;
  // this thread's cells in the elementwise pass (all cells, every workgroup); `mine`: this workgroup writes their planes
  int cell[PCELL];
  bool mine[PCELL];
  float wpi[PCELL], wpf[PCELL], wpo[PCELL];
  float dcn[PCELL][4], din[PCELL][4], dfn[PCELL][4];
#pragma unroll
  for (int j = 0; j < PCELL; j++) {
    cell[j] = tid + j * PNT;
    const int lc = cell[j] < C ? cell[j] : 0;
    mine[j] = cell[j] < C && (cell[j] >> 2) / TPW == (int)blockIdx.x;
    wpi[j] = a.pi[lc]; wpf[j] = a.pf[lc]; wpo[j] = a.po[lc];
#pragma unroll
    for (int s = 0; s < 4; s++) { dcn[j][s] = 0.f; din[j][s] = 0.f; dfn[j][s] = 0.f; }
    if (mine[j]) {                                   // the batched d_r product reads dgifo(T+1) as operand rows: keep them zero (:351)
      for (int s = 0; s < S; s++) {
        float *zp = a.dgifo + ((size_t)(T + 1) * S + s) * K + cell[j];
        zp[0] = 0.f; zp[C] = 0.f; zp[2 * C] = 0.f; zp[3 * C] = 0.f;
      }
    }
  }
  for (int i = tid; i < 4 * LDD; i += PNT) ldsD[i] = 0.f;
  if (tid == 0) *abortf = 0u;
  __syncthreads();

  // forward planes through buffer descriptors: one 32-bit lane offset per (cell, stream), the frame / gate part of the
  // address in the scalar offset (64-bit per-load addresses cost two VGPRs each and pushed this kernel into scratch)
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gifo), 0, (T + 2) * S * K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.hh), 0, (T + 2) * S * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.cc), 0, (T + 2) * S * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.P), 0, T * S * C * 4, 0x00020000);
  int offg[PCELL][4], offc[PCELL][4];
  float fn[PCELL][4];                                // f(t+1) = the yf this thread loaded one iteration earlier
#pragma unroll
  for (int j = 0; j < PCELL; j++)
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const bool on = cell[j] < C && s < S;
      offg[j][s] = on ? (s * K + cell[j]) * 4 : 0;
      offc[j][s] = on ? (s * C + cell[j]) * 4 : 0;
      fn[j][s] = 0.f;
    }

  for (int t = T; t >= 1; t--) {
    // forward planes of frame t for all cells: requested before the sweep (L2-resident, every workgroup reads the same rows)
    float yg[PCELL][4], yi[PCELL][4], yf[PCELL][4], yo[PCELL][4], yh[PCELL][4], cpv[PCELL][4], dm[PCELL][4];
    const int sg = t * S * K * 4, sc = t * S * C * 4;
#pragma unroll
    for (int j = 0; j < PCELL; j++)
#pragma unroll
      for (int s = 0; s < 4; s++) {
        yg[j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, offg[j][s], sg, 0));
        yi[j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, offg[j][s], sg + C * 4, 0));
        yf[j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, offg[j][s], sg + 2 * C * 4, 0));
        yo[j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, offg[j][s], sg + 3 * C * 4, 0));
        yh[j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_h, offc[j][s], sc, 0));
        cpv[j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_c, offc[j][s], sc - S * C * 4, 0));
        if (t == T) dm[j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_p, offc[j][s], (T - 1) * S * C * 4, 0));   // d_m(T) = P(T): dgifo(T+1) = 0
      }
    if (t < T) {
      if (!sweep_cells(a.gran + (size_t)(t & 1) * C * 4, C, S, epoch + (unsigned)t, cell, dm, t_start)) {
        *abortf = 1u;
        if (lane == 0) atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t);
      }
    }
    // elementwise BPTT of frame t for all cells (:411-440), replicated in every workgroup
#pragma unroll
    for (int j = 0; j < PCELL; j++)
      if (cell[j] < C) {
#pragma unroll
        for (int s = 0; s < 4; s++)
          if (s < S) {
            const float d_h = k_diff_tanh(dm[j][s] * yo[j][s], yh[j][s]);        // :411-412
            const float d_o = k_diff_sigmoid(dm[j][s] * yh[j][s], yo[j][s]);     // :415-416
            float d_c = d_h;                                                     // :424
            d_c = d_c + dcn[j][s] * fn[j][s];                                    // :425
            d_c = d_c + wpi[j] * din[j][s];                                      // :426
            d_c = d_c + wpf[j] * dfn[j][s];                                      // :427
            d_c = d_c + wpo[j] * d_o;                                            // :428
            const float o_f = k_diff_sigmoid(d_c * cpv[j][s], yf[j][s]);         // :431-432
            const float o_i = k_diff_sigmoid(d_c * yg[j][s], yi[j][s]);          // :435-436
            const float o_g = k_diff_tanh(d_c * yi[j][s], yg[j][s]);             // :439-440
            dcn[j][s] = d_c; din[j][s] = o_i; dfn[j][s] = o_f; fn[j][s] = yf[j][s];
            if (t > 1) {
              float *lp = ldsD + s * LDD + cell[j];
              lp[0] = o_g; lp[C] = o_i; lp[2 * C] = o_f; lp[3 * C] = d_o;
            }
            if (mine[j]) {
              const size_t row = (size_t)t * S + s;
              float *dp = a.dgifo + row * K + cell[j];
              dp[0] = o_g; dp[C] = o_i; dp[2 * C] = o_f; dp[3 * C] = d_o;
              a.dc[row * C + cell[j]] = d_c;
            }
          }
      }
    if (t == 1) break;
    __syncthreads();
    if (*reinterpret_cast<volatile unsigned *>(abortf)) break;
    // ---- d_m(t-1) rows of this tile: contraction of dgifo(t) over K = 4C with the resident W_rm^T rows ----
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MAXC; i++) {
      const int ch = kw + i * WPT;
      if (ch < nch) {
        const float *bp = ldsD + bj * LDD + ch * 128 + kg * 4;
        const float4 b0 = *reinterpret_cast<const float4 *>(bp), b1 = *reinterpret_cast<const float4 *>(bp + 64);
        const float av[8] = {a0[i].x, a0[i].y, a0[i].z, a0[i].w, a1[i].x, a1[i].y, a1[i].z, a1[i].w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], bv[j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j + 1], bv[j + 1], acc1, 0, 0, 0);
        }
      }
    }
    f32x4 v = acc0 + acc1;
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) {               // the 16 k-groups of a (row, stream) pair: lanes with equal lane&3
      v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
    }
    if (lane < 4) red[wave * 4 + lane] = v;
    __syncthreads();
    if (e_on) {
      // row (cell) e_i of stream e_j: component e_i of red[w][e_j]
      const float *rp = reinterpret_cast<const float *>(red) + ((tl * WPT) * 4 + e_j) * 4 + e_i;
      float sum = rp[0];
#pragma unroll
      for (int w = 1; w < WPT; w++) sum += rp[w * 16];
      const float dmv = sum + a.P[((size_t)(t - 2) * S + e_j) * C + e_cell];     // frame t-1 is row block t-2 of P  (:408 with :391)
      publish(a.gran + (size_t)((t - 1) & 1) * C * 4, e_cell * 4 + e_j, epoch + (unsigned)(t - 1), dmv);
    }
  }
  finish(a.ctrl, epoch, T);
}

// -------------------------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------------------------
static inline int pcdiv(int a, int b) { return (a + b - 1) / b; }

static int g_persist_tpw = 0;       // A-B knobs: tiles per workgroup / waves per workgroup (0 = automatic)
static int g_persist_waves = 0;
void set_persist_tpw(int v) { g_persist_tpw = v; }
void set_persist_waves(int v) { g_persist_waves = v; }

// Geometry: waves per workgroup (8 or 16), tiles (of 4 cells) per workgroup, chunk slots per wave.  Fewer, fatter
// workgroups mean fewer sweepers per exchange (less fabric contention); a wave must hold its chunks in registers.
struct PGeo { int waves, tpw, maxc; };
static PGeo pick_geo(int C, int nch) {
  const int waves = g_persist_waves ? g_persist_waves : 16;
  const int prefer[3] = {g_persist_tpw ? g_persist_tpw : 2, 4, 1};
  for (int tpw : prefer) {
    if (tpw > waves || (C / 4) % tpw != 0 || C / 4 / tpw > 200) continue;
    const int mc = pcdiv(nch, waves / tpw);
    if (mc > 8) continue;
    return PGeo{waves, tpw, mc <= 4 ? 4 : 8};
  }
  return PGeo{0, 0, 0};
}

bool persist_supported(const Dims &d) {
  if (d.S > 4 || d.C % 8 != 0 || d.I % 8 != 0 || d.C > PCMAX || d.S * (d.I / 4) > 512) return false;
  const int nf = pcdiv(d.C, KCH) + pcdiv(d.I, KCH), nb = pcdiv(4 * d.C, 128);
  return pick_geo(d.C, nf).tpw > 0 && pick_geo(d.C, nb).tpw > 0;
}
size_t persist_gran_bytes(const Dims &d) { return (size_t)2 * d.C * 4 * sizeof(unsigned long long); }

template <class K, class A>
static hipError_t plaunch(K kern, int grid, int threads, size_t shm, hipStream_t st, LaunchProbe pr, const A &a) {
  if (shm > 64 * 1024)                               // above the default dynamic-LDS limit (cell dim 1024 backward)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (pr.start) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, st, pr.start, pr.stop, 0, a);
  else hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, st, a);
  return hipGetLastError();
}
#define PDISPATCH(KERN, g, grid, shm, st, pr, a)                                                              \
  do {                                                                                                        \
    if (g.waves == 8 && g.tpw == 1 && g.maxc == 4) return plaunch(KERN<1, 4, 8>, grid, 512, shm, st, pr, a);   \
    if (g.waves == 8 && g.tpw == 1 && g.maxc == 8) return plaunch(KERN<1, 8, 8>, grid, 512, shm, st, pr, a);   \
    if (g.waves == 8 && g.tpw == 2 && g.maxc == 4) return plaunch(KERN<2, 4, 8>, grid, 512, shm, st, pr, a);   \
    if (g.waves == 8 && g.tpw == 2 && g.maxc == 8) return plaunch(KERN<2, 8, 8>, grid, 512, shm, st, pr, a);   \
    if (g.waves == 8 && g.tpw == 4 && g.maxc == 4) return plaunch(KERN<4, 4, 8>, grid, 512, shm, st, pr, a);   \
    if (g.waves == 8 && g.tpw == 4 && g.maxc == 8) return plaunch(KERN<4, 8, 8>, grid, 512, shm, st, pr, a);   \
    if (g.waves == 16 && g.tpw == 1 && g.maxc == 4) return plaunch(KERN<1, 4, 16>, grid, 1024, shm, st, pr, a); \
    if (g.waves == 16 && g.tpw == 1 && g.maxc == 8) return plaunch(KERN<1, 8, 16>, grid, 1024, shm, st, pr, a); \
    if (g.waves == 16 && g.tpw == 2 && g.maxc == 4) return plaunch(KERN<2, 4, 16>, grid, 1024, shm, st, pr, a); \
    if (g.waves == 16 && g.tpw == 2 && g.maxc == 8) return plaunch(KERN<2, 8, 16>, grid, 1024, shm, st, pr, a); \
    if (g.waves == 16 && g.tpw == 4 && g.maxc == 4) return plaunch(KERN<4, 4, 16>, grid, 1024, shm, st, pr, a); \
    if (g.waves == 16 && g.tpw == 4 && g.maxc == 8) return plaunch(KERN<4, 8, 16>, grid, 1024, shm, st, pr, a); \
    return hipErrorInvalidValue;                                                                              \
  } while (0)

hipError_t launch_fwd_persist(const Dims &d, const FwdPtrs &p, const float *in, int in_stride, unsigned long long *gran,
                              unsigned *ctrl, hipStream_t st, LaunchProbe pr) {
  PersistFwdArgs a;
  a.C = d.C; a.I = d.I; a.S = d.S; a.T = d.T;
  a.nchm = pcdiv(d.C, KCH); a.nch = a.nchm + pcdiv(d.I, KCH);
  a.wpk = p.pk_fold; a.bias = p.bias; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.mm = p.mm;
  a.x = in; a.x_stride = in_stride; a.c_save = p.prev_c; a.gran = gran; a.ctrl = ctrl;
  const PGeo g = pick_geo(d.C, a.nch);
  if (!g.tpw || !p.pk_fold || (reinterpret_cast<uintptr_t>(in) & 15) || in_stride % 4 != 0) return hipErrorInvalidValue;
  const size_t shm = (size_t)(4 * (a.nch * KCH + 16) + g.waves * 16 * 4 + 4) * sizeof(float);
  const int grid = d.C / 4 / g.tpw;
  PDISPATCH(k_fwd_persist, g, grid, shm, st, pr, a);
}

hipError_t launch_bwd_persist(const Dims &d, const BwdPtrs &p, const float *P, unsigned long long *gran, unsigned *ctrl,
                              hipStream_t st, LaunchProbe pr) {
  PersistBwdArgs a;
  a.C = d.C; a.S = d.S; a.T = d.T;
  a.nch = pcdiv(4 * d.C, 128);
  a.wpk = p.pk_fold; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.dgifo = p.dgifo; a.dc = p.dc; a.P = P; a.gran = gran; a.ctrl = ctrl;
  const PGeo g = pick_geo(d.C, a.nch);
  if (!g.tpw || !p.pk_fold) return hipErrorInvalidValue;
  const size_t shm = (size_t)(4 * (a.nch * 128 + 16) + g.waves * 4 * 4 + 4) * sizeof(float);
  const int grid = d.C / 4 / g.tpw;
  PDISPATCH(k_bwd_persist, g, grid, shm, st, pr, a);
}

}  // namespace klstm
