// kaldi-lstm_amd/csrc/klstm_persist.hip -- weights-RESIDENT recurrence chain (engine option "persist"): forward for up to 8
// streams, backward for up to 4; the default chain from 1 to 4 streams.
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
//   * every spin is bounded (wall clock, ~50 ms): on expiry the workgroup records the step in ctrl[2] and leaves; the
//     engine reports it at the next synchronising call.  All workgroups must be co-resident: grid <= 200 workgroups,
//     one per CU.
//   * wave roles: the waves that own cell math, granule stores and plane stores do NOT sweep (loads return in order behind
//     a wave's own stores: a sweeping wave with write-through stores in flight would wait for their acknowledgement
//     first); one wave per workgroup carries the products that hang off the chain (r = W_r_m m forward; P, d_r, in_diff
//     and the own plane rows backward); all other waves sweep, PCELL cells per thread.
//   * barriers wait for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): nothing global is ordered by them.
// Backward: only d_m travels.  dgifo(t+1) -- the 4C-wide operand of the contraction -- is recomputed by EVERY workgroup
// for all cells from d_m(t+1), its own replica of the d_c / d_i / d_f carry and the forward planes (L2-resident, requested
// before the sweep): S x C granules per step instead of S x 4C, and the replicas are bit-identical (same instruction
// sequence on the same inputs).  The workgroup that owns a cell writes its dgifo / dc rows for the gradient products.
// Step 1 of the forward pass closes over the CARRIED r (possibly produced under older weights) and is contracted against
// the natural [W_gifo_r | W_gifo_x] rows inside the same launch.
//
// Geometry = the 4-row form of v_mfma_f32_4x4x1_16b (16 blocks = 16 k-groups of one 4 rows x 4 streams tile, chunk = 128 k,
// A lane 4b+i = row i, B lane 4b+j = stream j), on the packed operands the fold product writes:
//   forward : a cell wave holds the 4 gate rows of ONE cell over the whole K = [m | x] (gathered from the 16-row gates operand)
//   backward: 4 K waves per tile of 4 cells, each a quarter of K = 4C of the W_rm^T operand
// A workgroup owns TPW tiles (1 by default: 200 workgroups of 12 waves at C = 800).
#include "klstm_kernels.h"
#include "klstm_math.h"

#include <hip/hip_ext.h>

namespace klstm {

#pragma clang fp contract(off)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr long long SPIN_LIMIT = 5000000;   // wall_clock64 ticks (100 MHz): 50 ms

struct PersistFwdArgs {
  int C, I, R, S, T;
  int nchm, nch;                  // 32-wide chunks over C (the m part) and in total (m + x)
  const float4 *wpk;              // packed W_rm (the m chunks of the folded gates operand): [C/4 tiles][nch][2][64]
  const float *wr, *wx;           // natural W_gifo_r [4C x R], W_gifo_x [4C x I] (step 1, and the x chunks of every step)
  const float *wm;                // natural W_r_m [R x C] (rin)
  int rin;                        // 1: r(t) = W_r_m m(t) (:312) is contracted here too (rr plane, out rows, prev_r); 0: by the caller
  float *out; int out_stride;     // output rows [T*S x R] (:328) (rin)
  const float *bias, *pi, *pf, *po;
  float *gifo, *cc, *hh, *mm, *rr; // activation planes, time-major row blocks of S
  const float *x; int x_stride;   // input rows [T*S x I]
  float *prev_c;                  // carried c [S x C]: read at step 1 (:231), written at step T (:331)
  float *prev_r;                  // carried r [S x R]: read at step 1, written with r(T) (:331) (rin)
  unsigned long long *gran;       // [2][C*4] granules, cell-major (4 stream slots per cell)
  unsigned *ctrl;                 // [0] epoch, [1] finished workgroups, [2] status (0 = ok)
  int nap0, nap;                  // sweepers sleep nap0 x 256 clocks before the first pass of a step, nap x 64 between passes
#ifdef KLSTM_PERSIST_TIMING
  long long *dbg;                 // per workgroup: shader-clock sums of the phases of a step (tools/persist_anatomy.hip)
#endif
};
#ifdef KLSTM_PERSIST_TIMING
#define PT_DECL() long long pt_prev = clock64(), pt_acc[6] = {0, 0, 0, 0, 0, 0}
#define PT_MARK(i) do { const long long pt_now = clock64(); pt_acc[i] += pt_now - pt_prev; pt_prev = pt_now; } while (0)
#define PT_FLUSH(base) do { if (lane == 0) for (int i_ = 0; i_ < 6; i_++) a.dbg[((size_t)blockIdx.x * 16 + wave) * 6 + i_] = pt_acc[i_]; } while (0)
#else
#define PT_DECL() do {} while (0)
#define PT_MARK(i) do {} while (0)
#define PT_FLUSH(base) do {} while (0)
#endif

struct PersistBwdArgs {
  int C, R, S, T;
  int pin;                        // 1: P = out_diff W_r_m is computed here (own columns, kept in LDS); 0: read from P
  const float *od; int od_stride; // out_diff rows [T*S x R]
  const float *wmT;               // W_r_m^T [C x R]
  int din, I;                     // din: d_r(1..T) = out_diff + dgifo(2..T+1) W_gifo_r (:391) and in_diff = dgifo W_gifo_x (:457) are
                                  //      contracted here too, 4 columns per workgroup (bit 1: in_diff wanted)
  const float *wrT, *wxT;         // W_gifo_r^T [R x 4C], W_gifo_x^T [I x 4C]
  float *dr;                      // d_r plane [(T+2)*S x R], time-major row blocks
  float *in_diff; int id_stride;  // [T*S x I]
  int nch;                        // 128-wide chunks over 4C
  const float4 *wpk;              // packed W_rm^T, 4-row geometry: [C/4 tiles][nch][2][64]
  const float *pi, *pf, *po;
  const float *gifo, *cc, *hh;
  float *dgifo, *dc;
  const float *P;                 // out_diff * W_r_m [T*S x C]
  unsigned long long *gran;
  unsigned *ctrl;
  int nap0, nap;
#ifdef KLSTM_PERSIST_TIMING
  long long *dbg;
#endif
};

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
// (lanes 12..15 then hold their row's sums), two bpermute rounds across the four rows.  The totals of streams 0..3 end
// up in lanes 12..15 (of every row): those are the epilogue lanes.
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

// workgroup barrier that orders LDS traffic only
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Sweep the 4 granules of each of this thread's cells until every tag of a live stream equals `tag`; returns false on
// timeout.  Two 16-byte sc1 loads per cell, all in flight before the first check; branch-free inside a pass (threads
// without a cell sweep cell 0 and ignore it).
// A polling wave competes with the cell / owner waves of its own CU for the vector-memory queue (their plane and granule
// stores queue behind its loads: measured 1.2-2.8 us for a 7-store epilogue next to unthrottled pollers), so a sweeper
// sleeps through the part of the step in which nothing can have arrived (nap0) and briefly between passes (nap).
template <int PCELL, int NG = 1>
__device__ __forceinline__ bool sweep_cells(const unsigned long long *slot, int C, int S, unsigned tag, const int (&cell)[PCELL],
                                            float (&v)[PCELL][4 * NG], long long t_start, int nap0, int nap) {
  // NG groups of 4 stream slots per cell: 32*NG bytes, 2*NG loads
  const __amdgpu_buffer_rsrc_t rs = buf_rsrc(slot, C * 32 * NG);
  for (int i = 0; i < nap0; i++) __builtin_amdgcn_s_sleep(4);
  for (unsigned spins = 0;; spins++) {
    u32x4 q[PCELL][2 * NG];
#pragma unroll
    for (int j = 0; j < PCELL; j++) {
      const int cl = cell[j] < C ? cell[j] : 0;
#pragma unroll
      for (int h = 0; h < 2 * NG; h++) q[j][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, cl * 32 * NG + 16 * h, 0, 16);   // aux 16 = sc1
    }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < PCELL; j++) {
      bool okc = true;
#pragma unroll
      for (int h = 0; h < 2 * NG; h++) {
        // (rvalue copies first: __builtin_bit_cast applied directly to a vector-element expression read element 0 for .z)
        const unsigned u0 = q[j][h].x, t0 = q[j][h].y, u1 = q[j][h].z, t1 = q[j][h].w;
        okc &= ((S < 2 * h + 1) | (t0 == tag)) & ((S < 2 * h + 2) | (t1 == tag));
        v[j][2 * h] = __uint_as_float(u0); v[j][2 * h + 1] = __uint_as_float(u1);
      }
      ok &= okc | (cell[j] >= C);
    }
    if (ok) return true;
    if ((spins & 31) == 31 && wall_clock64() - t_start > SPIN_LIMIT) return false;
    for (int i = 0; i < nap; i++) __builtin_amdgcn_s_sleep(1);
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
// forward: steps 1..T (+ one more exchange for r(T) when the projection runs here).
// Wave roles: the first 4*TPW waves are CELL waves -- wave w owns cell (w & 3) of tile (w >> 2) of the workgroup: its four
// gate rows over the whole K = [m | x] in the 4-row geometry (16 k-groups x 4 rows per MFMA), so the contraction of a
// cell needs no cross-wave combine: in-wave butterfly, cell math on lanes 12..15 (one per stream), granule + plane stores,
// all in that wave.  ONE workgroup barrier per step (slab ready).  Wave 4*TPW projects, the remaining waves sweep.
// The folded rows come from the 16-row packed gates operand, gathered once behind step 1.
// -------------------------------------------------------------------------------------------------------------------
// One contraction of a cell wave: NCHUNK 128-wide chunks of resident rows (a0/a1) against slab row `bj` (stride LD).
// Exactly NCHUNK chunks, no branch: chunks beyond the operand have zero weights and read zero slab columns, so every LDS
// read of the step is issued before the first MFMA.  Returns the gate pre-activations g,i,f,o of (cell, stream lane&3) in
// lanes 12..15.
template <int NCHUNK>
__device__ __forceinline__ f32x4 cell_contract(const float4 (&a0)[NCHUNK], const float4 (&a1)[NCHUNK], const float *slab_row, int kg,
                                               int *read_flag = nullptr, int flag_value = 0) {
  float4 b0[NCHUNK], b1[NCHUNK];
#pragma unroll
  for (int i = 0; i < NCHUNK; i++) {
    const float *bp = slab_row + i * 128 + kg * 4;
    b0[i] = *reinterpret_cast<const float4 *>(bp); b1[i] = *reinterpret_cast<const float4 *>(bp + 64);
  }
  __builtin_amdgcn_sched_barrier(0);                 // (otherwise the scheduler sinks every read next to its MFMAs: one LDS round trip per chunk)
  if (read_flag) {                                   // tell the slab's writers that this wave holds its copy
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(read_flag, flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int i = 0; i < NCHUNK; i++) {
    const float av[8] = {a0[i].x, a0[i].y, a0[i].z, a0[i].w, a1[i].x, a1[i].y, a1[i].z, a1[i].w};
    const float bv[8] = {b0[i].x, b0[i].y, b0[i].z, b0[i].w, b1[i].x, b1[i].y, b1[i].z, b1[i].w};
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], bv[j], acc[j & 3], 0, 0, 0);
  }
  return kgroup_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));   // the 16 k-groups of a (row, stream) pair
}

// 128-wide chunks of the UNFOLDED step-1 operand [r(0) | pad | x(1)] (K = RP + I, RP = R rounded up to 32) that go with
// MAXC chunks of the folded one (R <= C: a projection)
constexpr int persist_maxu(int maxc) { return maxc == 7 ? 5 : maxc; }

// NG: groups of 4 streams (NumStream <= 4*NG); the weights stay where they are, every step contracts them NG times
template <int TPW, int MAXC, int PNW, int PCELL, int NG>
__global__ __launch_bounds__(PNW * 64) void k_fwd_persist(PersistFwdArgs a) {
  constexpr int PNT = PNW * 64, NCW = 4 * TPW, NSW = (PNW - NCW - 1) * 64, MAXU = persist_maxu(MAXC);   // cell waves, one projection wave, sweepers
  constexpr int SS = 4 * NG;                         // stream slots per cell (slab rows, granules)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int C = a.C, S = a.S, T = a.T, I = a.I, R = a.R, nch = a.nch;
  const int RP = (R + KCH - 1) / KCH * KCH;          // x columns of the step-1 slab start here (the layout of the packed gates operand)
  const int XP = a.nchm * KCH;                       // ... and here in the folded slab
  constexpr int LDB = MAXC * 128 + 16;               // (LDB mod 64 == 16: the 16-lane groups of ds_read_b128 hit 16 distinct slots)
  constexpr int LDU = MAXU * 128 + 16;
  float *ldsB = lds;                                 // [SS][LDB]: row s = [ m(t-1)[s][0..C) | pad | x(t)[s][0..I) | pad ]
  float *ldsU = lds + SS * LDB;                      // [SS][LDU]: row s = [ r(0)[s][0..R) | pad | x(1)[s][0..I) | pad ]   (step 1 only)
  unsigned *abortf = reinterpret_cast<unsigned *>(ldsU + SS * LDU);
  int *projf = reinterpret_cast<int *>(abortf + 1);  // last step whose slab the projection wave has read
  int *pubcnt = reinterpret_cast<int *>(abortf + 2); // publishes issued by this workgroup's cell waves so far (one count per wave and step)
  const bool proj_on = a.rin && (int)blockIdx.x * 4 < R;   // this workgroup contracts rows 4*blockIdx .. +3 of W_r_m
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long t_start = wall_clock64();
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // zero both slabs once: pad columns and rows of absent streams stay zero for the whole launch
  for (int i = tid; i < SS * (LDB + LDU); i += PNT) lds[i] = 0.f;
  if (tid == 0) { *abortf = 0u; *projf = 0; *pubcnt = 0; }
  __syncthreads();
  PT_DECL();

  // The roles run SEPARATE loops with the same barrier sequence (one lds_barrier per step, an abort check behind it):
  // inside one loop body the register allocator keeps the resident weights of the cell waves AND the sweep state of the
  // sweepers alive in every wave.
  if (wave < NCW) {
    // =========================== cell wave: cell (wave & 3) of tile (wave >> 2) ===========================
    const int tile = blockIdx.x * TPW + (wave >> 2), cw = wave & 3;
    const int kg = lane >> 2, bj = lane & 3;
    const int e_cell = tile * 4 + cw, es = lane & 3;
    const int lc = e_cell < C ? e_cell : 0;
    const size_t wrow = (size_t)bj * C + lc;         // this lane's weight row: gate bj of the cell (rows of the 4C axis are g,i,f,o blocks)
    // epilogue lanes: lanes 12..15 = streams 0..3 of the cell (where kgroup_sum leaves the totals)
    const float pre0 = a.bias[lc], pre1 = a.bias[C + lc], pre2 = a.bias[2 * C + lc], pre3 = a.bias[3 * C + lc];
    const float wpi = a.pi[lc], wpf = a.pf[lc], wpo = a.po[lc];
    // plane stores through buffer descriptors: one 32-bit lane offset per group and plane shape, frame in the scalar offset
    // (64-bit per-store addresses pushed the two-group kernel into scratch)
    const __amdgpu_buffer_rsrc_t rs_g = buf_rsrc(a.gifo, (T + 2) * S * 4 * C * 4), rs_c = buf_rsrc(a.cc, (T + 2) * S * C * 4);
    const __amdgpu_buffer_rsrc_t rs_h = buf_rsrc(a.hh, (T + 2) * S * C * 4), rs_m = buf_rsrc(a.mm, (T + 2) * S * C * 4);
    bool e_ong[NG];
    float cpg[NG];                                   // c(t-1) of (cell, stream 4g + es)
#pragma unroll
    for (int g = 0; g < NG; g++) {
      e_ong[g] = (lane >> 2) == 3 && 4 * g + es < S && e_cell < C;
      cpg[g] = a.prev_c[(size_t)(e_ong[g] ? 4 * g + es : 0) * C + lc];            // carried c(0) (:231)
      if (e_ong[g]) a.cc[(size_t)(4 * g + es) * C + e_cell] = cpg[g];              // time block 0 of the c plane: BPTT reads it (:231)
    }
    auto cell_math = [&](int t, int g, const f32x4 &v) {
      if (!e_ong[g]) return;
      const int es_g = 4 * g + es;                   // the stream
      float &cp = cpg[g];
      float ag = v.x + pre0;
      float ai = v.y + pre1;
      float af = v.z + pre2;
      float ao = v.w + pre3;
      ai += wpi * cp;                              // :278
      af += wpf * cp;                              // :281
      const float gi = k_sigmoid(ai), gf = k_sigmoid(af), gg = k_tanh(ag);   // :284-288
      float c = gg * gi;                           // :291
      c = c + cp * gf;                             // :294
      c = c < -50.f ? -50.f : c;                   // :296
      c = c > 50.f ? 50.f : c;                     // :297
      const float h = k_tanh(c);                   // :300
      ao += wpo * c;                               // :303
      const float go = k_sigmoid(ao);              // :306
      const float m = h * go;                      // :309
      if (t < T || a.rin) publish(a.gran + (size_t)(t & 1) * C * SS, e_cell * SS + es_g, epoch + (unsigned)t, m);   // (m(T): for r(T) only)
      const int vg = (es_g * 4 * C + e_cell) * 4, vc = (es_g * C + e_cell) * 4, sg = t * S * 4 * C * 4, sc = t * S * C * 4;
      buf_store_f32(rs_g, vg, sg, gg); buf_store_f32(rs_g, vg, sg + C * 4, gi);
      buf_store_f32(rs_g, vg, sg + 2 * C * 4, gf); buf_store_f32(rs_g, vg, sg + 3 * C * 4, go);
      buf_store_f32(rs_c, vc, sc, c);
      buf_store_f32(rs_h, vc, sc, h);
      buf_store_f32(rs_m, vc, sc, m);
      if (t == T) a.prev_c[(size_t)es_g * C + e_cell] = c;       // :331 (c columns)
      cp = c;
    };
      // ---- step 1 closes over the CARRIED r (:275; set by Reset / the previous minibatch, possibly under older weights):
      // unfolded rows [W_gifo_r | W_gifo_x] straight from the natural matrices, dead after step 1
      float4 u0[MAXU], u1[MAXU];
#pragma unroll
      for (int i = 0; i < MAXU; i++) {
        float4 w[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int k0 = 128 * i + 64 * h + 4 * kg;
          const bool in_r = e_cell < C && k0 < R, in_x = e_cell < C && k0 >= RP && k0 - RP < I;
          const float *ap = in_x ? a.wx + wrow * I + (k0 - RP) : a.wr + wrow * R + (in_r ? k0 : 0);
          w[h] = *reinterpret_cast<const float4 *>(ap);
          if (!in_r && !in_x) w[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        u0[i] = w[0]; u1[i] = w[1];
      }
      float4 a0[MAXC], a1[MAXC];
      auto load_folded = [&]() {
        // ---- steps 2..T: resident folded rows [W_rm | W_x], k = 128*chunk + 64*h + 4*kg + e.  W_rm from the packed operand the
        // fold product writes (klstm_kernels.hip: pk[tile][chunk32][h32][lane32][4], row = lane32 & 15,
        // k = 32*chunk32 + 8*(lane32 >> 4) + 4*h32 + e), W_x from the natural W_gifo_x.
#pragma unroll
        for (int i = 0; i < MAXC; i++) {
          float4 w[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int k0 = 128 * i + 64 * h + 4 * kg;
            const int c32 = k0 >> 5, l32 = ((k0 & 31) >> 3) * 16 + 4 * cw + bj, h32 = (k0 & 7) >> 2;
            const bool in_m = e_cell < C && c32 < a.nchm, in_x = e_cell < C && k0 >= XP && k0 - XP < I;
            const float4 *ap = in_x ? reinterpret_cast<const float4 *>(a.wx + wrow * I + (k0 - XP))
                                    : a.wpk + (((size_t)(in_m ? tile : 0) * nch + (in_m ? c32 : 0)) * 2 + h32) * 64 + l32;
            w[h] = *ap;
            if (!in_m && !in_x) w[h] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          a0[i] = w[0]; a1[i] = w[1];
        }
      };
      // one group: requested now, behind the step-1 rows (they arrive while step 1 computes and travels); two groups: both
      // row sets at once do not fit the 168 registers of a 12-wave workgroup, requested once the step-1 rows are dead
      if (NG == 1) load_folded();
      PT_MARK(5);
      lds_barrier();                                 // slab of step 1 ready
      PT_MARK(1);
      f32x4 v1[NG];
#pragma unroll
      for (int g = 0; g < NG; g++) {
        if (g) __builtin_amdgcn_sched_barrier(0);
        v1[g] = cell_contract<MAXU>(u0, u1, ldsU + (4 * g + bj) * LDU, kg);
      }
      PT_MARK(2);
      if (NG > 1) load_folded();
#pragma unroll
      for (int g = 0; g < NG; g++) cell_math(1, g, v1[g]);
      if (lane == 0) __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // publishes of this step issued
      PT_MARK(4);
    bool dead = false;
    for (int t = 2; t <= T; t++) {
      PT_MARK(5);
      lds_barrier();                                 // slab of step t ready
      PT_MARK(1);
      if (*abortf) { dead = true; break; }           // (plain LDS read: the asm barrier's memory clobber forces the reload; a volatile
                                                     //  read through the generic pointer became a FLAT load behind vmcnt(0))
      f32x4 vt[NG];
#pragma unroll
      for (int g = 0; g < NG; g++) {
        if (g) __builtin_amdgcn_sched_barrier(0);    // (one group's 56 operand registers at a time)
        vt[g] = cell_contract<MAXC>(a0, a1, ldsB + (4 * g + bj) * LDB, kg);
      }
      PT_MARK(2);                                    // contractions + k-group sums
#pragma unroll
      for (int g = 0; g < NG; g++) cell_math(t, g, vt[g]);   // (back to back: the groups' dependent exp/rcp chains interleave)
      if (lane == 0) __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // publishes of this step issued
      PT_MARK(4);                                    // cell math + stores
    }
    if (a.rin && !dead) lds_barrier();               // (slab of m(T) for the projection wave)
  } else if (wave == NCW) {
    // =========================== projection wave: r(t-1) = W_r_m m(t-1) (:312) from the slab of step t ===========================
    // Rows 4*blockIdx .. +3 of W_r_m resident (same 4-row geometry, K = C), the first R/4 workgroups; everything it does
    // sits off the critical path (the cell waves read the same slab at the same time).  Writes the r plane, the output rows
    // (:328) and, for frame T, the carried r (:331).  rin == 0 / other workgroups: keeps the barrier count only.
    const int kg = lane >> 2, bj = lane & 3;
    const int prow = (int)blockIdx.x * 4 + bj;
    float4 a0[MAXC], a1[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; i++) {
      const int k = 128 * i + 4 * kg;
      a0[i] = proj_on && k < C ? *reinterpret_cast<const float4 *>(a.wm + (size_t)prow * C + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      a1[i] = proj_on && k + 64 < C ? *reinterpret_cast<const float4 *>(a.wm + (size_t)prow * C + k + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    lds_barrier();                                   // step 1
    for (int t = 2; t <= T + (a.rin ? 1 : 0); t++) {
      lds_barrier();
      if (*abortf) break;
      if (!proj_on) continue;
      // (columns >= C of the slab row hold x(t) and pad: their weights are zero)
#pragma unroll
      for (int g = 0; g < NG; g++) {
        if (g) __builtin_amdgcn_sched_barrier(0);
        const f32x4 v = cell_contract<MAXC>(a0, a1, ldsB + (4 * g + bj) * LDB, kg, g == NG - 1 ? projf : nullptr, t);
        const int ps = 4 * g + bj;
        if (kg == 3 && ps < S) {                     // lanes 12..15: stream ps, components = rows 4*blockIdx .. +3
          const int f = t - 1, g4 = (int)blockIdx.x * 4;
          *reinterpret_cast<float4 *>(a.rr + ((size_t)f * S + ps) * R + g4) = make_float4(v.x, v.y, v.z, v.w);
          float *op = a.out + ((size_t)(f - 1) * S + ps) * a.out_stride + g4;
          op[0] = v.x; op[1] = v.y; op[2] = v.z; op[3] = v.w;
          if (f == T) *reinterpret_cast<float4 *>(a.prev_r + (size_t)ps * R + g4) = make_float4(v.x, v.y, v.z, v.w);
        }
      }
    }
  } else {
    // =========================== sweeper: the B operand of every step into the slab ===========================
    const int sidx = (wave - NCW - 1) * 64 + lane;   // rank among the sweeping threads
    int cell[PCELL];
#pragma unroll
    for (int j = 0; j < PCELL; j++) cell[j] = sidx + j * NSW;
    const int nx4 = I / 4;                           // float4 per x row; the first sweeper wave also stages x(t)
    const bool x_on = sidx < S * nx4;
    const int xs = x_on ? sidx / nx4 : 0, xk = x_on ? (sidx % nx4) * 4 : 0;
    for (int t = 1; t <= T + (a.rin ? 1 : 0); t++) {  // (rin: one more slab, m(T), for r(T))
      PT_MARK(5);
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x_on && t <= T) xv = *reinterpret_cast<const float4 *>(a.x + ((size_t)(t - 1) * S + xs) * a.x_stride + xk);
      if (t == 1) {
        // step 1: the carried r(0) (:231, :275) into the unfolded slab, and into time block 0 of the r plane (BPTT reads it)
        for (int i = sidx; i < S * (R / 4); i += NSW) {
          const int s = i / (R / 4), k = (i % (R / 4)) * 4;
          const float4 rv = *reinterpret_cast<const float4 *>(a.prev_r + (size_t)s * R + k);
          *reinterpret_cast<float4 *>(ldsU + s * LDU + k) = rv;
          if (blockIdx.x == 0) *reinterpret_cast<float4 *>(a.rr + (size_t)s * R + k) = rv;
        }
        if (x_on) *reinterpret_cast<float4 *>(ldsU + xs * LDU + RP + xk) = xv;
      } else {
        float mv[PCELL][SS];                         // m(t-1) of every cell
        // Polling starts once this workgroup's OWN cell waves have issued their publishes of step t-1 (the others are about
        // as far): sweeper loads already in the CU's vector-memory queue hold the publishes back, and with them the whole
        // exchange.  A fixed sleep tuned to the cell waves' epilogue did the same job (nap0 = 9: 2.35 us per step, 2.7 at
        // 7 or 12; twice that for two stream groups); the event needs no tuning: 2.16-2.24 us for nap0 = 0..3.
        for (unsigned spins = 0; __hip_atomic_load(pubcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NCW * (t - 1); spins++) {
          __builtin_amdgcn_s_sleep(1);
          if ((spins & 1023) == 1023 && wall_clock64() - t_start > SPIN_LIMIT) break;   // (bounded like every other spin: the sweep below then times out and reports)
        }
        if (!sweep_cells<PCELL, NG>(a.gran + (size_t)((t - 1) & 1) * C * SS, C, S, epoch + (unsigned)(t - 1), cell, mv, t_start, a.nap0, a.nap)) {
          *abortf = 1u;
          if (lane == 0) atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t);
        }
        // (the slab is free: the sweep only completes once every cell wave of THIS workgroup has published step t-1,
        //  i.e. has finished reading the previous slab; the projection wave says so itself)
        if (proj_on && t > 2)
          while (__hip_atomic_load(projf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < t - 1) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int j = 0; j < PCELL; j++)
          if (cell[j] < C) {
#pragma unroll
            for (int s = 0; s < SS; s++) if (s < S) ldsB[s * LDB + cell[j]] = mv[j][s];
          }
        if (x_on) *reinterpret_cast<float4 *>(ldsB + xs * LDB + XP + xk) = xv;
      }
      PT_MARK(0);                                    // sweep + slab store
      lds_barrier();
      PT_MARK(1);
      if (*abortf) break;
    }
  }
  PT_FLUSH(0);
  finish(a.ctrl, epoch, T);
}

// -------------------------------------------------------------------------------------------------------------------
// backward: steps T..1.  Iteration t: every sweeping thread receives d_m(t) of its cell (all streams), recomputes
// dgifo(t) (:411-440) from it and its replica of the carry, and stores it into the operand slab; the 4 K waves of a tile
// (one per SIMD) contract their quarter of K = 4C with the resident W_rm^T rows, the tile's first K wave combines the four
// partial tiles, adds P(t-1) and publishes d_m(t-1).  The sweeping thread of a cell that belongs to this workgroup also
// writes that cell's dgifo / dc rows (plain stores, acknowledged long before its next sweep starts).  Two workgroup
// barriers per step (slab ready, partial tiles ready); the sweepers wait at the second one instead of polling the fabric
// while the K waves work.
// -------------------------------------------------------------------------------------------------------------------
struct BpttCarry { float dcn, din, dfn, fn; };
// The elementwise BPTT of a (cell, stream) pair (:411-440) is linear in d_m(t): everything else -- the forward planes of
// frame t, the carry of frame t+1 -- is known BEFORE d_m(t) has crossed the fabric.  The sweepers fold it into six
// coefficients while they would otherwise nap in front of the first poll, and spend 5 operations per pair once d_m is in:
//   d_h = d_m*[yo(1-yh^2)]   d_o = d_m*[yh*yo(1-yo)]   d_c = d_m*k1 + pre,  k1 = ah + wpo*ao,  pre = dcn*fn + wpi*din + wpf*dfn
//   d_f = d_c*[cpv*yf(1-yf)] d_i = d_c*[yg*yi(1-yi)]   d_g = d_c*[yi(1-yg^2)]
// (same algebra as :411-440, products associated differently: a few ulp, identical in every workgroup)
struct BpttCoef { float k1, ao, pre, bf, bi, bg; };
__device__ __forceinline__ BpttCoef bptt_coef(float yg, float yi, float yf, float yo, float yh, float cpv, float wpi, float wpf,
                                              float wpo, const BpttCarry &k) {
  BpttCoef c;
  const float ah = __builtin_fmaf(-yo, yh * yh, yo);                 // :411-412
  c.ao = yh * __builtin_fmaf(-yo, yo, yo);                          // :415-416
  c.k1 = __builtin_fmaf(wpo, c.ao, ah);                             // :424, :428
  c.pre = __builtin_fmaf(wpf, k.dfn, __builtin_fmaf(wpi, k.din, k.dcn * k.fn));   // :425-427
  c.bf = cpv * __builtin_fmaf(-yf, yf, yf);                         // :431-432
  c.bi = yg * __builtin_fmaf(-yi, yi, yi);                          // :435-436
  c.bg = __builtin_fmaf(-yi, yg * yg, yi);                          // :439-440
  return c;
}
__device__ __forceinline__ float4 bptt_apply(float dm, const BpttCoef &c, float yf, BpttCarry &k, float &d_c_out) {
  const float d_c = __builtin_fmaf(dm, c.k1, c.pre);
  const float d_o = dm * c.ao, o_g = d_c * c.bg, o_i = d_c * c.bi, o_f = d_c * c.bf;
  k.dcn = d_c; k.din = o_i; k.dfn = o_f; k.fn = yf;                  // f(t) is the f(t+1) of the next iteration
  d_c_out = d_c;
  return make_float4(o_g, o_i, o_f, d_o);
}
// one (cell, stream) pair of frame t; returns d(g,i,f,o) and updates the carry
__device__ __forceinline__ float4 bptt_cell(float dm, float yg, float yi, float yf, float yo, float yh, float cpv, float wpi,
                                            float wpf, float wpo, BpttCarry &k, float &d_c_out) {
  const float d_h = k_diff_tanh_fma(dm * yo, yh);    // :411-412   (single-rounding fp32 forms, see klstm_math.h)
  const float d_o = k_diff_sigmoid_fma(dm * yh, yo); // :415-416
  float d_c = d_h;                                   // :424
  d_c = d_c + k.dcn * k.fn;                          // :425
  d_c = d_c + wpi * k.din;                           // :426
  d_c = d_c + wpf * k.dfn;                           // :427
  d_c = d_c + wpo * d_o;                             // :428
  const float o_f = k_diff_sigmoid_fma(d_c * cpv, yf);   // :431-432
  const float o_i = k_diff_sigmoid_fma(d_c * yg, yi);    // :435-436
  const float o_g = k_diff_tanh_fma(d_c * yi, yg);       // :439-440
  k.dcn = d_c; k.din = o_i; k.dfn = o_f; k.fn = yf;  // f(t) is the f(t+1) of the next iteration
  d_c_out = d_c;
  return make_float4(o_g, o_i, o_f, d_o);
}

template <int TPW, int MAXC, int PNW, int PCELL>
__global__ __launch_bounds__(PNW * 64) void k_bwd_persist(PersistBwdArgs a) {
  constexpr int PNT = PNW * 64, NKW = 4 * TPW, NSW = (PNW - NKW - 1) * 64;   // K waves, one P wave, sweepers
  constexpr int LDD = 4 * MAXC * 128 + 16;           // (LDD mod 64 == 16: the 16-lane groups of ds_read_b128 hit 16 distinct slots)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int C = a.C, S = a.S, T = a.T, nch = a.nch, K = 4 * a.C;
  float *ldsD = lds;                                 // [4][LDD]: dgifo(t) rows, natural g|i|f|o order; columns >= 4C stay zero
  const int wslab = a.din ? 4 * LDD : 0;
  float *ldsW = lds + 4 * LDD;                       // [4][LDD] (din): 4 rows of W_gifo_r^T / W_gifo_x^T, same column order
  f32x4 *red = reinterpret_cast<f32x4 *>(lds + 4 * LDD + wslab);      // [NKW][4]
  unsigned *abortf = reinterpret_cast<unsigned *>(red + NKW * 4);
  constexpr int PLW = 4 * TPW;                       // own cells
  int *dflag = reinterpret_cast<int *>(abortf + 1);  // lowest step whose slab the d_r contraction has read (counts down)
  float *ldsDC = reinterpret_cast<float *>(abortf + 4);       // [4][PLW]: d_c(t) of the own cells (sweeper -> P wave -> dc plane)
  float *ldsP = ldsDC + 4 * 4 * TPW;                          // [T*S][PLW]: own columns of P = out_diff W_r_m (pin)
  // the d_r / in_diff columns of this workgroup: 4 rows of W_gifo_r^T (workgroups 0 .. R/4-1), then of W_gifo_x^T
  const int ngr = a.R / 4, ngx = (a.din & 2) ? a.I / 4 : 0;
  const bool d_on = a.din && (int)blockIdx.x < ngr + ngx, d_isr = (int)blockIdx.x < ngr;
  const int dcol = d_isr ? (int)blockIdx.x * 4 : ((int)blockIdx.x - ngr) * 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long t_start = wall_clock64();
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = tid; i < 4 * LDD + wslab; i += PNT) ldsD[i] = 0.f;
  if (tid == 0) { *abortf = 0u; *dflag = T + 2; }
  __syncthreads();
  if (d_on) {
    const float *src = (d_isr ? a.wrT : a.wxT) + (size_t)dcol * K;
    for (int i = tid; i < K; i += PNT) {             // K/4 float4 per row, 4 rows
      const int row = i / (K / 4), k4 = i % (K / 4);
      *reinterpret_cast<float4 *>(ldsW + row * LDD + 4 * k4) = *reinterpret_cast<const float4 *>(src + (size_t)row * K + 4 * k4);
    }
  }                                                  // (visible to the P wave behind the first barrier below)

  // separate loops per role, same barrier sequence (two lds_barriers per step, the abort check behind the first): see k_fwd_persist
  if (wave < NKW) {
    // =========================== K wave: quarter (wave & 3) of K for tile (wave >> 2) ===========================
    const int tl = wave >> 2, kw = wave & 3;
    const int tile = blockIdx.x * TPW + tl;
    const bool owner_wave = kw == 0;
    const int kg = lane >> 2, bj = lane & 3;
    float4 a0[MAXC], a1[MAXC];                       // resident weights: chunks kw, kw+4, ... of the tile's 4 rows (zero beyond the operand)
#pragma unroll
    for (int i = 0; i < MAXC; i++) {
      const int ch = kw + 4 * i;
      const bool on = ch < nch && tile * 4 < C;
      const float4 *ap = a.wpk + ((size_t)(on ? tile : 0) * nch + (on ? ch : 0)) * 128 + lane;
      a0[i] = ap[0]; a1[i] = ap[64];
      if (!on) { a0[i] = make_float4(0.f, 0.f, 0.f, 0.f); a1[i] = a0[i]; }
    }
    // epilogue lanes of the owner wave: lanes 0..15 = (cell 4*tile + lane/4, stream lane%4) receive d_m(t-1) of that pair
    const int e_i = (lane >> 2) & 3, e_j = lane & 3;
    const int e_cell = tile * 4 + e_i;
    const bool e_on = owner_wave && lane < 16 && e_j < S && e_cell < C;
    const __amdgpu_buffer_rsrc_t rs_p = buf_rsrc(a.P, T * S * C * 4);
    const int e_offc = e_on ? (e_j * C + e_cell) * 4 : 0;
    const float *e_pl = ldsP + (e_j < S ? e_j : 0) * PLW + tl * 4 + e_i;      // this lane's column of the LDS copy of P
    PT_DECL();
    if (a.pin) {
      lds_barrier();                                 // P of frames T, T-1 ready (P wave)
      // d_m(T) = P(T) (dgifo(T+1) = 0, :351) travels like every other step
      if (e_on) publish(a.gran + (size_t)(T & 1) * C * 4, e_cell * 4 + e_j, epoch + (unsigned)T, e_pl[(size_t)(T - 1) * S * PLW]);
    }
    bool dead = false;
    for (int t = T; t > 1; t--) {
      PT_MARK(5);
      // P(t-1) for the epilogue: requested now, consumed after the contraction (frame t-1 is row block t-2 of P)
      const float pnext = !owner_wave ? 0.f : a.pin ? e_pl[(size_t)(t - 2) * S * PLW] : buf_f32(rs_p, e_offc, (t - 2) * S * C * 4);
      lds_barrier();                                 // slab dgifo(t) ready
      PT_MARK(1);
      if (*abortf) { dead = true; break; }
      float4 b0[MAXC], b1[MAXC];
#pragma unroll
      for (int i = 0; i < MAXC; i++) {
        const float *bp = ldsD + bj * LDD + (kw + 4 * i) * 128 + kg * 4;
        b0[i] = *reinterpret_cast<const float4 *>(bp); b1[i] = *reinterpret_cast<const float4 *>(bp + 64);
      }
      __builtin_amdgcn_sched_barrier(0);             // every LDS read of the step in flight before the first MFMA
      f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int i = 0; i < MAXC; i++) {
        const float av[8] = {a0[i].x, a0[i].y, a0[i].z, a0[i].w, a1[i].x, a1[i].y, a1[i].z, a1[i].w};
        const float bv[8] = {b0[i].x, b0[i].y, b0[i].z, b0[i].w, b1[i].x, b1[i].y, b1[i].z, b1[i].w};
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], bv[j], acc[j & 3], 0, 0, 0);
      }
      const f32x4 v = kgroup_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
      if ((lane >> 2) == 3) red[wave * 4 + (lane & 3)] = v;   // lanes 12..15: streams 0..3, components = the tile's 4 cells
      PT_MARK(2);
      lds_barrier();                                 // partial tiles ready
      PT_MARK(3);
      if (owner_wave) {
        // row (cell) e_i of stream e_j: component e_i of red[4*tl + w][e_j], the four K quarters in fixed order
        const float *rp = reinterpret_cast<const float *>(red) + ((tl * 4) * 4 + e_j) * 4 + e_i;
        const float sum = ((rp[0] + rp[16]) + rp[32]) + rp[48];
        const float dmv = sum + pnext;               // :408 with :391 substituted: d_m(t-1) = contraction + P(t-1)
        if (e_on) publish(a.gran + (size_t)((t - 1) & 1) * C * 4, e_cell * 4 + e_j, epoch + (unsigned)(t - 1), dmv);
      }
      PT_MARK(4);
    }
    if (a.din && !dead) lds_barrier();               // (slab dgifo(1): in_diff of frame 1 only)
    PT_FLUSH(0);
  } else if (wave == NKW) {
    // =========================== P wave: own columns of P = out_diff W_r_m (:391's second term through :408) ===========================
    // Same 4-row geometry as the contraction: A = rows of W_r_m^T (the tile's 4 cells; R <= 512 = 4 chunks),
    // B = 4 rows of out_diff per batch; lanes 12..15 end up with P[row 4b + (lane & 3)][cells 0..3].  Frames T and T-1
    // before the first barrier, then one frame ahead of the owner's epilogue, entirely inside the time this wave would
    // otherwise spend waiting at the barriers.  (pin == 0: the wave only keeps the barrier count.)
    const int kg = lane >> 2, bj = lane & 3, R = a.R, rows = T * S;
    int next = (rows + 3) / 4 - 1;                   // batches in descending row order
    auto p_batches = [&](int row_lo) {               // every batch that holds a row >= row_lo
      while (next >= 0 && 4 * next + 3 >= row_lo) {
        const int row = 4 * next + bj;
        const float *op = a.od + (size_t)(row < rows ? row : 0) * a.od_stride + 4 * kg;
        float4 b0[4], b1[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = 128 * i + 4 * kg;
          b0[i] = row < rows && k < R ? *reinterpret_cast<const float4 *>(op + 128 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
          b1[i] = row < rows && k + 64 < R ? *reinterpret_cast<const float4 *>(op + 128 * i + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int tl = 0; tl < TPW; tl++) {
          // (rows of W_r_m^T re-read per batch, L1/L2 hits: resident they would push the d_r ring below into scratch)
          float4 w0[4], w1[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int k = 128 * i + 4 * kg, pc = ((int)blockIdx.x * TPW + tl) * 4 + bj;
            w0[i] = pc < C && k < R ? *reinterpret_cast<const float4 *>(a.wmT + (size_t)pc * R + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            w1[i] = pc < C && k + 64 < R ? *reinterpret_cast<const float4 *>(a.wmT + (size_t)pc * R + k + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float av[8] = {w0[i].x, w0[i].y, w0[i].z, w0[i].w, w1[i].x, w1[i].y, w1[i].z, w1[i].w};
            const float bv[8] = {b0[i].x, b0[i].y, b0[i].z, b0[i].w, b1[i].x, b1[i].y, b1[i].z, b1[i].w};
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], bv[j], acc[j & 3], 0, 0, 0);
          }
          const f32x4 v = kgroup_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
          if (kg == 3 && row < rows) *reinterpret_cast<float4 *>(ldsP + (size_t)row * PLW + tl * 4) = make_float4(v.x, v.y, v.z, v.w);
        }
        next--;
      }
    };
    if (a.pin) {
      p_batches((T - 2) * S);
      lds_barrier();
    }
    // d_r(T) = out_diff(T): dgifo(T+1) = 0 (:351, :391)
    if (d_on && d_isr && kg == 3 && bj < S)
      *reinterpret_cast<float4 *>(a.dr + ((size_t)T * S + bj) * R + dcol) =
          *reinterpret_cast<const float4 *>(a.od + ((size_t)(T - 1) * S + bj) * a.od_stride + dcol);
    const int nchw = (K + 127) / 128;
    for (int t = T; t >= (a.din ? 1 : 2); t--) {
      if (a.pin && t > 1) p_batches((t - 3) * S);    // frame t-2, read by the owner at step t-1
      lds_barrier();                                 // slab dgifo(t) ready
      if (*abortf) break;
      // rows of the dgifo / dc planes of this workgroup's cells, out of the slab (lane = stream, gate, cell)
#pragma unroll
      for (int tl = 0; tl < TPW; tl++) {
        const int os = lane >> 4, og = (lane >> 2) & 3, oc = ((int)blockIdx.x * TPW + tl) * 4 + (lane & 3);
        if (os < S && oc < C) a.dgifo[((size_t)t * S + os) * K + og * C + oc] = ldsD[os * LDD + og * C + oc];
        if (lane < 16) {
          const int ds = lane >> 2;
          if (ds < S && oc < C) a.dc[((size_t)t * S + ds) * C + oc] = ldsDC[ds * (4 * TPW) + tl * 4 + (lane & 3)];
        }
      }
      if (t > 1) lds_barrier();
      // dgifo(t) against this workgroup's 4 rows of W_gifo_r^T: d_r(t-1) = out_diff(t-1) + dgifo(t) W_gifo_r (:391);
      // against 4 rows of W_gifo_x^T: in_diff(t) (:457).  Both operands in LDS, behind the second barrier: the K waves
      // are done with the slab, the sweepers are out on the fabric, and they ask dflag before they overwrite it.
      if (d_on && (t > 1 || !d_isr)) {
        // groups of 2 chunks, double-buffered: the next group's 8 LDS reads in flight under the current group's 16 MFMAs (a rolled
        // read -> MFMA loop pays one LDS round trip per chunk: 2.2 us per step, and the sweepers then wait for dflag)
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const float *ap = ldsW + bj * LDD + kg * 4, *bp = ldsD + bj * LDD + kg * 4;
        float4 qa[2][2][2], qb[2][2][2];
        auto dload = [&](int u, int g) {
#pragma unroll
          for (int c = 0; c < 2; c++) {
            qa[u][c][0] = *reinterpret_cast<const float4 *>(ap + (2 * g + c) * 128); qa[u][c][1] = *reinterpret_cast<const float4 *>(ap + (2 * g + c) * 128 + 64);
            qb[u][c][0] = *reinterpret_cast<const float4 *>(bp + (2 * g + c) * 128); qb[u][c][1] = *reinterpret_cast<const float4 *>(bp + (2 * g + c) * 128 + 64);
          }
        };
        auto dmul = [&](int u) {
#pragma unroll
          for (int c = 0; c < 2; c++) {
            const float av[8] = {qa[u][c][0].x, qa[u][c][0].y, qa[u][c][0].z, qa[u][c][0].w, qa[u][c][1].x, qa[u][c][1].y, qa[u][c][1].z, qa[u][c][1].w};
            const float bv[8] = {qb[u][c][0].x, qb[u][c][0].y, qb[u][c][0].z, qb[u][c][0].w, qb[u][c][1].x, qb[u][c][1].y, qb[u][c][1].z, qb[u][c][1].w};
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], bv[j], acc[j & 3], 0, 0, 0);
          }
        };
        const int ngrp = (nchw + 1) / 2;             // (a chunk past the operand reads zero columns of both slabs: 4*MAXC chunks exist)
        dload(0, 0);
        for (int g = 0; g < ngrp; g += 2) {
          if (g + 1 < ngrp) dload(1, g + 1);
          dmul(0);
          if (g + 2 < ngrp) dload(0, g + 2);
          if (g + 1 < ngrp) dmul(1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(dflag, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const f32x4 v = kgroup_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
        if (kg == 3 && bj < S) {                     // lanes 12..15: stream bj, components = columns dcol .. +3
          if (d_isr) {
            const float4 o = *reinterpret_cast<const float4 *>(a.od + ((size_t)(t - 2) * S + bj) * a.od_stride + dcol);
            *reinterpret_cast<float4 *>(a.dr + ((size_t)(t - 1) * S + bj) * R + dcol) = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
          } else {
            float *ip = a.in_diff + ((size_t)(t - 1) * S + bj) * a.id_stride + dcol;
            ip[0] = v.x; ip[1] = v.y; ip[2] = v.z; ip[3] = v.w;
          }
        }
      } else if (d_on) {
        __hip_atomic_store(dflag, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  } else {
    // =========================== sweeper: d_m(t) of every cell -> dgifo(t) into the slab ===========================
    // forward planes through buffer descriptors: the lane offset is cell*4 bytes, frame / stream / gate go into the scalar
    // offset (64-bit per-load addresses cost two VGPRs each and pushed this kernel into scratch)
    const __amdgpu_buffer_rsrc_t rs_g = buf_rsrc(a.gifo, (T + 2) * S * K * 4), rs_h = buf_rsrc(a.hh, (T + 2) * S * C * 4);
    const __amdgpu_buffer_rsrc_t rs_c = buf_rsrc(a.cc, (T + 2) * S * C * 4), rs_p = buf_rsrc(a.P, T * S * C * 4);
    int cell[PCELL];
    bool mine[PCELL];                                // the cell belongs to this workgroup's tiles: this thread writes its plane rows
    float wpi[PCELL], wpf[PCELL], wpo[PCELL];
    int voff[PCELL];
    BpttCarry kk[PCELL][4];
#pragma unroll
    for (int j = 0; j < PCELL; j++) {
      cell[j] = (wave - NKW - 1) * 64 + lane + j * NSW;
      const int lc = cell[j] < C ? cell[j] : 0;
      mine[j] = cell[j] < C && (cell[j] >> 2) / TPW == (int)blockIdx.x;
      wpi[j] = a.pi[lc]; wpf[j] = a.pf[lc]; wpo[j] = a.po[lc];
      voff[j] = lc * 4;
#pragma unroll
      for (int s = 0; s < 4; s++) kk[j][s] = BpttCarry{0.f, 0.f, 0.f, 0.f};
      if (mine[j]) {                                 // the batched d_r product reads dgifo(T+1) as operand rows: keep them zero (:351)
        for (int s = 0; s < S; s++) {
          float *zp = a.dgifo + ((size_t)(T + 1) * S + s) * K + cell[j];
          zp[0] = 0.f; zp[C] = 0.f; zp[2 * C] = 0.f; zp[3 * C] = 0.f;
        }
      }
    }
    if (a.pin) lds_barrier();                       // (P of the last two frames ready: the P wave)
    PT_DECL();
    for (int t = T; t >= 1; t--) {
      PT_MARK(5);
      const int sg = t * S * K * 4, sc = t * S * C * 4;
      // planes of frame t for this thread's cells: requested before the sweep (L2-resident; every workgroup reads the same rows)
      float yg[PCELL][4], yi[PCELL][4], yf[PCELL][4], yo[PCELL][4], yh[PCELL][4], cpv[PCELL][4], dm[PCELL][4];
#pragma unroll
      for (int j = 0; j < PCELL; j++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const int ss = s < S ? s : 0;              // absent streams re-read stream 0 (their results are never stored)
          const int og = sg + ss * K * 4, oc = sc + ss * C * 4;
          yg[j][s] = buf_f32(rs_g, voff[j], og); yi[j][s] = buf_f32(rs_g, voff[j], og + C * 4);
          yf[j][s] = buf_f32(rs_g, voff[j], og + 2 * C * 4); yo[j][s] = buf_f32(rs_g, voff[j], og + 3 * C * 4);
          yh[j][s] = buf_f32(rs_h, voff[j], oc); cpv[j][s] = buf_f32(rs_c, voff[j], oc - S * C * 4);
          if (t == T && !a.pin) dm[j][s] = buf_f32(rs_p, voff[j], ((T - 1) * S + ss) * C * 4);   // d_m(T) = P(T): dgifo(T+1) = 0
        }
      // coefficient pass: waits for the plane loads, which is the nap in front of the first poll
      BpttCoef cf[PCELL][4];
      float yfk[PCELL][4];
#pragma unroll
      for (int j = 0; j < PCELL; j++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
          cf[j][s] = bptt_coef(yg[j][s], yi[j][s], yf[j][s], yo[j][s], yh[j][s], cpv[j][s], wpi[j], wpf[j], wpo[j], kk[j][s]);
          yfk[j][s] = yf[j][s];
        }
      if ((t < T || a.pin) && !sweep_cells(a.gran + (size_t)(t & 1) * C * 4, C, S, epoch + (unsigned)t, cell, dm, t_start, a.nap0, a.nap)) {
        *abortf = 1u;
        if (lane == 0) atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t);
      }
      PT_MARK(0);                                    // plane loads + sweep
      if (d_on && t < T)                             // the d_r contraction still reads the slab of step t+1?
        while (__hip_atomic_load(dflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > t + 1) __builtin_amdgcn_s_sleep(1);
      // elementwise BPTT of frame t (:411-440), replicated in every workgroup
      float4 dgk[PCELL][4];
      float dck[PCELL][4];
#pragma unroll
      for (int j = 0; j < PCELL; j++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
          dgk[j][s] = bptt_apply(dm[j][s], cf[j][s], yfk[j][s], kk[j][s], dck[j][s]);
          if (cell[j] < C && s < S && (t > 1 || a.din)) {   // B operand of the contraction (t == 1: of in_diff(1) only)
            float *lp = ldsD + s * LDD + cell[j];
            lp[0] = dgk[j][s].x; lp[C] = dgk[j][s].y; lp[2 * C] = dgk[j][s].z; lp[3 * C] = dgk[j][s].w;
            if (mine[j]) ldsDC[s * (4 * TPW) + cell[j] - (int)blockIdx.x * 4 * TPW] = dck[j][s];
          }
        }
      // own cells: rows of the dgifo / dc planes (gradient products).  The P wave copies them out of the slab behind the
      // barrier (as sweeper stores in front of it, the one wave per workgroup that owns cells held everybody up: 0.9 ->
      // 0.56 us for this phase); only the last frame without an in_diff step has no barrier and does it here
      auto own_rows = [&]() {
#pragma unroll
        for (int j = 0; j < PCELL; j++)
#pragma unroll
          for (int s = 0; s < 4; s++)
            if (mine[j] && s < S) {
              const size_t row = (size_t)t * S + s;
              float *dp = a.dgifo + row * K + cell[j];
              dp[0] = dgk[j][s].x; dp[C] = dgk[j][s].y; dp[2 * C] = dgk[j][s].z; dp[3 * C] = dgk[j][s].w;
              a.dc[row * C + cell[j]] = dck[j][s];
            }
      };
      PT_MARK(2);                                    // elementwise + slab stores
      if (t == 1 && !a.din) { own_rows(); break; }
      lds_barrier();                                 // slab ready
      PT_MARK(1);
      if (*abortf || t == 1) break;
      lds_barrier();                                 // (partial tiles ready: nothing to do here but keep the count; the K waves
                                                     //  contract meanwhile, and polling the fabric now would only slow them down)
      PT_MARK(3);
    }
    PT_FLUSH(0);
  }
  finish(a.ctrl, epoch, T);
}

// -------------------------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------------------------
static inline int pcdiv(int a, int b) { return (a + b - 1) / b; }

static int g_persist_tpw = 0;       // A-B knobs: tiles per workgroup / waves per workgroup (0 = automatic)
static int g_persist_waves = 0;
static int g_persist_nap0 = -1, g_persist_nap = -1;   // -1: defaults below
static int g_persist_nap0_bwd = -1;                   // backward launch separately (-1: follows g_persist_nap0)
void set_persist_nap0_bwd(int v) { g_persist_nap0_bwd = v; }
void set_persist_nap(int nap0, int nap) { if (nap0 >= -1) g_persist_nap0 = nap0; if (nap >= -1) g_persist_nap = nap; }
void set_persist_tpw(int v) { g_persist_tpw = v; }
void set_persist_waves(int v) { g_persist_waves = v; }

// Geometry.  Fewer, fatter workgroups mean fewer sweepers per exchange (less fabric contention).
//   forward : 4 cell waves per tile (one per cell, whole K in registers: maxc = 128-wide chunks), the rest sweep
//   backward: waves/tpw waves per tile split K (maxc = chunks per wave), one of them owns the tile, the rest sweep
struct PGeo { int waves, tpw, maxc, pcell; };
static PGeo pick_geo_fwd(int C, int nch, int ku = 0) {   // ku: width of the step-1 operand [r | x]
  const int waves = (g_persist_waves == 8 || g_persist_waves == 16) ? g_persist_waves : 12;   // measured at 40/800/512: 12 waves, 1 tile
  const int prefer[3] = {g_persist_tpw ? g_persist_tpw : 1, 1, 2};
  const int n128 = pcdiv(nch * KCH, 128);
  for (int tpw : prefer) {
    if ((tpw != 1 && tpw != 2) || 4 * tpw >= waves || (C / 4) % tpw != 0 || C / 4 / tpw > 200 || n128 > 12) continue;
    if (4 * tpw + 1 >= waves) continue;
    const int pc = pcdiv(C, (waves - 4 * tpw - 1) * 64);     // (one wave projects)
    if (pc > 4) continue;
    const int maxc = n128 <= 7 ? 7 : n128 <= 9 ? 9 : 12;
    if (pcdiv(ku, 128) > persist_maxu(maxc)) continue;
    return PGeo{waves, tpw, maxc, pc};
  }
  return PGeo{0, 0, 0, 0};
}
static PGeo pick_geo(int C, int nch) {              // backward: 4 K waves per tile, chunk slots per K wave
  const int waves = (g_persist_waves == 8 || g_persist_waves == 16) ? g_persist_waves : 12;
  const int prefer[3] = {g_persist_tpw ? g_persist_tpw : 1, 1, 2};
  const int mc = pcdiv(nch, 4);
  for (int tpw : prefer) {
    if ((tpw != 1 && tpw != 2) || 4 * tpw >= waves || (C / 4) % tpw != 0 || C / 4 / tpw > 200 || mc > 9) continue;
    if (4 * tpw + 1 >= waves) continue;
    const int pc = pcdiv(C, (waves - 4 * tpw - 1) * 64);     // (one wave contracts P)
    if (pc > 4) continue;
    return PGeo{waves, tpw, mc <= 7 ? 7 : 9, pc};
  }
  return PGeo{0, 0, 0, 0};
}

// forward: up to 8 streams (two groups of 4 against the same resident rows); backward: up to 4
bool persist_supported(const Dims &d) {
  if (d.S > 8 || d.C % 8 != 0 || d.I % 8 != 0 || d.R % 4 != 0 || d.S * (d.I / 4) > 192) return false;
  const int nf = pcdiv(d.C, KCH) + pcdiv(d.I, KCH), nb = pcdiv(4 * d.C, 128);
  const PGeo gf = pick_geo_fwd(d.C, nf, pcdiv(d.R, KCH) * KCH + d.I);
  if (d.S > 4 && gf.waves != 12) return false;
  return gf.tpw > 0 && pick_geo(d.C, nb).tpw > 0;
}
bool persist_bwd_supported(const Dims &d) { return d.S <= 4 && persist_supported(d); }
size_t persist_gran_bytes(const Dims &d) { return (size_t)2 * d.C * 8 * sizeof(unsigned long long); }   // (up to 8 stream slots per cell)

template <class K, class A>
static hipError_t plaunch(K kern, int grid, int threads, size_t shm, hipStream_t st, LaunchProbe pr, const A &a) {
  if (shm > 64 * 1024)                               // above the default dynamic-LDS limit (cell dim 1024 backward)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (pr.start) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, st, pr.start, pr.stop, 0, a);
  else hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, st, a);
  return hipGetLastError();
}
#define PD5(KERN, TP, MC, W)                                                                                    \
  if (g.waves == W && g.tpw == TP && g.maxc == MC) {                                                            \
    if (g.pcell == 1) return plaunch(KERN<TP, MC, W, 1>, grid, W * 64, shm, st, pr, a);                         \
    if (g.pcell == 2) return plaunch(KERN<TP, MC, W, 2>, grid, W * 64, shm, st, pr, a);                         \
    if (g.pcell == 3) return plaunch(KERN<TP, MC, W, 3>, grid, W * 64, shm, st, pr, a);                         \
    return plaunch(KERN<TP, MC, W, 4>, grid, W * 64, shm, st, pr, a);                                           \
  }
// forward: the same with the stream-group count; two groups (5..8 streams) only in the 12-wave geometries
#define PF5(KERN, TP, MC, W, NG_)                                                                               \
  if (g.waves == W && g.tpw == TP && g.maxc == MC && ng == NG_) {                                               \
    if (g.pcell == 1) return plaunch(KERN<TP, MC, W, 1, NG_>, grid, W * 64, shm, st, pr, a);                    \
    if (g.pcell == 2) return plaunch(KERN<TP, MC, W, 2, NG_>, grid, W * 64, shm, st, pr, a);                    \
    if (g.pcell == 3) return plaunch(KERN<TP, MC, W, 3, NG_>, grid, W * 64, shm, st, pr, a);                    \
    return plaunch(KERN<TP, MC, W, 4, NG_>, grid, W * 64, shm, st, pr, a);                                      \
  }
#define PDISPATCH_FWD(KERN)                                                                                     \
  do {                                                                                                          \
    PF5(KERN, 1, 7, 8, 1) PF5(KERN, 1, 9, 8, 1) PF5(KERN, 1, 12, 8, 1)                                          \
    PF5(KERN, 1, 7, 12, 1) PF5(KERN, 1, 9, 12, 1) PF5(KERN, 1, 12, 12, 1) PF5(KERN, 2, 7, 12, 1) PF5(KERN, 2, 9, 12, 1) PF5(KERN, 2, 12, 12, 1) \
    PF5(KERN, 1, 7, 16, 1) PF5(KERN, 1, 9, 16, 1) PF5(KERN, 1, 12, 16, 1) PF5(KERN, 2, 7, 16, 1) PF5(KERN, 2, 9, 16, 1) PF5(KERN, 2, 12, 16, 1) \
    PF5(KERN, 1, 7, 12, 2) PF5(KERN, 1, 9, 12, 2) PF5(KERN, 1, 12, 12, 2) PF5(KERN, 2, 7, 12, 2) PF5(KERN, 2, 9, 12, 2) PF5(KERN, 2, 12, 12, 2) \
    return hipErrorInvalidValue;                                                                                \
  } while (0)
#define PDISPATCH_BWD(KERN)                                                                                     \
  do {                                                                                                          \
    PD5(KERN, 1, 7, 8) PD5(KERN, 1, 9, 8)                                                                       \
    PD5(KERN, 1, 7, 12) PD5(KERN, 1, 9, 12) PD5(KERN, 2, 7, 12) PD5(KERN, 2, 9, 12)                             \
    PD5(KERN, 1, 7, 16) PD5(KERN, 1, 9, 16) PD5(KERN, 2, 7, 16) PD5(KERN, 2, 9, 16)                             \
    return hipErrorInvalidValue;                                                                                \
  } while (0)

// r(t) = W_r_m m(t) inside the forward launch: 4 rows of W_r_m per workgroup on its projection wave
bool persist_r_in_kernel(const Dims &d) {
  const PGeo g = pick_geo_fwd(d.C, pcdiv(d.C, KCH) + pcdiv(d.I, KCH), pcdiv(d.R, KCH) * KCH + d.I);
  return g.tpw > 0 && d.R % 4 == 0 && d.R / 4 <= d.C / 4 / g.tpw;
}

hipError_t launch_fwd_persist(const Dims &d, const FwdPtrs &p, const float *in, int in_stride, float *out, int out_stride,
                              unsigned long long *gran, unsigned *ctrl, hipStream_t st, LaunchProbe pr) {
  PersistFwdArgs a;
  a.C = d.C; a.I = d.I; a.R = d.R; a.S = d.S; a.T = d.T;
  a.nchm = pcdiv(d.C, KCH); a.nch = a.nchm + pcdiv(d.I, KCH);
  a.wpk = p.pk_fold; a.wr = p.wr; a.wx = p.wx; a.wm = p.wm; a.bias = p.bias; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.mm = p.mm; a.rr = p.rr;
  a.x = in; a.x_stride = in_stride; a.prev_c = p.prev_c; a.prev_r = p.prev_r; a.gran = gran; a.ctrl = ctrl;
  a.rin = out && persist_r_in_kernel(d); a.out = out; a.out_stride = out_stride;
  a.nap0 = g_persist_nap0 >= 0 ? g_persist_nap0 : d.S > 4 ? 4 : 2; a.nap = g_persist_nap >= 0 ? g_persist_nap : 0;     // (behind the publish flag; measured: tools/persist_anatomy, tools/nap_sweep.py)
  const PGeo g = pick_geo_fwd(d.C, a.nch, pcdiv(d.R, KCH) * KCH + d.I);
  if (!g.tpw || !p.pk_fold || (reinterpret_cast<uintptr_t>(in) & 15) || in_stride % 4 != 0) return hipErrorInvalidValue;
  const int ng = d.S > 4 ? 2 : 1;
  const size_t shm = (size_t)(4 * ng * (g.maxc * 128 + 16) + 4 * ng * (persist_maxu(g.maxc) * 128 + 16) + 4) * sizeof(float);   // (+ abort flag, projection flag)
  const int grid = d.C / 4 / g.tpw;
  PDISPATCH_FWD(k_fwd_persist);
}

// P = out_diff W_r_m inside the backward launch: own columns in LDS (T*S rows x 4*tpw cells), rows of W_r_m^T in registers
bool persist_p_in_kernel(const Dims &d) {
  const PGeo g = pick_geo(d.C, pcdiv(4 * d.C, 128));
  return g.tpw > 0 && d.R <= 512 && d.R % 4 == 0 && (size_t)d.T * d.S * 4 * g.tpw * sizeof(float) <= 32 * 1024;
}

// d_r and in_diff inside the backward launch: 4 columns per workgroup on its P wave, operand rows in LDS
bool persist_tail_in_kernel(const Dims &d, bool want_in_diff) {
  const PGeo g = pick_geo(d.C, pcdiv(4 * d.C, 128));
  if (!g.tpw || d.R % 4 != 0 || d.I % 4 != 0 || d.C % 4 != 0) return false;
  const size_t lds = (size_t)(8 * (4 * g.maxc * 128 + 16) + 4 * g.tpw * 16 + 4 + 16 * g.tpw + (persist_p_in_kernel(d) ? d.T * d.S * 4 * g.tpw : 0)) * sizeof(float);
  return d.R / 4 + (want_in_diff ? d.I / 4 : 0) <= d.C / 4 / g.tpw && lds <= 160 * 1024;
}

hipError_t launch_bwd_persist(const Dims &d, const BwdPtrs &p, const float *P, const float *out_diff, int od_stride,
                              float *in_diff, int id_stride, bool tail_inside, unsigned long long *gran, unsigned *ctrl,
                              hipStream_t st, LaunchProbe pr) {
  PersistBwdArgs a;
  a.C = d.C; a.R = d.R; a.S = d.S; a.T = d.T;
  a.pin = persist_p_in_kernel(d) && out_diff && (reinterpret_cast<uintptr_t>(out_diff) & 15) == 0 && od_stride % 4 == 0;
  if (!a.pin && !P) return hipErrorInvalidValue;
  a.od = out_diff; a.od_stride = od_stride; a.wmT = p.wmT;
  a.din = tail_inside ? (in_diff ? 3 : 1) : 0; a.I = d.I; a.wrT = p.wrT; a.wxT = p.wxT; a.dr = p.dr; a.in_diff = in_diff; a.id_stride = id_stride;
  if (a.din && (!persist_tail_in_kernel(d, in_diff != nullptr) || !out_diff || (reinterpret_cast<uintptr_t>(out_diff) & 15) || od_stride % 4 != 0))
    return hipErrorInvalidValue;
  a.nch = pcdiv(4 * d.C, 128);
  a.wpk = p.pk_fold; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.dgifo = p.dgifo; a.dc = p.dc; a.P = P; a.gran = gran; a.ctrl = ctrl;
  a.nap0 = g_persist_nap0_bwd >= 0 ? g_persist_nap0_bwd : g_persist_nap0 >= 0 ? g_persist_nap0 : 0;     // (the second barrier already keeps the sweepers off the fabric)
  a.nap = g_persist_nap >= 0 ? g_persist_nap : 0;
  const PGeo g = pick_geo(d.C, a.nch);
  if (!g.tpw || !p.pk_fold) return hipErrorInvalidValue;
  const size_t shm = (size_t)((a.din ? 8 : 4) * (4 * g.maxc * 128 + 16) + 4 * g.tpw * 4 * 4 + 4 + 16 * g.tpw + (a.pin ? d.T * d.S * 4 * g.tpw : 0)) * sizeof(float);
  const int grid = d.C / 4 / g.tpw;
  PDISPATCH_BWD(k_bwd_persist);
}

}  // namespace klstm
