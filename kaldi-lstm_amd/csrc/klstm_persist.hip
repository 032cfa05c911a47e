// kaldi-lstm_amd/csrc/klstm_persist.hip -- weights-RESIDENT recurrence chain (engine option "persist"), FORWARD direction, up
// to 8 streams (the backward direction: klstm_persist_bwd.hip); the default chain from 1 to 8 streams.
//
// The launch-per-step chain (klstm_kernels.hip) re-fetches its whole weight operand (~10.5 MB at 40/800/512) in every one
// of the 2T step kernels because nothing on-chip survives a kernel boundary: 12x the algorithmic HBM traffic of a
// minibatch.  Here ONE kernel runs all steps of the folded recurrence
//     a(t) = W_x x(t) + b + W_rm m(t-1)            (...streams.h:275 with r(t-1) = W_r_m m(t-1), :312)
// with each workgroup's rows of the packed operand [W_rm | W_x] held in VGPRs for the whole minibatch, and the per-step
// all-to-all (every workgroup needs all of m(t-1): S x C floats, 12.8 KB at 4 x 800) done INSIDE the launch:
//   * transport = data-tagged 8-byte granules {tag, fp32 value}, one sc1 (write-through, agent-scope relaxed atomic)
//     store per (cell, stream) by the owning lane, swept with 16-byte sc1 buffer loads by every workgroup until every tag
//     matches (cdna_hip_programming.md Guideline 16 recipe R2: the data IS the flag, no fence, no separate flag;
//     MI355X_MICROARCH.md "allgather" row).  Placement-independent: no dependence on dispatch order or XCD.
//   * two granule slots (parity of t): a workgroup can publish step t+1 only after it has seen ALL of step t, and every
//     workgroup publishes step t only after its sweep of step t-1 has finished, so a slot is never rewritten while
//     somebody still sweeps it.
//   * tags = epoch + t with a device-resident epoch that the last workgroup to finish advances by T + 2: no per-call
//     memset, and a hipGraph replay (frozen kernel arguments) still sees fresh tags.
//   * every wait is bounded (wall clock, 50 ms PER WAIT by default): on expiry the workgroup records the step in ctrl[2] and
//     leaves; the Update kernels read that word and leave the parameters alone, the engine reports it at the next
//     synchronising call and falls back.  All workgroups must be co-resident: grid <= 200 workgroups, one per CU (the engine
//     checks the CU count when it is created).
//   * wave roles: the waves that own cell math, granule stores and plane stores do NOT sweep (loads return in order behind
//     a wave's own stores: a sweeping wave with write-through stores in flight would wait for their acknowledgement
//     first); one wave per workgroup carries the product that hangs off the chain (r = W_r_m m); all other waves sweep,
//     PCELL cells per thread.
//   * barriers wait for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): nothing global is ordered by them.
// Step 1 closes over the CARRIED r (possibly produced under older weights) and is contracted against the natural
// [W_gifo_r | W_gifo_x] rows inside the same launch.
//
// Geometry = the 4-row form of v_mfma_f32_4x4x1_16b (16 blocks = 16 k-groups of one 4 rows x 4 streams tile, chunk = 128 k,
// A lane 4b+i = row i, B lane 4b+j = stream j), on the packed operand the fold product writes: a cell wave holds the 4 gate
// rows of ONE cell over the whole K = [m | x] (gathered from the 16-row gates operand).
// A workgroup owns TPW tiles (1 by default: 200 workgroups of 12 waves at C = 800).
#include "klstm_kernels.h"
#include "klstm_math.h"
#include "klstm_persist_dev.h"

#include <hip/hip_ext.h>

namespace klstm {

#pragma clang fp contract(off)

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
  const float *xg;                // non-null: x(t) W_gifo_x^T + bias is already in the gifo plane (batched product, :246/:259): wide inputs;
                                  // the kernel then runs with I = 0 (no x columns in the slabs, no x rows of the operands)
  const float *prev_c;            // carried c [S x C], read at step 1 (:231)
  const float *prev_r;            // carried r [S x R], read at step 1
  float *next_c, *next_r;         // c(T) (:331) and, with rin, r(T) go here: ANOTHER buffer (the engine flips the pair per minibatch), so
                                  // that a launch that gives up leaves the state it started from untouched
  unsigned *guard;                // the engine's control words (or null): [2] / [6] set by an earlier launch -> do nothing; [8] = persistent
                                  // launches of this engine that have run so far: this launch's ordinal, [8] + 1, goes into ctrl[3] if it gives up
  unsigned long long *gran;       // [2][C*4] granules, cell-major (4 stream slots per cell)
  unsigned *ctrl;                 // [0] epoch, [1] finished workgroups, [2] status (0 = ok)
  int nap0, nap;                  // sweepers sleep nap0 x 256 clocks before the first pass of a step, nap x 64 between passes
  long long spin_limit;           // wall-clock ticks (100 MHz) a single wait may take before the workgroup gives up
  int test_stall;                 // test hook: workgroup 0 stops publishing at this step (0: never) -> every sweep of that step times out
  unsigned *hstat;                // host-mapped status word (or null): set when a wait expires, read by the engine without a sync
#ifdef KLSTM_PERSIST_TIMING
  long long *dbg;                 // per workgroup: shader-clock sums of the phases of a step (tools/persist_anatomy.hip)
#endif
};

// Sweep the granules of a step until every tag of a live stream equals `tag`; returns false on timeout.  The slab of a step is an
// array of 16-byte units (unit u = 2 NG cell + h: {m, tag} of streams 2h and 2h + 1 of the cell); a sweeping thread takes units
// first + k stride, k < NUQ, so that every load instruction of a wave reads ONE contiguous KB (whole cells per thread put the lanes
// of an instruction 32 NG bytes apart: 2 NG instructions over the same lines, 2 NG x the line requests -- 8 streams: 96.4 -> 82.1 us
// per launch at T = 20, 4 streams: 2.10 -> 1.98 us per step).  All loads in flight before the first check; branch-free inside a pass.
// A polling wave competes with the cell / owner waves of its own CU for the vector-memory queue (their plane and granule
// stores queue behind its loads: measured 1.2-2.8 us for a 7-store epilogue next to unthrottled pollers), so a sweeper
// sleeps through the part of the step in which nothing can have arrived (nap0) and briefly between passes (nap).
template <int NUQ>
__device__ __forceinline__ bool sweep_units(const unsigned long long *slot, int nunits, int S, int upc, unsigned tag, int first, int stride,
                                            float (&v)[NUQ][2], long long limit, int nap0, int nap) {
  const __amdgpu_buffer_rsrc_t rs = buf_rsrc(slot, nunits * 16);
  for (int i = 0; i < nap0; i++) __builtin_amdgcn_s_sleep(4);
  const long long t0 = wall_clock64();
  for (unsigned spins = 0;; spins++) {
    u32x4 q[NUQ];
#pragma unroll
    for (int k = 0; k < NUQ; k++) {
      const int u = first + k * stride;
      q[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u < nunits ? u : 0) * 16, 0, 16);   // aux 16 = sc1
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NUQ; k++) {
      const int u = first + k * stride, h = u % upc;
      const unsigned u0 = q[k].x, t0_ = q[k].y, u1 = q[k].z, t1 = q[k].w;
      ok &= (((S < 2 * h + 1) | (t0_ == tag)) & ((S < 2 * h + 2) | (t1 == tag))) | (u >= nunits);
      v[k][0] = __uint_as_float(u0); v[k][1] = __uint_as_float(u1);
    }
    if (ok) return true;
    if ((spins & 31) == 31 && wall_clock64() - t0 > limit) return false;
    for (int i = 0; i < nap; i++) __builtin_amdgcn_s_sleep(1);
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
// XB: the x-projection is the caller's batched product (a.xg); compiled apart so that the default kernels carry nothing of it
// IL (NG = 2 .. 4, round 6): the groups of 4 streams as INTERLEAVED chains -- a group's step needs that group's m(t-1) from every
// workgroup and nothing of the other group.  Until round 5 both groups moved in lock-step (one sweep over all 8 streams' granules, one
// barrier, two contractions, two cell updates, then everybody waits for the 8-stream exchange: 4.1 us per step against 2.2 at 4 streams).
// Here every role walks (t, group 0), (t, group 1), (t + 1, group 0), ...: one barrier per (t, group); while group 0's m(t) crosses the
// fabric the workgroup sweeps, contracts and updates group 1, whose granules were published half a step earlier and are mostly there.
// Granules lie group-major ([parity][group][C][4 streams]) so that a group's sweep reads whole lines.  Same arithmetic per (cell,
// stream): bit-identical to the lock-step form (tests/test_persist_robustness_gpu.py).
template <int TPW, int MAXC, int PNW, int PCELL, int NG, bool XB = false, bool IL = false>
__global__ __launch_bounds__(PNW * 64) void k_fwd_persist(PersistFwdArgs a) {
  static_assert(!IL || (NG >= 2 && NG <= 4), "interleaved chains: two to four stream groups");
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
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // a launch queued behind one that gave up (its status word is still set: the host has not looked yet) must not run: its
  // inputs are that launch's invalid outputs.  Requested here, looked at behind the barrier of step 1 -- behind every role's
  // prologue loads (loads return in order: the look costs nothing; a branch up here cost 1.1 us per launch).  Before that barrier
  // a launch only mirrors the carried state into time block 0 of the c / r planes; behind it a skipping launch does nothing.
  unsigned behind_giveup = 0u;
#ifndef KLSTM_NO_CHAIN_GUARD
  if (a.guard) behind_giveup = __hip_atomic_load(&a.guard[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                               __hip_atomic_load(&a.guard[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
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
    constexpr bool xb = XB;
    const float pre0 = xb ? 0.f : a.bias[lc], pre1 = xb ? 0.f : a.bias[C + lc], pre2 = xb ? 0.f : a.bias[2 * C + lc], pre3 = xb ? 0.f : a.bias[3 * C + lc];
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
    // (the guard word was requested before prev_c and loads return in order: it is here.  Looked at NOW, in front of the weight
    //  requests -- behind them the compiler waits for ALL of them, and the folded rows are meant to arrive under step 1: 1.3 us)
    const bool skip_launch = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
    auto cell_math = [&](int t, int g, const f32x4 &v, const float4 &xp) {
      if (!e_ong[g]) return;
      const int es_g = 4 * g + es;                   // the stream
      float &cp = cpg[g];
      float ag = v.x + (XB ? xp.x : pre0);           // (XB: x(t) W_x^T + bias as the batched product left it)
      float ai = v.y + (XB ? xp.y : pre1);
      float af = v.z + (XB ? xp.z : pre2);
      float ao = v.w + (XB ? xp.w : pre3);
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
      if ((t < T || a.rin) && !(a.test_stall == t && blockIdx.x == 0))                                           // (m(T): for r(T) only)
        publish(a.gran + (size_t)(t & 1) * C * SS, IL ? (g * C + e_cell) * 4 + es : e_cell * SS + es_g, epoch + (unsigned)t, m);
      const int vg = (es_g * 4 * C + e_cell) * 4, vc = (es_g * C + e_cell) * 4, sg = t * S * 4 * C * 4, sc = t * S * C * 4;
      buf_store_f32(rs_g, vg, sg, gg); buf_store_f32(rs_g, vg, sg + C * 4, gi);
      buf_store_f32(rs_g, vg, sg + 2 * C * 4, gf); buf_store_f32(rs_g, vg, sg + 3 * C * 4, go);
      buf_store_f32(rs_c, vc, sc, c);
      buf_store_f32(rs_h, vc, sc, h);
      buf_store_f32(rs_m, vc, sc, m);
      if (t == T) a.next_c[(size_t)es_g * C + e_cell] = c;       // :331 (c columns)
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
      // batched x-projection: the four pre-activations of this lane's (cell, stream) of frame t, requested ahead of the barrier
      auto load_xp = [&](int t, float4 (&xp)[NG]) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
          xp[g] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (xb && e_ong[g]) {
            const float *gp = a.xg + ((size_t)t * S + 4 * g + es) * 4 * C + e_cell;
            xp[g] = make_float4(gp[0], gp[C], gp[2 * C], gp[3 * C]);
          }
        }
      };
      float4 xp1[NG];
      load_xp(1, xp1);
      PT_MARK(5);
      lds_barrier();                                 // slab of step 1 ready
      PT_MARK(1);
    if (!skip_launch) {
      f32x4 v1[NG];
#pragma unroll
      for (int g = 0; g < NG; g++) {
        if (g) __builtin_amdgcn_sched_barrier(0);
        v1[g] = cell_contract<MAXU>(u0, u1, ldsU + (4 * g + bj) * LDU, kg);
      }
      PT_MARK(2);
      if (NG > 1) load_folded();
#pragma unroll
      for (int g = 0; g < NG; g++) cell_math(1, g, v1[g], xp1[g]);
      if (lane == 0) __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // publishes of this step issued
      PT_MARK(4);
    bool dead = false;
    if constexpr (IL) {
      for (int t = 2; t <= T && !dead; t++) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
          if (dead) break;
          float4 xpg = make_float4(0.f, 0.f, 0.f, 0.f);
          if (xb && e_ong[g]) {
            const float *gp = a.xg + ((size_t)t * S + 4 * g + es) * 4 * C + e_cell;
            xpg = make_float4(gp[0], gp[C], gp[2 * C], gp[3 * C]);
          }
          PT_MARK(5);
          lds_barrier();                             // slab rows of group g for step t ready
          PT_MARK(1);
          if (*abortf) { dead = true; break; }
          const f32x4 v = cell_contract<MAXC>(a0, a1, ldsB + (4 * g + bj) * LDB, kg);
          PT_MARK(2);
          cell_math(t, g, v, xpg);
          if (lane == 0) __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // publishes of (t, g) issued
          PT_MARK(4);
        }
      }
      if (a.rin && !dead)                            // (slabs of m(T) of the groups for the projection wave)
        for (int g = 0; g < NG; g++) { lds_barrier(); if (*abortf) break; }
    } else {
    for (int t = 2; t <= T; t++) {
      float4 xpt[NG];
      load_xp(t, xpt);
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
      for (int g = 0; g < NG; g++) cell_math(t, g, vt[g], xpt[g]);   // (back to back: the groups' dependent exp/rcp chains interleave)
      if (lane == 0) __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // publishes of this step issued
      PT_MARK(4);                                    // cell math + stores
    }
    if (a.rin && !dead) lds_barrier();               // (slab of m(T) for the projection wave)
    }
    }
  } else if (wave == NCW) {
    // =========================== projection wave: r(t-1) = W_r_m m(t-1) (:312) from the slab of step t ===========================
    // Rows 4*blockIdx .. +3 of W_r_m resident (same 4-row geometry, K = C), the first R/4 workgroups; everything it does
    // sits off the critical path (the cell waves read the same slab at the same time).  Writes the r plane, the output rows
    // (:328) and, for frame T, the carried r (:331).  rin == 0 / other workgroups: keeps the barrier count only.
    const int kg = lane >> 2, bj = lane & 3;
    const int prow = (int)blockIdx.x * 4 + bj;
    const bool skip = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;     // (in front of this wave's own requests; it is off the chain until step 2)
    float4 a0[MAXC], a1[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; i++) {
      const int k = 128 * i + 4 * kg;
      a0[i] = proj_on && k < C ? *reinterpret_cast<const float4 *>(a.wm + (size_t)prow * C + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      a1[i] = proj_on && k + 64 < C ? *reinterpret_cast<const float4 *>(a.wm + (size_t)prow * C + k + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    lds_barrier();                                   // step 1
    if constexpr (IL) {
      int itn = 0;
      bool out_ = false;
      for (int t = 2; !skip && !out_ && t <= T + (a.rin ? 1 : 0); t++) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
          lds_barrier();
          if (*abortf) { out_ = true; break; }
          ++itn;
          if (!proj_on) continue;
          const f32x4 v = cell_contract<MAXC>(a0, a1, ldsB + (4 * g + bj) * LDB, kg, projf, itn);
          const int ps = 4 * g + bj;
          if (kg == 3 && ps < S) {                   // lanes 12..15: stream ps, components = rows 4*blockIdx .. +3
            const int f = t - 1, g4 = (int)blockIdx.x * 4;
            *reinterpret_cast<float4 *>(a.rr + ((size_t)f * S + ps) * R + g4) = make_float4(v.x, v.y, v.z, v.w);
            float *op = a.out + ((size_t)(f - 1) * S + ps) * a.out_stride + g4;
            op[0] = v.x; op[1] = v.y; op[2] = v.z; op[3] = v.w;
            if (f == T) *reinterpret_cast<float4 *>(a.next_r + (size_t)ps * R + g4) = make_float4(v.x, v.y, v.z, v.w);
          }
        }
      }
    } else
    for (int t = 2; !skip && t <= T + (a.rin ? 1 : 0); t++) {
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
          if (f == T) *reinterpret_cast<float4 *>(a.next_r + (size_t)ps * R + g4) = make_float4(v.x, v.y, v.z, v.w);
        }
      }
    }
  } else {
    // =========================== sweeper: the B operand of every step into the slab ===========================
    const int sidx = (wave - NCW - 1) * 64 + lane;   // rank among the sweeping threads
    const int nx4 = I / 4;                           // float4 per x row; the first sweeper wave also stages x(t)
    const bool x_on = sidx < S * nx4;
    const int xs = x_on ? sidx / nx4 : 0, xk = x_on ? (sidx % nx4) * 4 : 0;
    if constexpr (IL) {
      // step 1 (both groups at once, as in the lock-step form), then one iteration per (t, group)
      {
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x_on) xv = *reinterpret_cast<const float4 *>(a.x + (size_t)xs * a.x_stride + xk);
        for (int i = sidx; i < S * (R / 4); i += NSW) {
          const int s_ = i / (R / 4), k = (i % (R / 4)) * 4;
          const float4 rv = *reinterpret_cast<const float4 *>(a.prev_r + (size_t)s_ * R + k);
          *reinterpret_cast<float4 *>(ldsU + s_ * LDU + k) = rv;
          if (blockIdx.x == 0) *reinterpret_cast<float4 *>(a.rr + (size_t)s_ * R + k) = rv;
        }
        if (x_on) *reinterpret_cast<float4 *>(ldsU + xs * LDU + RP + xk) = xv;
        lds_barrier();
      }
      bool out_ = *abortf || __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
      for (int t = 2; !out_ && t <= T + (a.rin ? 1 : 0); t++) {
        for (int g = 0; g < NG; g++) {
          const int it = NG * (t - 2) + g;           // iteration index; the cell waves have issued NCW * (it + 2) publishes after it
          const int Sg = S - 4 * g < 4 ? S - 4 * g : 4;
          PT_MARK(5);
          const bool x_mine = x_on && (xs >> 2) == g && t <= T;
          float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (x_mine) xv = *reinterpret_cast<const float4 *>(a.x + ((size_t)(t - 1) * S + xs) * a.x_stride + xk);
          float mv[2 * PCELL][2];                    // m(t-1) of group g: 16-byte units (cell, stream pair)
          {
            const int target = t == 2 ? NCW : NCW * (it - NG + 2);   // publishes of (t - 1, g) issued by this workgroup's own cell waves (iteration it - NG)
            const long long w0 = wall_clock64();
            for (unsigned spins = 0; __hip_atomic_load(pubcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target; spins++) {
              __builtin_amdgcn_s_sleep(1);
              if ((spins & 1023) == 1023 && wall_clock64() - w0 > a.spin_limit) break;
            }
          }
          if (!sweep_units<2 * PCELL>(a.gran + (size_t)((t - 1) & 1) * C * SS + (size_t)g * C * 4, 2 * C, Sg, 2, epoch + (unsigned)(t - 1), sidx, NSW, mv,
                                      a.spin_limit, a.nap0, a.nap)) {
            *abortf = 1u;
            if (lane == 0) {
              atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
              atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t);
              if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
          // (group g's slab rows were read by the cell waves two barriers ago; the projection wave says when it has)
          if (proj_on && it >= NG)
            while (__hip_atomic_load(projf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < it - NG + 1) __builtin_amdgcn_s_sleep(1);
#pragma unroll
          for (int k = 0; k < 2 * PCELL; k++) {
            const int u = sidx + k * NSW, cl = u >> 1, s2 = 2 * (u & 1);
            if (cl < C) {
              if (s2 < Sg) ldsB[(4 * g + s2) * LDB + cl] = mv[k][0];
              if (s2 + 1 < Sg) ldsB[(4 * g + s2 + 1) * LDB + cl] = mv[k][1];
            }
          }
          if (x_mine) *reinterpret_cast<float4 *>(ldsB + xs * LDB + XP + xk) = xv;
          PT_MARK(0);
          lds_barrier();
          PT_MARK(1);
          if (*abortf) { out_ = true; break; }
        }
      }
    } else
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
        float mv[2 * NG * PCELL][2];                 // m(t-1): 16-byte units (cell, stream pair) first + k NSW
        // Polling starts once this workgroup's OWN cell waves have issued their publishes of step t-1 (the others are about
        // as far): sweeper loads already in the CU's vector-memory queue hold the publishes back, and with them the whole
        // exchange.  A fixed sleep tuned to the cell waves' epilogue did the same job (nap0 = 9: 2.35 us per step, 2.7 at
        // 7 or 12; twice that for two stream groups); the event needs no tuning: 2.16-2.24 us for nap0 = 0..3.
        {
          const long long w0 = wall_clock64();
          for (unsigned spins = 0; __hip_atomic_load(pubcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NCW * (t - 1); spins++) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 1023) == 1023 && wall_clock64() - w0 > a.spin_limit) break;   // (bounded like every other spin: the sweep below then times out and reports)
          }
        }
        if (!sweep_units<2 * NG * PCELL>(a.gran + (size_t)((t - 1) & 1) * C * SS, 2 * NG * C, S, 2 * NG, epoch + (unsigned)(t - 1), sidx, NSW, mv,
                                         a.spin_limit, a.nap0, a.nap)) {
          *abortf = 1u;
          if (lane == 0) {
            atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));            // (which launch: the first one wins, everything behind it does nothing)
            atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t);
            if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        // (the slab is free: the sweep only completes once every cell wave of THIS workgroup has published step t-1,
        //  i.e. has finished reading the previous slab; the projection wave says so itself)
        if (proj_on && t > 2)
          while (__hip_atomic_load(projf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < t - 1) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int k = 0; k < 2 * NG * PCELL; k++) {
          const int u = sidx + k * NSW, cl = u / (2 * NG), s2 = 2 * (u % (2 * NG));
          if (cl < C) {
            if (s2 < S) ldsB[s2 * LDB + cl] = mv[k][0];
            if (s2 + 1 < S) ldsB[(s2 + 1) * LDB + cl] = mv[k][1];
          }
        }
        if (x_on) *reinterpret_cast<float4 *>(ldsB + xs * LDB + XP + xk) = xv;
      }
      PT_MARK(0);                                    // sweep + slab store
      lds_barrier();
      PT_MARK(1);
      if (*abortf || __builtin_amdgcn_readfirstlane(behind_giveup) != 0u) break;
    }
  }
  PT_FLUSH(0);
  finish(a.ctrl, epoch, T + 2, a.guard ? a.guard + 8 : nullptr, a.guard, a.hstat ? a.hstat + 1 : nullptr);
}

// -------------------------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------------------------
static inline int pcdiv(int a, int b) { return (a + b - 1) / b; }

// Geometry.  Fewer, fatter workgroups mean fewer sweepers per exchange (less fabric contention).
//   forward : 4 cell waves per tile (one per cell, whole K in registers: maxc = 128-wide chunks), one projects, the rest sweep
struct PGeo { int waves, tpw, maxc, pcell; };
static PGeo pick_geo_fwd(const PersistOpts &o, int C, int nch, int ku = 0) {   // ku: width of the step-1 operand [r | x]
  const int waves = (o.waves == 8 || o.waves == 16) ? o.waves : 12;   // measured at 40/800/512: 12 waves, 1 tile
  const int prefer[3] = {o.tpw ? o.tpw : 1, 1, 2};
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

// forward: up to 8 streams (two groups of 4 against the same resident rows)
// Wide inputs (the 512 -> 800 / 512 inner layers of a stack): the x columns would take the operand past 9 chunks of 128 per
// cell wave (12 spill) and the x rows past what the sweepers stage; x(t) W_gifo_x^T + bias then stays the batched product
// of the reference (:246, :259) in front of the launch and the kernel adds it from the gifo plane.
bool persist_x_batched(const Dims &d) {
  return pcdiv((pcdiv(d.C, KCH) + pcdiv(d.I, KCH)) * KCH, 128) > 9 || d.S * (d.I / 4) > 192;
}
bool persist_supported(const Dims &d, const PersistOpts &o) {
  if (d.S > 16 || d.C % 8 != 0 || d.I % 8 != 0 || d.R % 4 != 0) return false;
  const int Ik = persist_x_batched(d) ? 0 : d.I;
  const int nf = pcdiv(d.C, KCH) + pcdiv(Ik, KCH);
  const PGeo gf = pick_geo_fwd(o, d.C, nf, pcdiv(d.R, KCH) * KCH + Ik);
  if ((d.S > 4 || persist_x_batched(d)) && (gf.waves != 12 || (persist_x_batched(d) && gf.maxc > 9))) return false;
  // 9..16 streams: three / four interleaved chains only (x inside the step, one tile per workgroup, <= 7 operand chunks of 128: C + I <= 896)
  if (d.S > 8 && (o.fwd_interleave == 0 || persist_x_batched(d) || gf.tpw != 1 || gf.maxc > 7)) return false;
  return gf.tpw > 0;
}
int persist_fwd_grid(const Dims &d, const PersistOpts &o) {
  const int Ik = persist_x_batched(d) ? 0 : d.I;
  const PGeo g = pick_geo_fwd(o, d.C, pcdiv(d.C, KCH) + pcdiv(Ik, KCH), pcdiv(d.R, KCH) * KCH + Ik);
  return g.tpw ? d.C / 4 / g.tpw : 0;
}
// forward: [2 parities][C][8 stream slots]; backward: [2 stream groups][BWD_RING = 32 ring slots][C][4 stream slots] (klstm_persist_bwd.hip)
size_t persist_gran_bytes(const Dims &d) { return (size_t)64 * d.C * 8 * sizeof(unsigned long long); }   // (backward: up to 4 groups x 32 ring slots)

template <class K, class A>
static hipError_t plaunch(K kern, int grid, int threads, size_t shm, hipStream_t st, LaunchProbe pr, const A &a) {
  if (shm > 64 * 1024)                               // above the default dynamic-LDS limit
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (pr.start) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, st, pr.start, pr.stop, 0, a);
  else hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, st, a);
  return hipGetLastError();
}
// geometry x stream-group count; two groups (5..8 streams) only in the 12-wave geometries
#define PF5(KERN, TP, MC, W, NG_)                                                                               \
  if (g.waves == W && g.tpw == TP && g.maxc == MC && ng == NG_) {                                               \
    if (g.pcell == 1) return plaunch(KERN<TP, MC, W, 1, NG_>, grid, W * 64, shm, st, pr, a);                    \
    if (g.pcell == 2) return plaunch(KERN<TP, MC, W, 2, NG_>, grid, W * 64, shm, st, pr, a);                    \
    if (g.pcell == 3) return plaunch(KERN<TP, MC, W, 3, NG_>, grid, W * 64, shm, st, pr, a);                    \
    return plaunch(KERN<TP, MC, W, 4, NG_>, grid, W * 64, shm, st, pr, a);                                      \
  }
#define PFX(KERN, TP, MC, NG_)                                                                                  \
  if (xbat && g.waves == 12 && g.tpw == TP && g.maxc == MC && ng == NG_) {                                      \
    if (g.pcell == 1) return plaunch(KERN<TP, MC, 12, 1, NG_, true>, grid, 768, shm, st, pr, a);                \
    if (g.pcell == 2) return plaunch(KERN<TP, MC, 12, 2, NG_, true>, grid, 768, shm, st, pr, a);                \
    if (g.pcell == 3) return plaunch(KERN<TP, MC, 12, 3, NG_, true>, grid, 768, shm, st, pr, a);                \
    return plaunch(KERN<TP, MC, 12, 4, NG_, true>, grid, 768, shm, st, pr, a);                                  \
  }
#define PF5I(KERN, TP, MC)                                                                                      \
  if (il && !xbat && g.waves == 12 && g.tpw == TP && g.maxc == MC && ng == 2) {                                 \
    if (g.pcell == 1) return plaunch(KERN<TP, MC, 12, 1, 2, false, true>, grid, 768, shm, st, pr, a);           \
    if (g.pcell == 2) return plaunch(KERN<TP, MC, 12, 2, 2, false, true>, grid, 768, shm, st, pr, a);           \
    if (g.pcell == 3) return plaunch(KERN<TP, MC, 12, 3, 2, false, true>, grid, 768, shm, st, pr, a);           \
    return plaunch(KERN<TP, MC, 12, 4, 2, false, true>, grid, 768, shm, st, pr, a);                             \
  }
#define PFXI(KERN, TP, MC)                                                                                      \
  if (il && xbat && g.waves == 12 && g.tpw == TP && g.maxc == MC && ng == 2) {                                  \
    if (g.pcell == 1) return plaunch(KERN<TP, MC, 12, 1, 2, true, true>, grid, 768, shm, st, pr, a);            \
    if (g.pcell == 2) return plaunch(KERN<TP, MC, 12, 2, 2, true, true>, grid, 768, shm, st, pr, a);            \
    if (g.pcell == 3) return plaunch(KERN<TP, MC, 12, 3, 2, true, true>, grid, 768, shm, st, pr, a);            \
    return plaunch(KERN<TP, MC, 12, 4, 2, true, true>, grid, 768, shm, st, pr, a);                              \
  }
#define PF5IN(KERN, MC, NG_)        /* 9..16 streams: three / four interleaved chains (12 waves, one tile per workgroup) */ \
  if (il && !xbat && g.waves == 12 && g.tpw == 1 && g.maxc == MC && ng == NG_) {                                \
    if (g.pcell == 1) return plaunch(KERN<1, MC, 12, 1, NG_, false, true>, grid, 768, shm, st, pr, a);          \
    if (g.pcell == 2) return plaunch(KERN<1, MC, 12, 2, NG_, false, true>, grid, 768, shm, st, pr, a);          \
    if (g.pcell == 3) return plaunch(KERN<1, MC, 12, 3, NG_, false, true>, grid, 768, shm, st, pr, a);          \
    return plaunch(KERN<1, MC, 12, 4, NG_, false, true>, grid, 768, shm, st, pr, a);                            \
  }
#define PDISPATCH_FWD(KERN)                                                                                     \
  do {                                                                                                          \
    PF5IN(KERN, 7, 3) PF5IN(KERN, 7, 4)                                                                          \
    if (ng > 2) return hipErrorInvalidValue;                                                                     \
    PFXI(KERN, 1, 7) PFXI(KERN, 1, 9) PF5I(KERN, 1, 7) PF5I(KERN, 1, 9) PF5I(KERN, 1, 12)                        \
    PFX(KERN, 1, 7, 1) PFX(KERN, 1, 9, 1) PFX(KERN, 2, 7, 1) PFX(KERN, 2, 9, 1)                                  \
    PFX(KERN, 1, 7, 2) PFX(KERN, 1, 9, 2) PFX(KERN, 2, 7, 2) PFX(KERN, 2, 9, 2)                                  \
    if (xbat) return hipErrorInvalidValue;                                                                       \
    PF5(KERN, 1, 7, 8, 1) PF5(KERN, 1, 9, 8, 1) PF5(KERN, 1, 12, 8, 1)                                          \
    PF5(KERN, 1, 7, 12, 1) PF5(KERN, 1, 9, 12, 1) PF5(KERN, 1, 12, 12, 1) PF5(KERN, 2, 7, 12, 1) PF5(KERN, 2, 9, 12, 1) PF5(KERN, 2, 12, 12, 1) \
    PF5(KERN, 1, 7, 16, 1) PF5(KERN, 1, 9, 16, 1) PF5(KERN, 1, 12, 16, 1) PF5(KERN, 2, 7, 16, 1) PF5(KERN, 2, 9, 16, 1) PF5(KERN, 2, 12, 16, 1) \
    PF5(KERN, 1, 7, 12, 2) PF5(KERN, 1, 9, 12, 2) PF5(KERN, 1, 12, 12, 2) PF5(KERN, 2, 7, 12, 2) PF5(KERN, 2, 9, 12, 2) PF5(KERN, 2, 12, 12, 2) \
    return hipErrorInvalidValue;                                                                                \
  } while (0)

// r(t) = W_r_m m(t) inside the forward launch: 4 rows of W_r_m per workgroup on its projection wave
bool persist_r_in_kernel(const Dims &d, const PersistOpts &o) {
  const int Ik = persist_x_batched(d) ? 0 : d.I;
  const PGeo g = pick_geo_fwd(o, d.C, pcdiv(d.C, KCH) + pcdiv(Ik, KCH), pcdiv(d.R, KCH) * KCH + Ik);
  return g.tpw > 0 && d.R % 4 == 0 && d.R / 4 <= d.C / 4 / g.tpw;
}

hipError_t launch_fwd_persist(const Dims &d, const FwdPtrs &p, const float *in, int in_stride, float *out, int out_stride,
                              unsigned long long *gran, unsigned *ctrl, const PersistOpts &o, hipStream_t st, LaunchProbe pr) {
  PersistFwdArgs a;
  const bool xbat = persist_x_batched(d);            // (the caller has run the batched x-projection into the gifo plane)
  a.C = d.C; a.I = xbat ? 0 : d.I; a.R = d.R; a.S = d.S; a.T = d.T;
  a.xg = xbat ? p.gifo : nullptr;
  a.nchm = pcdiv(d.C, KCH); a.nch = a.nchm + pcdiv(d.I, KCH);   // (stride of the packed operand: the fold product writes the layer's full row)
  const int nch_k = a.nchm + pcdiv(a.I, KCH);                      // chunks this launch contracts
  a.wpk = p.pk_fold; a.wr = p.wr; a.wx = p.wx; a.wm = p.wm; a.bias = p.bias; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.mm = p.mm; a.rr = p.rr;
  a.x = in; a.x_stride = in_stride; a.prev_c = p.prev_c; a.prev_r = p.prev_r; a.next_c = p.next_c; a.next_r = p.next_r; a.gran = gran; a.ctrl = ctrl;
  a.guard = o.guard;
  a.rin = out && persist_r_in_kernel(d, o); a.out = out; a.out_stride = out_stride;
  const bool il = d.S > 4 && o.fwd_interleave != 0;  // 5..8 streams: the two groups as interleaved chains (tpw = 1 geometries)
  a.nap0 = o.nap0 >= 0 ? o.nap0 : il ? 0 : d.S > 4 ? 4 : 2; a.nap = o.nap >= 0 ? o.nap : 0;     // (behind the publish flag; measured: tools/persist_anatomy, tools/nap_sweep.py)
  a.spin_limit = o.spin_limit > 0 ? o.spin_limit : SPIN_LIMIT_DEFAULT;
  a.test_stall = o.test_stall_fwd;
  a.hstat = o.hstat;
#ifdef KLSTM_PERSIST_TIMING
  a.dbg = o.dbg;
#endif
  const PGeo g = pick_geo_fwd(o, d.C, nch_k, pcdiv(d.R, KCH) * KCH + a.I);
  if (!g.tpw || !p.pk_fold || (!xbat && ((reinterpret_cast<uintptr_t>(in) & 15) || in_stride % 4 != 0))) return hipErrorInvalidValue;
  const int ng = (d.S + 3) / 4;
  const size_t shm = (size_t)(4 * ng * (g.maxc * 128 + 16) + 4 * ng * (persist_maxu(g.maxc) * 128 + 16) + 4) * sizeof(float);   // (+ abort flag, projection flag)
  const int grid = d.C / 4 / g.tpw;
  PDISPATCH_FWD(k_fwd_persist);
}

}  // namespace klstm
