// kaldi-lstm_amd/csrc/klstm_persist_bwd.hip -- weights-RESIDENT truncated-BPTT chain (engine option "persist"), 1..8 streams:
// ONE launch runs steps T..1 of the folded recurrence
//     d_m(t-1) = P(t-1) + dgifo(t) W_rm,   P = out_diff W_r_m          (...streams.h:391 substituted into :408)
//     dgifo(t) = elementwise BPTT of d_m(t), the carry of frame t+1 and the forward planes of frame t   (:411-440)
// with the per-step all-to-all of d_m (S x C floats) inside the launch: data-tagged 8-byte granules {tag, fp32}, one sc1
// store per (cell, stream) by the lane that owns it, swept with 16-byte sc1 loads until every tag matches (the transport of
// klstm_persist.hip; cdna_hip_programming.md Guideline 16 recipe R2).
//
// What changed against the round-2 kernel (sweepers -> LDS slab -> barrier -> 4 K waves -> barrier -> owner; 3.4-3.6 us per
// step, MI355X, 40/800/512, 4 streams): the waves that RECEIVE d_m(t) are the waves that contract it.  A sweep-and-contract
// (SC) wave owns a fixed set of cells; its lanes hold W_rm^T[own output cell i][k = (gate, cell)] for those cells in MFMA
// A-operand order, so what the sweep delivers goes through the elementwise pass in registers and straight into
// v_mfma_f32_4x4x1_16b as the B operand: no operand slab, no workgroup barrier, no idle contraction waves while the others
// sweep.  The cross-wave part of the contraction (every SC wave holds a slice of K) is a 16-byte partial per wave and
// stream in LDS plus an LDS counter; the OWNER wave adds the partials in fixed order, adds P(t-1) and publishes d_m(t-1).
//   per 32-cell "slot" of an SC wave, lane = (cell lane & 31, stream pair h = lane >> 5):
//     sweep     one 16-byte sc1 load = the granules of (cell, streams 2h, 2h+1)
//     planes    g, i, f, o, h of frame t and c of frame t-1 for the lane's two (cell, stream) pairs, requested before the
//               sweep (L2 hits; every workgroup reads the same rows; 128 contiguous bytes per half wave and instruction),
//               folded into six coefficients per pair while d_m flies
//     apply     d_c = d_m k1 + carry; d(g,i,f) = d_c (bg,bi,bf); d_o = d_m ao; carry' = d_c q       (5 operations per pair)
//     transpose the 8 dgifo values through a wave-private LDS tile into B-operand order (k-group lane >> 2, stream lane & 3)
//     contract  4 MFMAs per 16 cells (one per gate): A = resident weights, B = dgifo, 16 cells per MFMA
//   d_r(t-1) = out_diff(t-1) + dgifo(t) W_gifo_r (:391) and in_diff(t) = dgifo(t) W_gifo_x (:457) are four more output columns
//   per workgroup for the same B operand: their weights sit in LDS (read after the d_m partial has been signalled, off the
//   chain), partials go to a second LDS array and the P wave finishes them.
// Only d_m travels; dgifo is recomputed by every workgroup (replicas are bit-identical: same instruction sequence on the
// same inputs).  Shipping dgifo instead (each workgroup computing only its own cells) would quadruple the swept bytes
// (S x 4C values, 68-77 KB per workgroup and pass even in 16-byte {tag, 3 x fp32} granules) for no shorter chain: after the
// sweep a replicated pair costs five multiply-adds.
// 5..8 streams: two groups of 4 against the same resident weights, each with its own granule slots -- as two INTERLEAVED chains
// (k_bwd_persist2i below: 101 us per launch at 8 streams) or, option "persist_bwd_interleave" = 0, one chain after the other
// (k_bwd_persist2: 129 us).
// Wave roles: wave 0 = owner (combine, publish, dgifo / dc plane rows of the own cells), wave 1 = P wave (own columns of
// P = out_diff W_r_m ahead of the chain; finishes d_r / in_diff), waves 2.. = SC waves.  Nothing global is ordered by
// anything but the granule tags; every wait is bounded (per wait, wall clock).
#include "klstm_kernels.h"
#include "klstm_math.h"
#include "klstm_persist_dev.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>

namespace klstm {

#pragma clang fp contract(off)

// Granule ring of the backward chains: frame t of a group lives in slot t % BWD_RING.  The chain itself needs 2 (a workgroup publishes
// frame t - 1 only after it has seen all of frame t); the tail workgroups read the same granules WITHOUT being waited for -- they start
// late (their resident rows are 256 KB per workgroup) and catch up -- so a slot has to outlive that: 32 slots = they may be 30 frames
// behind before a frame is lost (which they notice: bwd_tail_role).  1.6 MB per engine at C = 800.
constexpr int BWD_RING = 32;

struct PersistBwd2Args {
  int C, R, S, T, I;
  int pin;                        // 1: P = out_diff W_r_m is computed here (own columns, kept in LDS); 0: read from P
  int din;                        // bit 0: d_r(1..T) contracted here, bit 1: in_diff too (4 columns per workgroup)
  const float *od; int od_stride; // out_diff rows [T*S x R]
  const float *wmT;               // W_r_m^T [C x R]
  const float *wrT, *wxT;         // W_gifo_r^T [R x 4C], W_gifo_x^T [I x 4C] (d_r / in_diff on the chain's workgroups)
  const float *wrN, *wxN;         // W_gifo_r [4C x R], W_gifo_x [4C x I] as the parameter blob holds them (the tail workgroups)
  float *dr;                      // d_r plane [(T+2)*S x R], time-major row blocks
  float *in_diff; int id_stride;  // [T*S x I]
  int nch1;                       // 32-wide chunks per row tile of wpk (the m part and the x part of the folded gates operand)
  const float *wpk;               // the folded GATES operand (klstm_fold.hip pk1: [C/4 row tiles][nch1][2][64] float4, float4 = 4 consecutive k of
                                  // row 4 cell + gate): this workgroup's 4 columns of W_rm are ONE float4 per row, 16 rows = 256 contiguous bytes
  const float *pi, *pf, *po;
  const float *gifo, *cc, *hh;
  float *dgifo, *dc;
  const float *P;                 // out_diff * W_r_m [T*S x C] (pin == 0)
  unsigned long long *gran;       // [stream groups][BWD_RING][C*4] granules, cell-major (4 stream slots per cell); frame t lives in ring slot t % BWD_RING
  int tqpp;                       // ... column quads per part (a multiple of 4)
  int tw0;                        // ... 1: wave 0 prepares only (the others contract: NW - 1 waves x TAIL_TG groups of 4 quads hold a part); 0: it contracts too
  int tq, tparts;                 // tail workgroups (blockIdx >= C/4; d_r / in_diff off the chain, see bwd_tail_role): column quads in total
                                  // (R/4 of d_r, then I/4 of in_diff; 0: no tail workgroups) and workgroups per 32-cell slot (column parts)
  float *tws;                     // their partial rows [C/32 slots][T*S][4 tq]: k_tail_reduce adds them behind the launch
  unsigned *ctrl;                 // [0] epoch, [1] finished workgroups, [2] status (0 = ok)
  int nap0, nap;                  // SC waves sleep nap0 x 256 clocks before the first pass of a step, nap x 64 between passes
  long long spin_limit;           // wall-clock ticks (100 MHz) a single wait may take
  int test_stall;                 // test hook: workgroup 0 does not publish d_m(test_stall) (0: never)
  unsigned *hstat;                // host-mapped status word (or null): set when a wait expires, read by the engine without a sync
  unsigned *guard;                // the engine's control words (or null): [2] / [6] set by an earlier launch -> do nothing; [8] = persistent
                                  // launches of this engine that have run so far: this launch's ordinal, [8] + 1, goes into ctrl[3] if it gives up
#ifdef KLSTM_PERSIST_TIMING
  long long *dbg;
#endif
};

// The elementwise BPTT of a (cell, stream) pair (:411-440) is linear in d_m(t); everything else is known before d_m(t) has
// crossed the fabric.  Six coefficients per pair from the forward planes of frame t:
//   d_h = d_m [yo(1-yh^2)]   d_o = d_m [yh yo(1-yo)]   d_c = d_m k1 + carry,  k1 = yo(1-yh^2) + wpo ao       (:411-428)
//   d_f = d_c [c(t-1) yf(1-yf)]   d_i = d_c [yg yi(1-yi)]   d_g = d_c [yi(1-yg^2)]                            (:431-440)
//   carry(t-1) = d_c(t) f(t) + d_i(t) wpi + d_f(t) wpf = d_c(t) q(t),  q = yf + wpi bi + wpf bf               (:425-427 of frame t-1)
// (same algebra as the reference, products associated differently: a few ulp; identical in every workgroup)
struct Bptt2Coef { float k1, ao, q, bg, bi, bf; };
__device__ __forceinline__ Bptt2Coef bptt2_coef(float yg, float yi, float yf, float yo, float yh, float cpv, float wpi, float wpf,
                                                float wpo) {
  Bptt2Coef c;
  const float ah = __builtin_fmaf(-yo, yh * yh, yo);
  c.ao = yh * __builtin_fmaf(-yo, yo, yo);
  c.k1 = __builtin_fmaf(wpo, c.ao, ah);
  c.bf = cpv * __builtin_fmaf(-yf, yf, yf);
  c.bi = yg * __builtin_fmaf(-yi, yi, yi);
  c.bg = __builtin_fmaf(-yi, yg * yg, yi);
  c.q = __builtin_fmaf(wpf, c.bf, __builtin_fmaf(wpi, c.bi, yf));
  return c;
}
// d(g, i, f, o) of frame t; updates the carry
__device__ __forceinline__ float4 bptt2_apply(float dm, const Bptt2Coef &c, float &carry, float &d_c_out) {
  const float d_c = __builtin_fmaf(dm, c.k1, carry);
  carry = d_c * c.q;
  d_c_out = d_c;
  return make_float4(d_c * c.bg, d_c * c.bi, d_c * c.bf, dm * c.ao);
}

__device__ __forceinline__ float dpp_quad(unsigned v, bool odd_pair) {
  // lane 4b + j reads lane 4b + 2*(j >> 1) (+ 1): quad_perm [0,0,2,2] / [1,1,3,3]
  return odd_pair ? __int_as_float(__builtin_amdgcn_update_dpp(0, (int)v, 0xF5, 0xf, 0xf, true))
                  : __int_as_float(__builtin_amdgcn_update_dpp(0, (int)v, 0xA0, 0xf, 0xf, true));
}


// -------------------------------------------------------------------------------------------------------------------
// TAIL WORKGROUPS (round 6): d_r(t-1) = out_diff(t-1) + dgifo(t) W_gifo_r (:391) and in_diff(t) = dgifo(t) W_gifo_x (:457) OFF the chain.
// Until round 5 the first (R + I) / 4 chain workgroups carried 4 of these output columns each: a second contraction of the same
// dgifo(t) per step, in the loop that sets the pace of the whole chain (69.0 vs 61.0 us per launch at 4 streams, 101.5 vs 83.5 at 8;
// tools/ab_step.py, "persist_tail" = 0 vs 2).  A chain of C / 4 = 200 workgroups leaves 56 of the 256 compute units idle: workgroups
// C / 4 .. of the SAME launch land there.
//   * First form (measured, dropped: profiles/r06_tail_wg_anatomy.txt): every tail workgroup sweeps ALL of d_m(t) like a chain
//     workgroup, recomputes dgifo(t) of all cells and owns a few output columns.  46 more compute units pulling the 77 KB of planes per
//     step slowed the CHAIN's plane loads (1.53 -> 1.91 us) and a tail step took 3.96 us against the chain's 2.84: slower than rounds 3-5.
//   * This form splits K instead: tail workgroup j owns the 32-cell slot j -- the rows (gate, cell) of W_gifo_r^T / W_gifo_x^T of ITS
//     cells for ALL output columns, resident in registers (72 per lane at 552 columns) -- sweeps the 1 KB of d_m granules of its cells,
//     pulls the planes of its cells (1.5 KB instead of 77 KB per step), applies the elementwise BPTT exactly as the chain does and
//     writes its PARTIAL d_r / in_diff rows of frame t to a workspace; k_tail_reduce adds the C / 32 partials in slot order behind the
//     launch (fixed order: deterministic).  Planes and granules are read once in total, the chain is not disturbed.
//   * Nobody waits for a tail workgroup, so it may fall behind (a slow start, a co-tenant on its compute unit).  The granule ring has
//     BWD_RING = 32 slots; a sweep that finds a NEWER frame of this launch in its slot (tag in [tag0 + 1, expected)) has lost its frame:
//     the workgroup gives up like any expired wait (status word; the engine runs the minibatch again on the launch-per-step chain).
// Wave 0 sweeps and applies (lane = (cell, stream pair), the SC waves' natural layout) and leaves dgifo(t) of the slot in LDS in MFMA
// B-operand order; after ONE workgroup barrier every wave contracts it against its TQW column quads (4x4x1 geometry: 16 k per MFMA, 8
// MFMAs per quad) and lanes 12..15 store 16 bytes per quad and stream.  Two LDS buffers: wave 0 prepares frame t - 1 while the others
// still contract frame t.
// IL: the (frame, group) order of the interleaved kernel (t, 0), (t, 1), (t - 1, 0), ...; otherwise group after group.
// -------------------------------------------------------------------------------------------------------------------
// (tail_reduce_outputs: klstm_persist_dev.h -- the gradient launch runs it too, "tail_merge")
constexpr int TAIL_TG = 2;       // groups of 4 column quads per wave (32 weight registers each)

template <int NW, bool IL, int NGI = 2>
__device__ __forceinline__ void bwd_tail_role(const PersistBwd2Args &a, float *lds, unsigned epoch, unsigned behind_giveup) {
  unsigned *abortf = reinterpret_cast<unsigned *>(lds);
  float *xt2 = lds + 16;                             // [2][4 gates][32 cells][4 streams]
  const int C = a.C, S = a.S, T = a.T, K = 4 * a.C, R = a.R;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ngrp = (S + 3) >> 2;
  const int tb = (int)blockIdx.x - C / 4;
  const int slot = tb / a.tparts, part = tb % a.tparts;      // the 32-cell slot; which a.tqpp column quads of it
  const int nq = a.tq, ngr = R / 4, ncols = 4 * nq;          // quads 0 .. ngr - 1: d_r columns; then in_diff columns
  const int q_lo = part * a.tqpp, q_hi = q_lo + a.tqpp < nq ? q_lo + a.tqpp : nq;
  const int tlo = nq > ngr ? 1 : 2;                  // frame 1 is swept only where in_diff(1) is contracted
  const int nfr = T - tlo + 1, nsteps = nfr * ngrp;
  if (__builtin_amdgcn_readfirstlane(behind_giveup) != 0u) return;
  auto frame_of = [&](int n) { return IL ? T - n / NGI : T - n % nfr; };
  auto group_of = [&](int n) { return IL ? n % NGI : n / nfr; };

  // ---- resident rows.  Geometry of v_mfma_f32_4x4x1_16b here: block bb = lane >> 2 = (column quad qq = bb >> 2 of the group, k-group kk = bb & 3),
  // MFMA m contracts k = 4 m + kk (k = 32 e + cell of the slot: 32 MFMAs for the 128 rows), A lane 4 bb + i = W^T[column 4 cq + i][k],
  // B lane 4 bb + j = dgifo[k][stream j].  Only FOUR k-groups share an output, and they sit in one row of 16 lanes: two DPP row shifts
  // finish the sum (16 k-groups per output -- the chain's layout -- needs two more cross-row stages per value: 36 values x 4 stages per
  // wave and step made the contraction VALU-bound, 2.8 us per step; profiles/r06_tail_wg_anatomy.txt) ----
  // Read from the NATURAL matrices (W[k][column]: the four columns of a quad are 16 contiguous bytes, the quads of a group 64): the
  // transposed copies have no reader then and the Update leaves them out (klstm_engine.hip "wT32_stale").
  // Where the part's groups fit NW - 1 waves, wave 0 only prepares (a.tw0: the elementwise side of frame t - 1 runs under the others'
  // contraction of frame t); group G of a part goes to contracting wave G % ncw.
  const int bb = lane >> 2, i4 = lane & 3, qq = bb >> 2, kk = bb & 3;
  const int ncw = NW - a.tw0, cwave = wave - a.tw0;  // contracting waves; this wave's rank among them
  const bool contracts = cwave >= 0;
  float wq[TAIL_TG][32];
  int cqg[TAIL_TG];
#pragma unroll
  for (int gq = 0; gq < TAIL_TG; gq++) {
    const int cq = q_lo + 4 * (cwave + ncw * gq) + qq;   // this lane's column quad of the wave's group gq
    cqg[gq] = cq;
    const bool qv = contracts && cq < q_hi;
    const bool isr = cq < ngr;
    const int ld = isr ? R : a.I;
    const float *src = !qv ? a.wrN : isr ? a.wrN + 4 * cq + i4 : a.wxN + 4 * (cq - ngr) + i4;
#pragma unroll
    for (int m = 0; m < 32; m++) {
      const int cell = 32 * slot + 4 * (m & 7) + kk;
      const float v = src[(size_t)((m >> 3) * C + (cell < C ? cell : 0)) * (qv ? ld : 0)];
      wq[gq][m] = qv && cell < C ? v : 0.f;
    }
  }
  const bool two_groups = contracts && q_lo + 4 * (cwave + ncw) < q_hi;     // (wave-uniform: the second group of this wave exists)
  float *pslot = a.tws + (size_t)slot * T * S * ncols;       // this slot's partial rows [T * S][ncols]

  // ---- wave 0: the elementwise side (natural layout: cell c32 = lane & 31, stream pair h = lane >> 5) ----
  const int c32 = lane & 31, h = lane >> 5;
  const int cl = 32 * slot + c32;
  const bool live = cl < C;
  const int clc = live ? cl : 0;
  const int svoff = clc * 32 + h * 16;
  const int voffG = (2 * h * K + clc) * 4, voffC = (2 * h * C + clc) * 4;
  float wpi = 0.f, wpf = 0.f, wpo = 0.f;
  if (wave == 0) { wpi = a.pi[clc]; wpf = a.pf[clc]; wpo = a.po[clc]; }
  const __amdgpu_buffer_rsrc_t rs_g = buf_rsrc(a.gifo, (T + 2) * S * K * 4), rs_h = buf_rsrc(a.hh, (T + 2) * S * C * 4);
  const __amdgpu_buffer_rsrc_t rs_c = buf_rsrc(a.cc, (T + 2) * S * C * 4);
  const __amdgpu_buffer_rsrc_t rs_gr = buf_rsrc(a.gran, ngrp * BWD_RING * C * 32);
  float carry[IL ? NGI : 1][2];
#pragma unroll
  for (int gc = 0; gc < (IL ? NGI : 1); gc++) { carry[gc][0] = 0.f; carry[gc][1] = 0.f; }
  PT_DECL();

  // the elementwise side of one step: planes of the slot's cells, sweep of their granules, dgifo(t) into xt (operand order)
  auto prepare = [&](auto GC, int t, int g, float *xt) -> bool {
    constexpr int gc = decltype(GC)::value;
    const int Sg = S - 4 * g < 4 ? S - 4 * g : 4;
    const bool need0 = 2 * h < Sg, need1 = 2 * h + 1 < Sg;
    const int sG = (t * S + 4 * g) * K * 4, sC = (t * S + 4 * g) * C * 4;
    Bptt2Coef cf[2];
#pragma unroll
    for (int x = 0; x < 2; x++) {                  // stream 2h + x of the group (absent streams read a later row or zeros: never used)
      const int oG = sG + x * K * 4, oC = sC + x * C * 4;
      const float yg = buf_f32(rs_g, voffG, oG), yi = buf_f32(rs_g, voffG, oG + C * 4);
      const float yf = buf_f32(rs_g, voffG, oG + 2 * C * 4), yo = buf_f32(rs_g, voffG, oG + 3 * C * 4);
      const float yh = buf_f32(rs_h, voffC, oC), cpv = buf_f32(rs_c, voffC, oC - S * C * 4);
      cf[x] = bptt2_coef(yg, yi, yf, yo, yh, cpv, wpi, wpf, wpo);
    }
    const unsigned tag0 = epoch + (unsigned)(g * (T + 2));
    const int soff = (g * BWD_RING + t % BWD_RING) * C * 32;
    u32x4 q;
    const long long t0 = wall_clock64();
    for (unsigned spins = 0;; spins++) {
      q = __builtin_amdgcn_raw_buffer_load_b128(rs_gr, svoff, soff, 16);   // aux 16 = sc1
      const unsigned d0 = q.y - tag0, d1 = q.w - tag0;                      // frame the granule carries (this launch's: 1 .. T)
#ifdef KLSTM_PERSIST_TIMING
      const bool ok = !live | (((!need0) | (d0 - 1u < (unsigned)t)) & ((!need1) | (d1 - 1u < (unsigned)t)));   // (tools/persist_anatomy: a lost frame counts as seen -- timing only)
      const bool lost = false;
#else
      const bool ok = !live | (((!need0) | (d0 == (unsigned)t)) & ((!need1) | (d1 == (unsigned)t)));
      const bool lost = live & ((need0 & (d0 - 1u < (unsigned)t - 1u)) | (need1 & (d1 - 1u < (unsigned)t - 1u)));   // a NEWER frame sits in the slot
#endif
      if (__all(ok)) break;
      if (__any(lost) || ((spins & 15) == 15 && wall_clock64() - t0 > a.spin_limit)) {
        __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0) { atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard)); atomicMax(&a.ctrl[2], 0x80000000u | 0x4000u | (unsigned)t); }
        return false;
      }
      for (int i = 0; i < a.nap; i++) __builtin_amdgcn_s_sleep(1);
    }
    PT_MARK(0);                                      // planes + sweep
    float dcv;
    const float4 d0 = bptt2_apply(live ? __uint_as_float(q.x) : 0.f, cf[0], carry[gc][0], dcv);
    const float4 d1 = bptt2_apply(live ? __uint_as_float(q.z) : 0.f, cf[1], carry[gc][1], dcv);
    float2 *xw = reinterpret_cast<float2 *>(xt + c32 * 4 + 2 * h);
    xw[0] = make_float2(d0.x, d1.x); xw[64] = make_float2(d0.y, d1.y);             // gate e at + e*128 floats
    xw[128] = make_float2(d0.z, d1.z); xw[192] = make_float2(d0.w, d1.w);
    PT_MARK(1);                                      // apply + tile
    return true;
  };
  // the contraction side: every wave, its groups of 4 column quads against the slot's dgifo(t)
  auto contract = [&](int t, int g, const float *xt) {
    const int Sg = S - 4 * g < 4 ? S - 4 * g : 4;
    f32x4 acc[TAIL_TG][2];
#pragma unroll
    for (int gq = 0; gq < TAIL_TG; gq++) { acc[gq][0] = (f32x4){0, 0, 0, 0}; acc[gq][1] = (f32x4){0, 0, 0, 0}; }
    const float *xb = xt + kk * 4 + i4;              // B lane 4 bb + j: dgifo[k = 4 m + kk][j]: [e = m >> 3][cell 4 (m & 7) + kk][j]
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {                 // (the B operand in two halves of 16 registers)
      float Bv[16];
#pragma unroll
      for (int mm = 0; mm < 16; mm++) { const int m = 16 * hh + mm; Bv[mm] = xb[(m >> 3) * 128 + (m & 7) * 16]; }
#pragma unroll
      for (int mm = 0; mm < 16; mm++)
        acc[0][mm & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[0][16 * hh + mm], Bv[mm], acc[0][mm & 1], 0, 0, 0);
      if (two_groups) {
#pragma unroll
        for (int mm = 0; mm < 16; mm++)
          acc[1][mm & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[1][16 * hh + mm], Bv[mm], acc[1][mm & 1], 0, 0, 0);
      }
    }
    float *orow = pslot + ((size_t)(t - 1) * S + 4 * g + i4) * ncols;
    const bool st_lane = kk == 3 && i4 < Sg;         // lanes 12..15 of every row of 16: (stream i4) of the row's quad
#pragma unroll
    for (int gq = 0; gq < TAIL_TG; gq++) {
      if (gq == 1 && !two_groups) break;
      const f32x4 v = acc[gq][0] + acc[gq][1];
      float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {                  // the row's four k-groups (klstm_persist_dev.h kgroup_sum, first half)
        c[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c[e]), 0x114, 0xf, 0xf, true));   // row_shr:4
        c[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c[e]), 0x118, 0xf, 0xf, true));   // row_shr:8
      }
      if (st_lane && cqg[gq] < q_hi) *reinterpret_cast<float4 *>(orow + 4 * cqg[gq]) = make_float4(c[0], c[1], c[2], c[3]);
    }
  };

  // Iteration n: wave 0 prepares step n into buffer n & 1 while the other waves contract step n - 1 out of the other buffer; ONE barrier per
  // iteration hands buffer n & 1 over and gives buffer (n - 1) & 1 back.
  for (int n = 0; n <= nsteps; n++) {
    PT_MARK(5);
    if (wave == 0) {
      if (n < nsteps) {
        const int t = frame_of(n), g = group_of(n);
        float *xt = xt2 + (n & 1) * 512;
        bool ok;
        if (IL) {                                    // (the carry set is a compile-time index: one call site per group)
          ok = g == 0 ? prepare(std::integral_constant<int, 0>(), t, g, xt)
             : g == 1 ? prepare(std::integral_constant<int, ((IL && (NGI > 1)) ? 1 : 0)>(), t, g, xt)
             : g == 2 ? prepare(std::integral_constant<int, ((IL && (NGI > 2)) ? 2 : 0)>(), t, g, xt)
                      : prepare(std::integral_constant<int, ((IL && (NGI > 3)) ? 3 : 0)>(), t, g, xt);
        } else {
          if (t == T) { carry[0][0] = 0.f; carry[0][1] = 0.f; }     // a new group's chain
          ok = prepare(std::integral_constant<int, 0>(), t, g, xt);
        }
        (void)ok;
      }
    }
    if (contracts && n > 0) {
      contract(frame_of(n - 1), group_of(n - 1), xt2 + ((n - 1) & 1) * 512);
      PT_MARK(3);                                    // contraction + partial rows
    }
    lds_barrier();
    PT_MARK(2);
    if (*abortf) break;
  }
  PT_FLUSH(0);
  // (Measured and dropped, round 6, both to save k_tail_reduce's 4.2 us behind the launch: (a) the tail workgroups adding the partial rows
  //  THEMSELVES behind an arrival counter -- agent-scope release of their rows, acquire before reading the others': 75.1 us per launch
  //  against 61.5 + 4.2 at 4 streams; the release / acquire pair writes back and invalidates L2 far beyond the 4.4 MB in question.
  //  (b) partial rows as {tag, value} granules (write-through) and REDUCER workgroups on the 6 compute units left, sweeping a frame's
  //  granules of all 25 slots and adding them in slot order inside the launch: correct, but 74 KB of sc1 loads per frame and reducer, polled
  //  until complete, take 5.5 us per frame -- 113.7 us per launch at 4 streams, 207 at 8.  (c) the same granules, but every tail workgroup's
  //  LAST wave adding a 1 / 50 share of the outputs of the step two back (9.6 KB of sc1 loads per step): a coherent load of another XCD's
  //  fresh write-through data takes ~2 us, two passes per step, and the wave sits in the workgroup's per-step barrier: 88.5 us per launch
  //  at 4 streams, 157.6 at 8.  (d) the reduction on the FIRST workgroups of the gradient launch that follows (option "tail_merge",
  //  k_grads_tm: write-through d_r, arrival counter, the W_r_m tiles wait and read with sc1 loads): 20.0 us against 4.3 + 14.4 at 4
  //  streams, 27.3 against 4.4 + 18.2 at 8 -- every gradient tile is resident from the start and latency-bound, the tiles that wait end
  //  one cross-XCD hand-over (~6 us) late.  All four are bit-correct; k_tail_reduce behind the launch stays.)
}

// d_r / in_diff from the tail workgroups' partial rows: the nslots partials of an output added in slot order (fixed order), out_diff
// added to the d_r columns (:391), d_r(T) = out_diff(T) (:351).  One thread per (frame row, column quad).
struct TailReduceArgs {
  TailReduceJob j;
  const unsigned *guard;          // a launch in front gave up: the partial rows are not there -- write nothing (everything is run again)
};
__global__ __launch_bounds__(256) void k_tail_reduce(TailReduceArgs a) {
  const int gidx = blockIdx.x * 256 + threadIdx.x;
  tail_reduce_outputs<false>(a.j.tws, a.j.nslots, a.j.T, a.j.S, a.j.R, a.j.ncols, a.j.od, a.j.od_stride, a.j.dr, a.j.in_diff, a.j.id_stride, gidx >> 3,
                             (int)gridDim.x * 32, gidx & 7, a.guard);
}

template <int NW, int NU>
__global__ __launch_bounds__(NW * 64) void k_bwd_persist2(PersistBwd2Args a) {
  constexpr int NSC = NW - 2, NSLOT = NSC * NU;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned *abortf = reinterpret_cast<unsigned *>(lds);
  int *pcnt = reinterpret_cast<int *>(lds) + 1;      // d_m partials written (one count per SC wave and step)
  int *dcnt = reinterpret_cast<int *>(lds) + 2;      // d_r / in_diff partials written
  int *dcons = reinterpret_cast<int *>(lds) + 3;     // d steps the P wave has consumed
  int *pdone = reinterpret_cast<int *>(lds) + 4;     // frames of P in LDS (T per stream group, descending frames)
  int *pubn = reinterpret_cast<int *>(lds) + 5;      // publishes the owner has issued (T per stream group)
  int *odone = reinterpret_cast<int *>(lds) + 6;     // stream groups the owner has finished
  f32x4 *red = reinterpret_cast<f32x4 *>(lds + 16);  // [NSC][4 streams]: components = the 4 own cells
  f32x4 *red2 = red + NSC * 4;                       // [2][NSC][4 streams]: components = the 4 d_r / in_diff columns
  float *xtile = reinterpret_cast<float *>(red2 + 2 * NSC * 4);   // [NSC][4 gates][32 cells][4 streams]: natural -> operand order
  float *wD = xtile + NSC * 512;                         // [NSLOT][2 sets][4 gates][64 lanes]: A operands of the d_r / in_diff columns
  f32x4 *ldsP = reinterpret_cast<f32x4 *>(wD + NSLOT * 512);     // [T][4 streams] (pin): components = the 4 own cells
  const int C = a.C, S = a.S, T = a.T, K = 4 * a.C;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ngrp = (S + 3) >> 2;
  const long long limit = a.spin_limit;
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // queued behind a launch that gave up (status word still set: the host has not looked yet): the planes this launch would read
  // are invalid -- do nothing.  Requested here, looked at by every role behind the loads of its own prologue (loads return in
  // order: by then the word is there, the look costs nothing; a branch up here cost 1.2 us per launch).  A role that skips
  // touches nothing global and takes no part in any LDS hand-shake; everybody meets again in finish().
  unsigned behind_giveup = 0u;
#ifndef KLSTM_NO_CHAIN_GUARD
  if (a.guard) behind_giveup = __hip_atomic_load(&a.guard[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                               __hip_atomic_load(&a.guard[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  // the d_r / in_diff columns of this workgroup: 4 rows of W_gifo_r^T (workgroups 0 .. R/4-1), then of W_gifo_x^T
  const int ngr = a.R / 4, ngx = (a.din & 2) ? a.I / 4 : 0;
  const bool d_on = a.din && (int)blockIdx.x < ngr + ngx, d_isr = (int)blockIdx.x < ngr;
  const int dcol = d_isr ? (int)blockIdx.x * 4 : ((int)blockIdx.x - ngr) * 4;
  if (tid < 16) reinterpret_cast<int *>(lds)[tid] = 0;
  __syncthreads();
  PT_DECL();

  if (a.tq && (int)blockIdx.x >= C / 4) {
    bwd_tail_role<NW, false>(a, lds, epoch, behind_giveup);          // a tail workgroup: d_r / in_diff partials of its 32 cells, off the chain
  } else if (wave == 0) {
    // =========================== owner: combine, publish, own plane rows ===========================
    // lanes 0..15 = (own cell oi = lane >> 2, stream oj = lane & 3)
    const int oi = (lane >> 2) & 3, oj = lane & 3;
    const int ocell = (int)blockIdx.x * 4 + oi;
    const float wpi = a.pi[ocell], wpf = a.pf[ocell], wpo = a.po[ocell];
    const float *redf = reinterpret_cast<const float *>(red);
    int np = 0;
    bool dead = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
    for (int g = 0; g < ngrp && !dead; g++) {
      const int Sg = S - 4 * g < 4 ? S - 4 * g : 4;
      const bool on = lane < 16 && oj < Sg;
      const int srow = 4 * g + (oj < Sg ? oj : 0);
      unsigned long long *gr = a.gran + (size_t)g * BWD_RING * C * 4;
      const unsigned tag0 = epoch + (unsigned)(g * (T + 2));
      if (on) {                                      // the batched d_r product (tail outside) reads dgifo(T+1) as operand rows: zero (:351)
        float *zp = a.dgifo + ((size_t)(T + 1) * S + srow) * K + ocell;
        zp[0] = 0.f; zp[C] = 0.f; zp[2 * C] = 0.f; zp[3 * C] = 0.f;
      }
      // d_m(T) = P(T): dgifo(T+1) = 0 (:351, :391) -- travels like every other step
      float dmv;
      if (a.pin) {
        if (!lds_wait_ge(pdone, g * T + 1, abortf, limit)) { dead = true; break; }
        dmv = reinterpret_cast<const float *>(&ldsP[(T - 1) * 4 + oj])[oi];
      } else {
        dmv = a.P[((size_t)(T - 1) * S + srow) * C + ocell];
      }
      if (on && !(a.test_stall == T && blockIdx.x == 0)) publish(gr + (size_t)(T % BWD_RING) * C * 4, ocell * 4 + oj, tag0 + (unsigned)T, dmv);
      __hip_atomic_store(pubn, g * T + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      float carry = 0.f;
      float yg, yi, yf, yo, yh, cpv;
      auto load_planes = [&](int t) {                // the own pair's planes of frame t: consumed one publish later
        const float *gp = a.gifo + ((size_t)t * S + srow) * K + ocell;
        yg = gp[0]; yi = gp[C]; yf = gp[2 * C]; yo = gp[3 * C];
        yh = a.hh[((size_t)t * S + srow) * C + ocell];
        cpv = a.cc[((size_t)(t - 1) * S + srow) * C + ocell];
      };
      auto own_rows = [&](int t, float dm) {         // dgifo / dc rows of the own cells (the gradient products read them)
        const Bptt2Coef cf = bptt2_coef(yg, yi, yf, yo, yh, cpv, wpi, wpf, wpo);
        float dcv;
        const float4 d = bptt2_apply(dm, cf, carry, dcv);
        if (on) {
          float *dp = a.dgifo + ((size_t)t * S + srow) * K + ocell;
          dp[0] = d.x; dp[C] = d.y; dp[2 * C] = d.z; dp[3 * C] = d.w;
          a.dc[((size_t)t * S + srow) * C + ocell] = dcv;
        }
      };
      load_planes(T);
      for (int t = T; t >= 2; t--) {
        PT_MARK(5);
        float pnext = 0.f;
        if (!a.pin) pnext = a.P[((size_t)(t - 2) * S + srow) * C + ocell];
        ++np;
        if (!lds_wait_ge(pcnt, NSC * np, abortf, limit)) { dead = true; break; }
        PT_MARK(0);                                  // waiting for the partials of step t
        float part[NSC];
#pragma unroll
        for (int w = 0; w < NSC; w++) part[w] = redf[(w * 4 + oj) * 4 + oi];
        if (a.pin) {
          if (!lds_wait_ge(pdone, g * T + (T - t + 2), abortf, limit)) { dead = true; break; }
          pnext = reinterpret_cast<const float *>(&ldsP[(t - 2) * 4 + oj])[oi];
        }
        float sum = part[0];
#pragma unroll
        for (int w = 1; w < NSC; w++) sum += part[w];  // fixed order
        const float dmn = sum + pnext;               // :408 with :391 substituted: d_m(t-1) = contraction + P(t-1)
        if (on && !(a.test_stall == t - 1 && blockIdx.x == 0))
          publish(gr + (size_t)((t - 1) % BWD_RING) * C * 4, ocell * 4 + oj, tag0 + (unsigned)(t - 1), dmn);
        __hip_atomic_store(pubn, g * T + (T - t + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        PT_MARK(1);                                  // combine + publish
        own_rows(t, dmv);
        dmv = dmn;
        load_planes(t - 1);
        PT_MARK(2);
      }
      if (dead) break;
      own_rows(1, dmv);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __hip_atomic_store(odone, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    PT_FLUSH(0);
  } else if (wave == 1) {
    // =========================== P wave: own columns of P = out_diff W_r_m; finishes d_r / in_diff ===========================
    // 4-row geometry: A = rows of W_r_m^T (the 4 own cells; R <= 512 = 4 chunks, resident), B = the 4 stream rows of out_diff
    // of one frame; lanes 12..15 end up with P[frame][stream lane & 3][cells 0..3].  Runs ahead of the chain.
    const int kg = lane >> 2, bj = lane & 3, R = a.R;
    float4 w0[4], w1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int k = 128 * i + 4 * kg, pc = (int)blockIdx.x * 4 + bj;
      w0[i] = a.pin && k < R ? *reinterpret_cast<const float4 *>(a.wmT + (size_t)pc * R + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      w1[i] = a.pin && k + 64 < R ? *reinterpret_cast<const float4 *>(a.wmT + (size_t)pc * R + k + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float *red2f = reinterpret_cast<const float *>(red2);
    const int fi = (lane >> 2) & 3, fj = lane & 3;   // finishing lanes 0..15 = (column fi, stream fj)
    int nd = 0;
    bool dead = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
    for (int g = 0; g < ngrp && !dead; g++) {
      const int Sg = S - 4 * g < 4 ? S - 4 * g : 4;
      if (g > 0 && a.pin && !lds_wait_ge(odone, g, abortf, limit)) break;   // (the P rows of the previous group are still being read)
      int nextf = T - 1;                             // next frame of P (descending)
      auto p_until = [&](int flo) {
        while (nextf >= flo && nextf >= 0) {
          const int f = nextf;
          const bool rv = bj < Sg;
          const float *op = a.od + ((size_t)f * S + 4 * g + (rv ? bj : 0)) * a.od_stride + 4 * kg;
          float4 b0[4], b1[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int k = 128 * i + 4 * kg;
            b0[i] = rv && k < R ? *reinterpret_cast<const float4 *>(op + 128 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            b1[i] = rv && k + 64 < R ? *reinterpret_cast<const float4 *>(op + 128 * i + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float av[8] = {w0[i].x, w0[i].y, w0[i].z, w0[i].w, w1[i].x, w1[i].y, w1[i].z, w1[i].w};
            const float bv[8] = {b0[i].x, b0[i].y, b0[i].z, b0[i].w, b1[i].x, b1[i].y, b1[i].z, b1[i].w};
#pragma unroll
            for (int jj = 0; jj < 8; jj++) acc[jj & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[jj], bv[jj], acc[jj & 3], 0, 0, 0);
          }
          const f32x4 v = kgroup_sum_pl((acc[0] + acc[1]) + (acc[2] + acc[3]));
          if (lane >= 12 && lane < 16) ldsP[f * 4 + bj] = v;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __hip_atomic_store(pdone, g * T + (T - f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          nextf--;
        }
      };
      // d_r(T) = out_diff(T): dgifo(T+1) = 0 (:351, :391)
      if (d_on && d_isr && lane < 16 && fj < Sg)
        a.dr[((size_t)T * S + 4 * g + fj) * R + dcol + fi] = a.od[((size_t)(T - 1) * S + 4 * g + fj) * a.od_stride + dcol + fi];
      if (a.pin) p_until(T - 2);
      if (d_on) {
        for (int t = T; t >= (d_isr ? 2 : 1); t--) {
          if (a.pin) p_until(t - 3);                 // one frame ahead of the owner
          float odv = 0.f;
          if (d_isr && lane < 16 && fj < Sg) odv = a.od[((size_t)(t - 2) * S + 4 * g + fj) * a.od_stride + dcol + fi];
          ++nd;
          if (!lds_wait_ge(dcnt, NSC * nd, abortf, limit)) { dead = true; break; }
          float part[NSC];
#pragma unroll
          for (int w = 0; w < NSC; w++) part[w] = red2f[(((nd & 1) * NSC + w) * 4 + fj) * 4 + fi];
          float sum = part[0];
#pragma unroll
          for (int w = 1; w < NSC; w++) sum += part[w];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __hip_atomic_store(dcons, nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (lane < 16 && fj < Sg) {
            if (d_isr) a.dr[((size_t)(t - 1) * S + 4 * g + fj) * R + dcol + fi] = odv + sum;          // :391
            else a.in_diff[((size_t)(t - 1) * S + 4 * g + fj) * a.id_stride + dcol + fi] = sum;       // :457
          }
        }
      }
      if (a.pin && !dead) p_until(0);
    }
  } else {
    // =========================== sweep-and-contract waves ===========================
    // Two lane layouts per 32-cell slot.  NATURAL (sweep, planes, elementwise): lane = (cell c32 = lane & 31, stream pair
    // h = lane >> 5: streams 2h, 2h+1) -- every plane load instruction reads two 128-byte row segments (lanes of a quad on
    // four different rows cost 1.8-3.0 us of plane loads per step here, and the slowest workgroup sets everybody's pace).
    // OPERAND (contraction): lane = (k-group b = lane >> 2, stream j = lane & 3).  dgifo goes from one to the other through a
    // wave-private 2 KB LDS tile [gate][cell][stream] (in-order DS queue of ONE wave: no barrier, no counter).
    const int w = wave - 2, b = lane >> 2, j = lane & 3, c32 = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x;
    float *xt = xtile + w * 512;                     // [4 gates][32 cells][4 streams]
    float wA[NU][2][4];                              // W_rm^T[own cell j][k = e*C + cell] for the wave's cells, A-operand order
    float wpi[NU], wpf[NU], wpo[NU];
    int svoff[NU];
    unsigned cmask = 0;                              // bit u: the lane's cell of slot u exists
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int sl = u * NSC + w;
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const int cell = 32 * sl + 16 * p + b;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          // W_rm[row (cell, gate e)][k = own cell 4 tile + j]: gates operand, row tile cell / 4, k chunk tile / 8, half tile & 1,
          // k-group (tile & 7) / 2, row slot 4 (cell & 3) + e, component j
          const int cc = cell < C ? cell : 0;
          const float v = a.wpk[(((((size_t)(cc >> 2) * a.nch1 + (tile >> 3)) * 2 + (tile & 1)) * 64 + ((tile & 7) >> 1) * 16 + 4 * (cc & 3) + e) << 2) + j];
          wA[u][p][e] = cell < C ? v : 0.f;
        }
      }
      const int cl = 32 * sl + c32;                  // the lane's cell in the natural layout
      if (cl < C) cmask |= 1u << u;
      const int clc = cl < C ? cl : 0;
      svoff[u] = clc * 32 + h * 16;
      wpi[u] = a.pi[clc]; wpf[u] = a.pf[clc]; wpo[u] = a.po[clc];
    }
    // The A operands of the d_r / in_diff columns: every SC wave fills the LDS slots it will read itself (wave-private, in-order
    // DS queue: no synchronisation), AFTER the workgroup barrier -- in front of it these 51 KB per workgroup delayed the P wave's
    // P(T) and with it the first publish of every workgroup (2 us per launch).
    if (d_on) {
      const float *src = (d_isr ? a.wrT : a.wxT) + (size_t)dcol * K + (size_t)j * K;
#pragma unroll 8
      for (int q = 0; q < NU * 8; q++) {             // q = (slot u, half p, gate e); 8 loads in flight
        const int e = q & 3, p = (q >> 2) & 1, sl = (q >> 3) * NSC + w, cell = 32 * sl + 16 * p + b;
        wD[((sl * 2 + p) * 4 + e) * 64 + lane] = cell < C ? src[e * C + cell] : 0.f;
      }
    }
    // plane addresses: lane part in a VGPR, slot part in the immediate, stream / gate / frame parts in the scalar offset
    const int voffG = (2 * h * K + 32 * w + c32) * 4, voffC = (2 * h * C + 32 * w + c32) * 4;
    const __amdgpu_buffer_rsrc_t rs_g = buf_rsrc(a.gifo, (T + 2) * S * K * 4), rs_h = buf_rsrc(a.hh, (T + 2) * S * C * 4);
    const __amdgpu_buffer_rsrc_t rs_c = buf_rsrc(a.cc, (T + 2) * S * C * 4);
    int nd = 0;
    bool dead = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
    for (int g = 0; g < ngrp && !dead; g++) {
      const int Sg = S - 4 * g < 4 ? S - 4 * g : 4;
      const bool need0 = 2 * h < Sg, need1 = 2 * h + 1 < Sg;
      const __amdgpu_buffer_rsrc_t rs_gr = buf_rsrc(a.gran + (size_t)g * BWD_RING * C * 4, BWD_RING * C * 32);
      const unsigned tag0 = epoch + (unsigned)(g * (T + 2));
      float carry[NU][2];
#pragma unroll
      for (int u = 0; u < NU; u++) { carry[u][0] = 0.f; carry[u][1] = 0.f; }
      const int tlo = (d_on && !d_isr) ? 1 : 2;      // frame 1 is swept only where in_diff(1) is contracted
      for (int t = T; t >= tlo; t--) {
        PT_MARK(5);
        // plane loads go out once the owner's publish of d_m(t) has been ISSUED: loads already in this CU's vector-memory
        // queue would hold the write-through store back, and with it every other workgroup
        if (!lds_wait_ge(pubn, g * T + (T - t + 1), abortf, limit)) { dead = true; break; }
        const int sG = (t * S + 4 * g) * K * 4, sC = (t * S + 4 * g) * C * 4;
        Bptt2Coef cf[NU][2];
#pragma unroll
        for (int u = 0; u < NU; u++)
#pragma unroll
          for (int x = 0; x < 2; x++) {                // stream 2h + x (absent streams read a later row or zeros: never used)
            const int imm = u * NSC * 32 * 4, oG = sG + x * K * 4, oC = sC + x * C * 4;
            const float yg = buf_f32(rs_g, voffG + imm, oG), yi = buf_f32(rs_g, voffG + imm, oG + C * 4);
            const float yf = buf_f32(rs_g, voffG + imm, oG + 2 * C * 4), yo = buf_f32(rs_g, voffG + imm, oG + 3 * C * 4);
            const float yh = buf_f32(rs_h, voffC + imm, oC), cpv = buf_f32(rs_c, voffC + imm, oC - S * C * 4);
            cf[u][x] = bptt2_coef(yg, yi, yf, yo, yh, cpv, wpi[u], wpf[u], wpo[u]);
          }
        PT_MARK(0);                                  // planes + coefficients
        // ---- sweep d_m(t): until every live tag equals tag0 + t ----
        for (int i = 0; i < a.nap0; i++) __builtin_amdgcn_s_sleep(4);
        const unsigned tag = tag0 + (unsigned)t;
        const int soff = (t % BWD_RING) * C * 32;
        u32x4 q[NU];
        {
          const long long t0 = wall_clock64();
          for (unsigned spins = 0;; spins++) {
#pragma unroll
            for (int u = 0; u < NU; u++) q[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_gr, svoff[u], soff, 16);   // aux 16 = sc1
            bool ok = true;
#pragma unroll
            for (int u = 0; u < NU; u++) {
              const unsigned t0g = q[u].y, t1g = q[u].w;
              const bool live = (cmask >> u) & 1u;
              ok &= !live | (((!need0) | (t0g == tag)) & ((!need1) | (t1g == tag)));
            }
            if (__all(ok)) break;
            if ((spins & 15) == 15) {
              if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { dead = true; break; }
              if (wall_clock64() - t0 > limit) {
                __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane == 0) { atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard)); atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t); }
                dead = true;
                break;
              }
            }
            for (int i = 0; i < a.nap; i++) __builtin_amdgcn_s_sleep(1);
          }
        }
        if (dead) break;
        PT_MARK(1);                                  // sweep
        // ---- elementwise BPTT of frame t (:411-440) for the lane's two streams, then into operand order through the tile ----
        float Bv[NU][2][4];
#pragma unroll
        for (int u = 0; u < NU; u++) {
          const unsigned v0 = q[u].x, v1 = q[u].z;
          const bool live = (cmask >> u) & 1u;       // (absent cells: zero weights, and nothing but zeros may meet them)
          float dcv;
          const float4 d0 = bptt2_apply(live ? __uint_as_float(v0) : 0.f, cf[u][0], carry[u][0], dcv);
          const float4 d1 = bptt2_apply(live ? __uint_as_float(v1) : 0.f, cf[u][1], carry[u][1], dcv);
          float2 *xw = reinterpret_cast<float2 *>(xt + c32 * 4 + 2 * h);
          xw[0] = make_float2(d0.x, d1.x); xw[64] = make_float2(d0.y, d1.y);           // gate e at + e*128 floats
          xw[128] = make_float2(d0.z, d1.z); xw[192] = make_float2(d0.w, d1.w);
          asm volatile("" ::: "memory");             // (lanes exchange data: the compiler must not move the reads above the writes;
                                                     //  the hardware runs one wave's DS operations in order)
#pragma unroll
          for (int p = 0; p < 2; p++)
#pragma unroll
            for (int e = 0; e < 4; e++) Bv[u][p][e] = xt[e * 128 + p * 64 + lane];       // [e][16p + b][j]
          asm volatile("" ::: "memory");
        }
        if (t > 1) {
          // ---- this wave's slice of dgifo(t) W_rm for the 4 own cells ----
          f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
          for (int u = 0; u < NU; u++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
              acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[u][p][0], Bv[u][p][0], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[u][p][1], Bv[u][p][1], acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[u][p][2], Bv[u][p][2], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[u][p][3], Bv[u][p][3], acc1, 0, 0, 0);
            }
          const f32x4 v = kgroup_sum_pl(acc0 + acc1);
          if (lane >= 12 && lane < 16) red[w * 4 + (lane & 3)] = v;   // stream lane & 3, components = the 4 own cells
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(pcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        PT_MARK(2);                                  // elementwise + contraction + partial
        if (d_on && (t > 1 || !d_isr)) {
          // ---- the same dgifo(t) against 4 rows of W_gifo_r^T / W_gifo_x^T (LDS): d_r(t-1) (:391) / in_diff(t) (:457) ----
          ++nd;
          // The P wave counts arrivals in ONE counter: nobody may add to it for step nd before it has passed its check for
          // step nd - 1 (a fast wave's early arrival would stand in for a slow wave's missing one).  It did so a whole step ago.
          if (nd > 1 && !lds_wait_ge(dcons, nd - 1, abortf, limit)) { dead = true; break; }
          float wd[NU][2][4];
#pragma unroll
          for (int u = 0; u < NU; u++)
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
              for (int e = 0; e < 4; e++) wd[u][p][e] = wD[(((u * NSC + w) * 2 + p) * 4 + e) * 64 + lane];
          f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
#pragma unroll
          for (int u = 0; u < NU; u++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
              d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[u][p][0], Bv[u][p][0], d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[u][p][1], Bv[u][p][1], d1, 0, 0, 0);
              d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[u][p][2], Bv[u][p][2], d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[u][p][3], Bv[u][p][3], d1, 0, 0, 0);
            }
          const f32x4 v = kgroup_sum_pl(d0 + d1);
          if (lane >= 12 && lane < 16) red2[((nd & 1) * NSC + w) * 4 + (lane & 3)] = v;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(dcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        PT_MARK(3);                                  // d_r / in_diff slice
      }
    }
    PT_FLUSH(0);
  }
  __syncthreads();
  if (tid == 0 && *abortf) {                         // (a bounded LDS wait expired or a sweep timed out)
    atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
    atomicMax(&a.ctrl[2], 0x80000000u | 0x7fffu);
    if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  finish(a.ctrl, epoch, ngrp * (T + 2), a.guard ? a.guard + 8 : nullptr, a.guard, a.hstat ? a.hstat + 1 : nullptr);
}

// -------------------------------------------------------------------------------------------------------------------
// 5..8 streams, INTERLEAVED: the two groups of 4 streams are two independent chains (a group's step t needs that group's step
// t + 1 from every workgroup, nothing of the other group).  k_bwd_persist2 runs them one after the other -- 2 T exchange round
// trips in a row.  Here every role walks (frame t, group 0), (frame t, group 1), (frame t - 1, group 0), ...: while the d_m of one
// group crosses the fabric the SC waves pull planes, apply and contract for the other, so a step-pair costs about two computes
// instead of two computes plus two round trips.  Same arithmetic, same instruction sequences per (cell, stream) as
// k_bwd_persist2: bit-identical results.  (Round 2 measured "groups pipelined against each other" slower -- in a kernel whose
// sweeper waves polled while OTHER waves of the workgroup wanted the vector-memory queue; here a wave never polls and loads at
// the same time, it does one after the other.)
// What is per group: granule slots and tags (as before), the carry, the owner's pair state, the partial buffers `red`, the P
// rows in LDS and the counters a role waits on for ONE group (partials written, publishes issued, frames of P): within a group
// the chain itself keeps the producers from running ahead, across groups nothing does, so a shared counter could be satisfied
// by a mix of two steps.  The d_r / in_diff partials keep their one counter with the hand-shake of k_bwd_persist2 (nobody adds
// for set n before the P wave has consumed set n - 1).
// -------------------------------------------------------------------------------------------------------------------
// NGI: interleaved stream groups (2: 5..8 streams; round 6: 3 / 4 for 9..16 streams -- the same walk (t, 0), (t, 1), ..., (t, NGI - 1), (t - 1, 0), ...)
template <int NW, int NU, int NGI = 2>
__global__ __launch_bounds__(NW * 64) void k_bwd_persist2i(PersistBwd2Args a) {
  constexpr int NSC = NW - 2, NSLOT = NSC * NU;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned *abortf = reinterpret_cast<unsigned *>(lds);
  int *dcnt = reinterpret_cast<int *>(lds) + 2;      // d_r / in_diff partials written
  int *dcons = reinterpret_cast<int *>(lds) + 3;     // d sets the P wave has consumed
  int *pcnt = reinterpret_cast<int *>(lds) + 4;      // [NGI] d_m partials written, per group (one count per SC wave and step)
  int *pdone = reinterpret_cast<int *>(lds) + 8;     // [NGI] frames of P in LDS, per group (descending frames)
  int *pubn = reinterpret_cast<int *>(lds) + 12;     // [NGI] publishes the owner has issued, per group
  f32x4 *red = reinterpret_cast<f32x4 *>(lds + 16);  // [NGI groups][NSC][4 streams]: components = the 4 own cells
  f32x4 *red2 = red + NGI * NSC * 4;                 // [2][NSC][4 streams]: components = the 4 d_r / in_diff columns
  float *xtile = reinterpret_cast<float *>(red2 + 2 * NSC * 4);   // [NSC][4 gates][32 cells][4 streams]: natural -> operand order
  float *wD = xtile + NSC * 512;                         // [NSLOT][2 sets][4 gates][64 lanes]: A operands of the d_r / in_diff columns
  f32x4 *ldsP = reinterpret_cast<f32x4 *>(wD + NSLOT * 512);     // [NGI groups][T][4 streams] (pin): components = the 4 own cells
  const int C = a.C, S = a.S, T = a.T, K = 4 * a.C;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long limit = a.spin_limit;
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned behind_giveup = 0u;                       // (as in k_bwd_persist2: looked at behind each role's prologue loads)
#ifndef KLSTM_NO_CHAIN_GUARD
  if (a.guard) behind_giveup = __hip_atomic_load(&a.guard[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                               __hip_atomic_load(&a.guard[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  const int ngr = a.R / 4, ngx = (a.din & 2) ? a.I / 4 : 0;
  const bool d_on = a.din && (int)blockIdx.x < ngr + ngx, d_isr = (int)blockIdx.x < ngr;
  const int dcol = d_isr ? (int)blockIdx.x * 4 : ((int)blockIdx.x - ngr) * 4;
  if (tid < 16) reinterpret_cast<int *>(lds)[tid] = 0;
  __syncthreads();
  auto sg_of = [&](int g) { return S - 4 * g < 4 ? S - 4 * g : 4; };   // streams of group g (the last one may be partial)

  if (a.tq && (int)blockIdx.x >= C / 4) {
    bwd_tail_role<NW, true, NGI>(a, lds, epoch, behind_giveup);           // a tail workgroup: d_r / in_diff partials of its 32 cells, off the chains
  } else if (wave == 0) {
    // =========================== owner: combine, publish, own plane rows ===========================
    const int oi = (lane >> 2) & 3, oj = lane & 3;
    const int ocell = (int)blockIdx.x * 4 + oi;
    const float wpi = a.pi[ocell], wpf = a.pf[ocell], wpo = a.po[ocell];
    const float *redf = reinterpret_cast<const float *>(red);
    bool dead = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
    bool on[NGI]; int srow[NGI]; unsigned long long *gr[NGI]; unsigned tag0[NGI];
    float carry[NGI], dmv[NGI];
    float yg[NGI], yi[NGI], yf[NGI], yo[NGI], yh[NGI], cpv[NGI];
    int np[NGI];
#pragma unroll
    for (int g = 0; g < NGI; g++) {
      const int Sg = sg_of(g);
      carry[g] = 0.f; dmv[g] = 0.f; np[g] = 0;
      on[g] = lane < 16 && oj < Sg;
      srow[g] = 4 * g + (oj < Sg ? oj : 0);
      gr[g] = a.gran + (size_t)g * BWD_RING * C * 4;
      tag0[g] = epoch + (unsigned)(g * (T + 2));
    }
    auto load_planes = [&](int t, int g) {
      const float *gp = a.gifo + ((size_t)t * S + srow[g]) * K + ocell;
      yg[g] = gp[0]; yi[g] = gp[C]; yf[g] = gp[2 * C]; yo[g] = gp[3 * C];
      yh[g] = a.hh[((size_t)t * S + srow[g]) * C + ocell];
      cpv[g] = a.cc[((size_t)(t - 1) * S + srow[g]) * C + ocell];
    };
    auto own_rows = [&](int t, int g) {
      const Bptt2Coef cf = bptt2_coef(yg[g], yi[g], yf[g], yo[g], yh[g], cpv[g], wpi, wpf, wpo);
      float dcv;
      const float4 d = bptt2_apply(dmv[g], cf, carry[g], dcv);
      if (on[g]) {
        float *dp = a.dgifo + ((size_t)t * S + srow[g]) * K + ocell;
        dp[0] = d.x; dp[C] = d.y; dp[2 * C] = d.z; dp[3 * C] = d.w;
        a.dc[((size_t)t * S + srow[g]) * C + ocell] = dcv;
      }
    };
#pragma unroll
    for (int g = 0; g < NGI; g++) {
      if (dead) break;
      if (on[g]) {                                   // dgifo(T+1) = 0 (:351): operand rows of the batched d_r product (tail outside)
        float *zp = a.dgifo + ((size_t)(T + 1) * S + srow[g]) * K + ocell;
        zp[0] = 0.f; zp[C] = 0.f; zp[2 * C] = 0.f; zp[3 * C] = 0.f;
      }
      if (a.pin) {
        if (!lds_wait_ge(pdone + g, 1, abortf, limit)) { dead = true; break; }
        dmv[g] = reinterpret_cast<const float *>(&ldsP[(g * T + T - 1) * 4 + oj])[oi];
      } else {
        dmv[g] = a.P[((size_t)(T - 1) * S + srow[g]) * C + ocell];
      }
      if (on[g] && !(a.test_stall == T && blockIdx.x == 0)) publish(gr[g] + (size_t)(T % BWD_RING) * C * 4, ocell * 4 + oj, tag0[g] + (unsigned)T, dmv[g]);
      __hip_atomic_store(pubn + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      load_planes(T, g);
    }
    for (int t = T; t >= 2 && !dead; t--) {
#pragma unroll
      for (int g = 0; g < NGI; g++) {
        if (dead) break;
        float pnext = 0.f;
        if (!a.pin) pnext = a.P[((size_t)(t - 2) * S + srow[g]) * C + ocell];
        ++np[g];
        if (!lds_wait_ge(pcnt + g, NSC * np[g], abortf, limit)) { dead = true; break; }
        float part[NSC];
#pragma unroll
        for (int w = 0; w < NSC; w++) part[w] = redf[((g * NSC + w) * 4 + oj) * 4 + oi];
        if (a.pin) {
          if (!lds_wait_ge(pdone + g, T - t + 2, abortf, limit)) { dead = true; break; }
          pnext = reinterpret_cast<const float *>(&ldsP[(g * T + t - 2) * 4 + oj])[oi];
        }
        float sum = part[0];
#pragma unroll
        for (int w = 1; w < NSC; w++) sum += part[w];  // fixed order
        const float dmn = sum + pnext;               // :408 with :391 substituted
        if (on[g] && !(a.test_stall == t - 1 && blockIdx.x == 0))
          publish(gr[g] + (size_t)((t - 1) % BWD_RING) * C * 4, ocell * 4 + oj, tag0[g] + (unsigned)(t - 1), dmn);
        __hip_atomic_store(pubn + g, T - t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        own_rows(t, g);
        dmv[g] = dmn;
        load_planes(t - 1, g);
      }
    }
    if (!dead) {
#pragma unroll
      for (int g = 0; g < NGI; g++) own_rows(1, g);
    }
  } else if (wave == 1) {
    // =========================== P wave: own columns of P = out_diff W_r_m; finishes d_r / in_diff ===========================
    const int kg = lane >> 2, bj = lane & 3, R = a.R;
    float4 w0[4], w1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int k = 128 * i + 4 * kg, pc = (int)blockIdx.x * 4 + bj;
      w0[i] = a.pin && k < R ? *reinterpret_cast<const float4 *>(a.wmT + (size_t)pc * R + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      w1[i] = a.pin && k + 64 < R ? *reinterpret_cast<const float4 *>(a.wmT + (size_t)pc * R + k + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float *red2f = reinterpret_cast<const float *>(red2);
    const int fi = (lane >> 2) & 3, fj = lane & 3;   // finishing lanes 0..15 = (column fi, stream fj)
    int nd = 0;
    bool dead = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
    int nextf = T - 1;                               // next frame of P (descending), BOTH groups per frame
    auto p_until = [&](int flo) {
      while (nextf >= flo && nextf >= 0) {
        const int f = nextf;
#pragma unroll
        for (int g = 0; g < NGI; g++) {
          const bool rv = bj < sg_of(g);
          const float *op = a.od + ((size_t)f * S + 4 * g + (rv ? bj : 0)) * a.od_stride + 4 * kg;
          float4 b0[4], b1[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int k = 128 * i + 4 * kg;
            b0[i] = rv && k < R ? *reinterpret_cast<const float4 *>(op + 128 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            b1[i] = rv && k + 64 < R ? *reinterpret_cast<const float4 *>(op + 128 * i + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float av[8] = {w0[i].x, w0[i].y, w0[i].z, w0[i].w, w1[i].x, w1[i].y, w1[i].z, w1[i].w};
            const float bv[8] = {b0[i].x, b0[i].y, b0[i].z, b0[i].w, b1[i].x, b1[i].y, b1[i].z, b1[i].w};
#pragma unroll
            for (int jj = 0; jj < 8; jj++) acc[jj & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[jj], bv[jj], acc[jj & 3], 0, 0, 0);
          }
          const f32x4 v = kgroup_sum_pl((acc[0] + acc[1]) + (acc[2] + acc[3]));
          if (lane >= 12 && lane < 16) ldsP[(g * T + f) * 4 + bj] = v;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __hip_atomic_store(pdone + g, T - f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        nextf--;
      }
    };
    if (!dead) {
      // d_r(T) = out_diff(T): dgifo(T+1) = 0 (:351, :391)
#pragma unroll
      for (int g = 0; g < NGI; g++)
        if (d_on && d_isr && lane < 16 && fj < sg_of(g))
          a.dr[((size_t)T * S + 4 * g + fj) * R + dcol + fi] = a.od[((size_t)(T - 1) * S + 4 * g + fj) * a.od_stride + dcol + fi];
      if (a.pin) p_until(T - 2);
      if (d_on) {
        for (int t = T; t >= (d_isr ? 2 : 1) && !dead; t--) {
          if (a.pin) p_until(t - 3);                 // one frame ahead of the owner
#pragma unroll
          for (int g = 0; g < NGI; g++) {
            if (dead) break;
            const int Sg = sg_of(g);
            float odv = 0.f;
            if (d_isr && lane < 16 && fj < Sg) odv = a.od[((size_t)(t - 2) * S + 4 * g + fj) * a.od_stride + dcol + fi];
            ++nd;
            if (!lds_wait_ge(dcnt, NSC * nd, abortf, limit)) { dead = true; break; }
            float part[NSC];
#pragma unroll
            for (int w = 0; w < NSC; w++) part[w] = red2f[(((nd & 1) * NSC + w) * 4 + fj) * 4 + fi];
            float sum = part[0];
#pragma unroll
            for (int w = 1; w < NSC; w++) sum += part[w];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __hip_atomic_store(dcons, nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane < 16 && fj < Sg) {
              if (d_isr) a.dr[((size_t)(t - 1) * S + 4 * g + fj) * R + dcol + fi] = odv + sum;          // :391
              else a.in_diff[((size_t)(t - 1) * S + 4 * g + fj) * a.id_stride + dcol + fi] = sum;       // :457
            }
          }
        }
      }
      if (a.pin && !dead) p_until(0);
    }
  } else {
    // =========================== sweep-and-contract waves (layouts: k_bwd_persist2) ===========================
    const int w = wave - 2, b = lane >> 2, j = lane & 3, c32 = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x;
    float *xt = xtile + w * 512;
    float wA[NU][2][4];
    float wpi[NU], wpf[NU], wpo[NU];
    int svoff[NU];
    unsigned cmask = 0;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int sl = u * NSC + w;
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const int cell = 32 * sl + 16 * p + b;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int cc = cell < C ? cell : 0;
          const float v = a.wpk[(((((size_t)(cc >> 2) * a.nch1 + (tile >> 3)) * 2 + (tile & 1)) * 64 + ((tile & 7) >> 1) * 16 + 4 * (cc & 3) + e) << 2) + j];
          wA[u][p][e] = cell < C ? v : 0.f;
        }
      }
      const int cl = 32 * sl + c32;
      if (cl < C) cmask |= 1u << u;
      const int clc = cl < C ? cl : 0;
      svoff[u] = clc * 32 + h * 16;
      wpi[u] = a.pi[clc]; wpf[u] = a.pf[clc]; wpo[u] = a.po[clc];
    }
    if (d_on) {
      const float *src = (d_isr ? a.wrT : a.wxT) + (size_t)dcol * K + (size_t)j * K;
#pragma unroll 8
      for (int q = 0; q < NU * 8; q++) {
        const int e = q & 3, p = (q >> 2) & 1, sl = (q >> 3) * NSC + w, cell = 32 * sl + 16 * p + b;
        wD[((sl * 2 + p) * 4 + e) * 64 + lane] = cell < C ? src[e * C + cell] : 0.f;
      }
    }
    const int voffG = (2 * h * K + 32 * w + c32) * 4, voffC = (2 * h * C + 32 * w + c32) * 4;
    const __amdgpu_buffer_rsrc_t rs_g = buf_rsrc(a.gifo, (T + 2) * S * K * 4), rs_h = buf_rsrc(a.hh, (T + 2) * S * C * 4);
    const __amdgpu_buffer_rsrc_t rs_c = buf_rsrc(a.cc, (T + 2) * S * C * 4);
    const __amdgpu_buffer_rsrc_t rs_gr = buf_rsrc(a.gran, NGI * BWD_RING * C * 32);   // [NGI groups][BWD_RING][C][4 streams] granules
    int nd = 0;
    bool dead = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
    float carry[NGI][NU][2];
#pragma unroll
    for (int g = 0; g < NGI; g++)
#pragma unroll
      for (int u = 0; u < NU; u++) { carry[g][u][0] = 0.f; carry[g][u][1] = 0.f; }
    const int tlo = (d_on && !d_isr) ? 1 : 2;        // frame 1 is swept only where in_diff(1) is contracted
    for (int t = T; t >= tlo && !dead; t--) {
#pragma unroll
      for (int g = 0; g < NGI; g++) {
        if (dead) break;
        const int Sg = sg_of(g);
        const bool need0 = 2 * h < Sg, need1 = 2 * h + 1 < Sg;
        // plane loads go out once the owner's publish of d_m(t) of THIS group has been issued (k_bwd_persist2)
        if (!lds_wait_ge(pubn + g, T - t + 1, abortf, limit)) { dead = true; break; }
        const int sG = (t * S + 4 * g) * K * 4, sC = (t * S + 4 * g) * C * 4;
        Bptt2Coef cf[NU][2];
#pragma unroll
        for (int u = 0; u < NU; u++)
#pragma unroll
          for (int x = 0; x < 2; x++) {
            const int imm = u * NSC * 32 * 4, oG = sG + x * K * 4, oC = sC + x * C * 4;
            const float yg = buf_f32(rs_g, voffG + imm, oG), yi = buf_f32(rs_g, voffG + imm, oG + C * 4);
            const float yf = buf_f32(rs_g, voffG + imm, oG + 2 * C * 4), yo = buf_f32(rs_g, voffG + imm, oG + 3 * C * 4);
            const float yh = buf_f32(rs_h, voffC + imm, oC), cpv = buf_f32(rs_c, voffC + imm, oC - S * C * 4);
            cf[u][x] = bptt2_coef(yg, yi, yf, yo, yh, cpv, wpi[u], wpf[u], wpo[u]);
          }
        // ---- sweep d_m(t) of group g ----
        for (int i = 0; i < a.nap0; i++) __builtin_amdgcn_s_sleep(4);
        const unsigned tag = epoch + (unsigned)(g * (T + 2)) + (unsigned)t;
        const int soff = (g * BWD_RING + t % BWD_RING) * C * 32;
        u32x4 q[NU];
        {
          const long long t0 = wall_clock64();
          for (unsigned spins = 0;; spins++) {
#pragma unroll
            for (int u = 0; u < NU; u++) q[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_gr, svoff[u], soff, 16);   // aux 16 = sc1
            bool ok = true;
#pragma unroll
            for (int u = 0; u < NU; u++) {
              const unsigned t0g = q[u].y, t1g = q[u].w;
              const bool live = (cmask >> u) & 1u;
              ok &= !live | (((!need0) | (t0g == tag)) & ((!need1) | (t1g == tag)));
            }
            if (__all(ok)) break;
            if ((spins & 15) == 15) {
              if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { dead = true; break; }
              if (wall_clock64() - t0 > limit) {
                __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane == 0) { atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard)); atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t); }
                dead = true;
                break;
              }
            }
            for (int i = 0; i < a.nap; i++) __builtin_amdgcn_s_sleep(1);
          }
        }
        if (dead) break;
        // ---- elementwise BPTT of frame t (:411-440), then into operand order through the wave's tile ----
        float Bv[NU][2][4];
#pragma unroll
        for (int u = 0; u < NU; u++) {
          const unsigned v0 = q[u].x, v1 = q[u].z;
          const bool live = (cmask >> u) & 1u;
          float dcv;
          const float4 d0 = bptt2_apply(live ? __uint_as_float(v0) : 0.f, cf[u][0], carry[g][u][0], dcv);
          const float4 d1 = bptt2_apply(live ? __uint_as_float(v1) : 0.f, cf[u][1], carry[g][u][1], dcv);
          float2 *xw = reinterpret_cast<float2 *>(xt + c32 * 4 + 2 * h);
          xw[0] = make_float2(d0.x, d1.x); xw[64] = make_float2(d0.y, d1.y);
          xw[128] = make_float2(d0.z, d1.z); xw[192] = make_float2(d0.w, d1.w);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int p = 0; p < 2; p++)
#pragma unroll
            for (int e = 0; e < 4; e++) Bv[u][p][e] = xt[e * 128 + p * 64 + lane];
          asm volatile("" ::: "memory");
        }
        if (t > 1) {
          f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
          for (int u = 0; u < NU; u++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
              acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[u][p][0], Bv[u][p][0], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[u][p][1], Bv[u][p][1], acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[u][p][2], Bv[u][p][2], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[u][p][3], Bv[u][p][3], acc1, 0, 0, 0);
            }
          const f32x4 v = kgroup_sum_pl(acc0 + acc1);
          if (lane >= 12 && lane < 16) red[(g * NSC + w) * 4 + (lane & 3)] = v;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(pcnt + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (d_on && (t > 1 || !d_isr)) {
          ++nd;
          if (nd > 1 && !lds_wait_ge(dcons, nd - 1, abortf, limit)) { dead = true; break; }
          float wd[NU][2][4];
#pragma unroll
          for (int u = 0; u < NU; u++)
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
              for (int e = 0; e < 4; e++) wd[u][p][e] = wD[(((u * NSC + w) * 2 + p) * 4 + e) * 64 + lane];
          f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
#pragma unroll
          for (int u = 0; u < NU; u++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
              d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[u][p][0], Bv[u][p][0], d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[u][p][1], Bv[u][p][1], d1, 0, 0, 0);
              d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[u][p][2], Bv[u][p][2], d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[u][p][3], Bv[u][p][3], d1, 0, 0, 0);
            }
          const f32x4 v = kgroup_sum_pl(d0 + d1);
          if (lane >= 12 && lane < 16) red2[((nd & 1) * NSC + w) * 4 + (lane & 3)] = v;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(dcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0 && *abortf) {                         // (a bounded LDS wait expired or a sweep timed out)
    atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
    atomicMax(&a.ctrl[2], 0x80000000u | 0x7fffu);
    if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  finish(a.ctrl, epoch, NGI * (T + 2), a.guard ? a.guard + 8 : nullptr, a.guard, a.hstat ? a.hstat + 1 : nullptr);
}

// -------------------------------------------------------------------------------------------------------------------
// launcher
// -------------------------------------------------------------------------------------------------------------------
static inline int pcdiv2(int a, int b) { return (a + b - 1) / b; }

// Geometry: waves per workgroup and 32-cell slots per SC wave.  16 waves (128 registers each): up to 3 slots (C <= 1344);
// 12 waves: 3 slots (C <= 960).
struct PGeo2 { int nw, nu; };
static PGeo2 pick_geo_bwd2(const Dims &d, const PersistOpts &o) {
  const int slots = pcdiv2(d.C, 32);
  const int order[2] = {o.bwd_waves == 12 ? 12 : 16, o.bwd_waves == 12 ? 16 : 12};
  for (int nw : order) {
    const int nu = pcdiv2(slots, nw - 2);
    if (nw == 16 && nu <= 3) return PGeo2{16, nu <= 2 ? 2 : 3};
    if (nw == 12 && nu <= 3) return PGeo2{12, 3};
  }
  return PGeo2{0, 0};
}
static size_t bwd2_lds_bytes(const PGeo2 &g, int T, bool pin, int ngi = 1) {
  const int nsc = g.nw - 2, nslot = nsc * g.nu, ng = ngi;                       // (interleaved: `red` and the P rows per group)
  return (size_t)(16 + ng * nsc * 16 + 2 * nsc * 16 + nsc * 512 + nslot * 512 + (pin ? ng * T * 16 : 0)) * sizeof(float);   // (a tail workgroup needs 4 KB of it)
}
// Tail workgroups (bwd_tail_role): one per 32-cell slot and column part.  Returns the column parts per slot (0: no tail workgroups --
// the chain's workgroups carry the columns, or the batched products run after the launch).  They need compute units of their own next
// to the chain's C / 4.
static int bwd2_tail_parts(const Dims &d, bool want_in_diff, const PersistOpts &o, int *ntw = nullptr, int *qpp = nullptr, int *tw0 = nullptr) {
  const PGeo2 g = pick_geo_bwd2(d, o);
  if (ntw) *ntw = 0;
  if (!g.nw || o.tail_mode == 2 || o.tail_mode == 0 || o.ncu <= d.C / 4 || d.R % 4 != 0 || d.I % 4 != 0) return 0;
  const int nq = d.R / 4 + (want_in_diff ? d.I / 4 : 0), nslots = pcdiv2(d.C, 32);
  // the fewest column parts whose quads fit the waves (TAIL_TG groups of 4 quads per contracting wave) -- every part is one more workgroup
  // per slot; wave 0 is left out of the contraction where the part fits without it
  for (int parts = 1; d.C / 4 + nslots * parts <= o.ncu; parts++) {
    const int per = 4 * pcdiv2(pcdiv2(nq, parts), 4);
    if (per > 4 * TAIL_TG * g.nw) continue;
    if (ntw) *ntw = nslots * parts;
    if (qpp) *qpp = per;
    if (tw0) *tw0 = per <= 4 * TAIL_TG * (g.nw - 1) ? 1 : 0;
    return parts;
  }
  return 0;
}
size_t persist_bwd_tail_ws_floats(const Dims &d, bool want_in_diff) {
  return (size_t)pcdiv2(d.C, 32) * d.T * d.S * (d.R + (want_in_diff ? d.I : 0));
}
// 5..8 streams: the two groups as interleaved chains (k_bwd_persist2i) unless the option says otherwise
static bool bwd_interleaved(const Dims &d, const PersistOpts &o) { return d.S > 4 && o.bwd_interleave != 0; }
static int bwd_groups(const Dims &d) { return (d.S + 3) / 4; }

bool persist_bwd_supported(const Dims &d, const PersistOpts &o) {
  if (d.S > 16 || d.C % 8 != 0 || d.R % 4 != 0 || d.C / 4 > 256) return false;
  const PGeo2 g = pick_geo_bwd2(d, o);
  if (d.S > 8 && (o.bwd_interleave == 0 || g.nw != 16 || g.nu != 2)) return false;   // 9..16 streams: three / four interleaved chains, 16 waves x 2 slots only
  return g.nw != 0;
}
int persist_bwd_grid(const Dims &d) { return d.C / 4; }
// P = out_diff W_r_m inside the backward launch: own columns in LDS (T frames x 4 streams x 4 cells), rows of W_r_m^T in registers
bool persist_p_in_kernel(const Dims &d, const PersistOpts &o) {
  const PGeo2 g = pick_geo_bwd2(d, o);
  return g.nw != 0 && d.R <= 512 && d.R % 4 == 0 && bwd2_lds_bytes(g, d.T, true, bwd_interleaved(d, o) ? bwd_groups(d) : 1) <= 152 * 1024;
}
// d_r and in_diff inside the backward launch: 4 columns per workgroup
bool persist_tail_in_chain(const Dims &d, bool want_in_diff, const PersistOpts &o) {       // ... on the chain's own workgroups (rounds 3-5)
  const PGeo2 g = pick_geo_bwd2(d, o);
  if (!g.nw || d.R % 4 != 0 || d.I % 4 != 0 || d.C % 4 != 0) return false;
  return d.R / 4 + (want_in_diff ? d.I / 4 : 0) <= d.C / 4;
}
bool persist_tail_in_kernel(const Dims &d, bool want_in_diff, const PersistOpts &o) {      // ... there, or on tail workgroups (given 16-byte rows at the boundary)
  return persist_tail_in_chain(d, want_in_diff, o) || bwd2_tail_parts(d, want_in_diff, o) > 0;
}

int persist_bwd_tail_wgs(const Dims &d, bool want_in_diff, const PersistOpts &o) {
  int ntw = 0;
  (void)bwd2_tail_parts(d, want_in_diff, o, &ntw);
  return ntw;
}

template <class Kn>
static hipError_t plaunch2(Kn kern, int grid, int threads, size_t shm, hipStream_t st, LaunchProbe pr, const PersistBwd2Args &a) {
  if (shm > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (pr.start) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, st, pr.start, pr.stop, 0, a);
  else hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, st, a);
  return hipGetLastError();
}

hipError_t launch_bwd_persist(const Dims &d, const BwdPtrs &p, const float *P, const float *out_diff, int od_stride,
                              float *in_diff, int id_stride, bool tail_inside, unsigned long long *gran, unsigned *ctrl,
                              const PersistOpts &o, hipStream_t st, LaunchProbe pr, float *tws, size_t tws_floats, LaunchProbe pr_reduce,
                              TailReduceJob *defer) {
  PersistBwd2Args a;
  if (defer) *defer = TailReduceJob{};
  a.C = d.C; a.R = d.R; a.S = d.S; a.T = d.T; a.I = d.I;
  a.pin = persist_p_in_kernel(d, o) && out_diff && (reinterpret_cast<uintptr_t>(out_diff) & 15) == 0 && od_stride % 4 == 0;
  if (!a.pin && !P) return hipErrorInvalidValue;
  a.od = out_diff; a.od_stride = od_stride; a.wmT = p.wmT; a.wrN = p.wr_nat; a.wxN = p.wx_nat;
  a.din = tail_inside ? (in_diff ? 3 : 1) : 0; a.wrT = p.wrT; a.wxT = p.wxT; a.dr = p.dr; a.in_diff = in_diff; a.id_stride = id_stride;
  if (a.din && !out_diff) return hipErrorInvalidValue;
  // d_r / in_diff on workgroups of their own next to the chain (compute units the chain leaves idle) instead of on the chain's
  int ntw = 0;
  a.tq = 0; a.tparts = 0; a.tqpp = 0; a.tw0 = 0; a.tws = tws;
  if (tail_inside && tws && out_diff && p.wr_nat && p.wx_nat && (reinterpret_cast<uintptr_t>(out_diff) & 15) == 0 && od_stride % 4 == 0 &&
      (!in_diff || ((reinterpret_cast<uintptr_t>(in_diff) & 15) == 0 && id_stride % 4 == 0)) &&
      tws_floats >= persist_bwd_tail_ws_floats(d, in_diff != nullptr))
    a.tparts = bwd2_tail_parts(d, in_diff != nullptr, o, &ntw, &a.tqpp, &a.tw0);
  if (a.tparts) { a.tq = d.R / 4 + (in_diff ? d.I / 4 : 0); a.din = 0; }
  else if (a.din && !persist_tail_in_chain(d, in_diff != nullptr, o)) return hipErrorInvalidValue;   // (neither form takes this call: the caller asks first)
  a.nch1 = p.nch_gates;
  a.wpk = reinterpret_cast<const float *>(p.pk_fold_gates); a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.dgifo = p.dgifo; a.dc = p.dc; a.P = P; a.gran = gran; a.ctrl = ctrl;
  a.nap0 = o.nap0_bwd >= 0 ? o.nap0_bwd : 0;
  a.nap = o.nap >= 0 ? o.nap : 0;
  a.spin_limit = o.spin_limit > 0 ? o.spin_limit : SPIN_LIMIT_DEFAULT;
  a.test_stall = o.test_stall_bwd;
  a.hstat = o.hstat;
  a.guard = o.guard;
#ifdef KLSTM_PERSIST_TIMING
  a.dbg = o.dbg;
#endif
  const PGeo2 g = pick_geo_bwd2(d, o);
  if (!g.nw || !p.pk_fold_gates || d.S > 16) return hipErrorInvalidValue;
  const int grid = persist_bwd_grid(d) + ntw;
  hipError_t err = hipErrorInvalidValue;
  if (bwd_interleaved(d, o)) {
    const int ngi = bwd_groups(d);
    const size_t shmi = bwd2_lds_bytes(g, d.T, a.pin != 0, ngi);
    if (ngi == 3 && g.nw == 16 && g.nu == 2) err = plaunch2(k_bwd_persist2i<16, 2, 3>, grid, 1024, shmi, st, pr, a);
    else if (ngi == 4 && g.nw == 16 && g.nu == 2) err = plaunch2(k_bwd_persist2i<16, 2, 4>, grid, 1024, shmi, st, pr, a);
    else if (ngi > 2) err = hipErrorInvalidValue;
    else if (g.nw == 16 && g.nu == 2) err = plaunch2(k_bwd_persist2i<16, 2>, grid, 1024, shmi, st, pr, a);
    else if (g.nw == 16 && g.nu == 3) err = plaunch2(k_bwd_persist2i<16, 3>, grid, 1024, shmi, st, pr, a);
    else if (g.nw == 12 && g.nu == 3) err = plaunch2(k_bwd_persist2i<12, 3>, grid, 768, shmi, st, pr, a);
  } else {
    const size_t shm = bwd2_lds_bytes(g, d.T, a.pin != 0);
    if (g.nw == 16 && g.nu == 2) err = plaunch2(k_bwd_persist2<16, 2>, grid, 1024, shm, st, pr, a);
    else if (g.nw == 16 && g.nu == 3) err = plaunch2(k_bwd_persist2<16, 3>, grid, 1024, shm, st, pr, a);
    else if (g.nw == 12 && g.nu == 3) err = plaunch2(k_bwd_persist2<12, 3>, grid, 768, shm, st, pr, a);
  }
  if (err != hipSuccess || !a.tq) return err;
  // the tail workgroups' partial rows -> d_r, in_diff
  TailReduceJob j;
  j.tws = tws; j.nslots = pcdiv2(d.C, 32); j.T = d.T; j.S = d.S; j.R = d.R; j.ncols = 4 * a.tq;
  j.od = out_diff; j.od_stride = od_stride; j.dr = p.dr; j.in_diff = in_diff; j.id_stride = id_stride;
  if (defer) { *defer = j; return hipSuccess; }      // (the gradient launch that follows runs it on its first workgroups)
  return launch_tail_reduce(j, o.guard, st, pr_reduce);
}

hipError_t launch_tail_reduce(const TailReduceJob &j, const unsigned *guard, hipStream_t st, LaunchProbe pr) {
  TailReduceArgs ra;
  ra.j = j; ra.guard = guard;
  const int nthr = j.T * j.S * (j.ncols / 4) * 8;
  if (pr.start) hipExtLaunchKernelGGL(k_tail_reduce, dim3(pcdiv2(nthr, 256)), dim3(256), 0, st, pr.start, pr.stop, 0, ra);
  else hipLaunchKernelGGL(k_tail_reduce, dim3(pcdiv2(nthr, 256)), dim3(256), 0, st, ra);
  return hipGetLastError();
}
int tail_reduce_blocks(const TailReduceJob &j) { return pcdiv2(pcdiv2(j.T * j.S * (j.ncols / 4) * 8, 256), 8) * 8; }

}  // namespace klstm
