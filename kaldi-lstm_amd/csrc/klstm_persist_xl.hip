// kaldi-lstm_amd/csrc/klstm_persist_xl.hip -- weights-RESIDENT forward and BPTT chains for MANY streams (9 .. 32 per GPU, bf16 operand mode,
// C = 1024; BASELINE.json configs[4]) as EIGHT INDEPENDENT machines, one per XCD (k_fwd_persist_xl; k_bwd_persist_xl further down): the stream groups of an LSTM are independent chains, and an XCD has what one chain of
// 4 streams needs -- 32 CUs whose registers hold a whole copy of W_rm between them (4096 x 1024 bf16 = 8 MB: 256 KB per CU, 64
// VGPRs per lane at 16 waves) and an L2 of its own, so that the per-step all-to-all of m(t) never leaves the XCD.
//
// klstm_persist_ms.hip spreads ONE copy of W_rm over all 256 CUs and exchanges m(t) of all 32 streams between all of them: 64 KB per
// step and workgroup through the fabric (every XCD has its own L2: sc1 write-through stores, sc1 loads from beyond the L2), 4.0 us
// per step for the exchange alone, 5.0 us per step in the launch.  Here a workgroup finds out which XCC it runs on
// (s_getreg HW_REG_XCC_ID), takes the next free slot of that XCC's group (an atomic counter per XCC) and serves the group's <= 4
// streams with 32 cells = 128 rows of W_rm; the exchange is PLAIN stores (the line stays in the XCD's L2) + sc1 loads among the 32
// workgroups of one XCC (tools/xcd_probe: 1.5 us per all-gather step inside an XCC against 2.1 across the chip, with plain stores
// across XCCs: stale reads).  Correct by construction: a group is DEFINED as the workgroups that read the same XCC id, so its
// members share an L2 wherever the dispatcher put them; what the dispatcher must deliver for the launch to RUN is 32 workgroups on
// every XCC (it does, one per CU, when the chip is free: profiles/r02_xcd_probe.txt) -- an XCC that gets 33 or 31 makes the launch give
// up (bounded waits, status word), and the engine answers a give-up by running the minibatch again on the other kernels
// (klstm_engine.hip recover()).
//   workgroup (XCC g, slot s): streams [g sx, g sx + sx), sx = ceil(S / 8); cells [32 s, 32 s + 32) = rows [128 s, 128 s + 128) of the
//     logical-row bf16 W_rm (launch_fold_ms); wave (row tile i = wave / 2, K half kp = wave % 2): 16 rows x 512 k resident
//   step t: every thread sweeps its share of the group's granules {tag, m(t-1) of two adjacent cells as bf16} (4 streams x 512 pairs, 8
//     bytes each: one 16-byte load per thread and pass) into the LDS slab [stream][cell]; barrier; v_mfma_f32_16x16x32_bf16 (weights on the M side, the group's streams on the N
//     side: a result lane holds g, i, f, o of one (cell, stream)); K halves combined through LDS; barrier; the cell update
//     (:278-309) on two waves, lane = (cell of 16, stream): publish, plane rows in 64-byte runs.  x(t) W_gifo_x^T + bias is the batched product of the
//     reference (:246, :259), in the gifo plane when the launch starts.
//   r(t) = W_r_m m(t) (:312): 16 rows of W_r_m per workgroup (R <= 512), every wave a sixteenth of K, one pass later (the slab of
//     step t + 1 IS m(t)); one more pass, T + 1, for r(T).
//   step 1 closes over the CARRIED r: natural W_gifo_r rows (K = R), transient registers.
// Rounding = that of the bf16 operand mode and of klstm_persist_ms.hip (tests/bf16_emul.py fold = True): same tests, same bars.
#include "klstm_kernels.h"
#include "klstm_math.h"
#include "klstm_persist_dev.h"

#include <hip/hip_ext.h>

namespace klstm {

#pragma clang fp contract(off)

typedef __bf16 xl_bf16x8 __attribute__((ext_vector_type(8)));

struct PersistXlArgs {
  int C, R, S, T, sx;             // C % 32 == 0, C <= XL_C (round 6; until then C = 1024 only); sx = streams per XCC group
  const unsigned short *wrm;      // folded W_rm as bf16, LOGICAL rows (4 cell + gate) x C
  const float *wr;                // natural W_gifo_r [4C x R] (step 1)
  const unsigned short *wrb;      // (or null) the same rounded to bf16, same layout (the fold product's operand plane): half the bytes of the prologue
  const float *wm;                // natural W_r_m [R x C]
  float *out; int out_stride;     // output rows [T*S x R] (:328)
  float *next_r;                  // r(T) (:331)
  const float *pi, *pf, *po;
  float *gifo, *cc, *hh, *mm, *rr; // planes; gifo rows of frames 1..T hold x W_gifo_x^T + bias on entry
  const float *prev_c, *prev_r;
  float *next_c;
  unsigned long long *gran;       // [8 groups][2 parities][4 streams][C / 2] granules {tag, m of two adjacent cells as bf16}
  unsigned *xcnt;                 // [8] workgroups registered per XCC (the last workgroup of a launch puts them back to 0)
  unsigned *ctrl;                 // [0] epoch, [1] finished workgroups, [2] status, [3] ordinal of the launch that gave up
  unsigned *guard;
  unsigned *hstat;
  long long spin_limit;
  int test_stall;
};

// XL_C: the LARGEST cell count (LDS slabs, granule arrays and thread maps are laid out for it); a layer with fewer cells (C % 32 == 0)
// uses the first C / 32 slots of every XCC group as cell owners -- the other workgroups of the group still project (16 rows of W_r_m
// each: R <= 512 needs all 32) and therefore sweep; K = C is cut into two halves of ceil(C / 64) chunks of 32, chunks past the operand
// carry zero weights against zero slab columns (the slab is cleared once, cells >= C are never written).
constexpr int XL_C = 1024, XL_LD = 2 * XL_C + 16;    // slab row bytes: [cell] bf16 + 16 (conflict-free ds_read_b128)

__device__ __forceinline__ xl_bf16x8 xl_load8(const float *p, bool on) {
  const float4 lo = on ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 hi = on ? *reinterpret_cast<const float4 *>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  return (xl_bf16x8){(__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w, (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w};
}

// FULL: C = XL_C as a compile-time constant (the configs[4] layers: every chunk live, every slot an owner -- the round-5 kernel to the
// instruction; the run-time cell count costs the BPTT launch 5 % there)
template <bool FULL>
__global__ __launch_bounds__(1024) void k_fwd_persist_xl(PersistXlArgs a) {
  const int C = FULL ? XL_C : a.C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *slab = smem;                                           // [5][XL_LD]: rows 0..3 = the group's streams, row 4 = zeros
  f32x4 *part = reinterpret_cast<f32x4 *>(smem + 5 * XL_LD);            // [2 K halves][8 row tiles][64]
  f32x4 *partr = part + 16 * 64;                                        // [16 waves][64]: partial projection tiles
  unsigned *abortf = reinterpret_cast<unsigned *>(partr + 16 * 64);
  int *pubcnt = reinterpret_cast<int *>(abortf + 1);
  unsigned *place = abortf + 2;                                         // [2]: XCC id, slot
  const int R = a.R, S = a.S, T = a.T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned behind_giveup = 0u;
  if (a.guard) behind_giveup = __hip_atomic_load(&a.guard[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                               __hip_atomic_load(&a.guard[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = tid * 16; i < 5 * XL_LD; i += 1024 * 16) *reinterpret_cast<uint4 *>(slab + i) = make_uint4(0u, 0u, 0u, 0u);
  if (tid == 0) {
    *abortf = 0u; *pubcnt = 0;
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;          // HW_REG_XCC_ID
    place[0] = xcc;
    place[1] = xcc < 8 ? atomicAdd(&a.xcnt[xcc], 1u) : 0xffffffffu;
  }
  __syncthreads();
  const int grp = (int)place[0], slot = (int)place[1];
  const int s0 = grp * a.sx, sxl = grp >= 8 ? 0 : (S - s0 < 0 ? 0 : (S - s0 < a.sx ? S - s0 : a.sx));   // this group's streams
  const bool misplaced = grp >= 8 || slot >= 32 || slot < 0;           // (a 33rd workgroup on an XCC: the launch cannot run)
  const bool skip = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
  if (misplaced && !skip && tid == 0) {
    atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
    atomicMax(&a.ctrl[2], 0x80000000u | 0x7ffeu);
    if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const bool owner = !misplaced && 32 * slot < C;                       // this workgroup owns 32 cells (C < XL_C: the first C / 32 slots of a group)
  const bool idle = misplaced || skip || sxl == 0 || (!owner && 16 * slot >= R);   // (a group without streams has nothing to exchange; a slot without cells or rows of W_r_m nothing to do)

  if (!idle) {
    const int i16 = lane & 15, kg = lane >> 4;                          // MFMA operand lane: row / column i16, k-group kg
    const int ti = wave >> 1, kp = wave & 1;                            // row tile, K half
    const int nchC = C / 32, kh = (nchC + 1) / 2;                       // chunks of 32 over K = C, per K half
    unsigned long long *gr = a.gran + (size_t)grp * 2 * 4 * (XL_C / 2); // the group's granules: [2 parities][4 streams][XL_C / 2 cell pairs]
    const __amdgpu_buffer_rsrc_t rs_gr = buf_rsrc(gr, 2 * 4 * (XL_C / 2) * 8);
    // ---- cell-update lanes: waves 0 and 1, lane = (cell cl = lane & 15 of the wave's 16, stream n2 = lane >> 4): one (cell, stream) pair
    //      per lane, taken out of the partial tiles in LDS (with the MFMA result layout as the cell layout -- 8 waves, 16 live lanes each --
    //      every plane store wrote sixteen 16-byte runs and the workgroup issued 64 of them per step in front of the next sweep) ----
    const bool cellw = wave < 2 && owner;
    const int cl = lane & 15, n2 = lane >> 4, c32 = 16 * (wave & 1) + cl;
    const bool on = cellw && n2 < sxl;
    const int cell = owner ? 32 * slot + c32 : 0, strm = s0 + (n2 < sxl ? n2 : 0);
    const int psrc = (c32 >> 2) * 64 + 16 * (c32 & 3) + n2;            // tile c32 / 4, result lane (k-group c32 % 4, column n2): g, i, f, o of the pair
    float cp = on ? a.prev_c[(size_t)strm * C + cell] : 0.f;            // carried c(0) (:231)
    if (on) a.cc[(size_t)strm * C + cell] = cp;                         // time block 0 of the c plane: BPTT reads it
    const float wpi = a.pi[cell], wpf = a.pf[cell], wpo = a.po[cell];   // (no cells here: cell = 0, never used)
    // ---- time block 0 of the r plane + the slab of step 1: the carried r(0) of the group's streams, rounded like every staged activation
    for (int i = tid; i < sxl * (R / 4); i += 1024) {
      const int n = i / (R / 4), k = (i % (R / 4)) * 4;
      const float4 rv = *reinterpret_cast<const float4 *>(a.prev_r + (size_t)(s0 + n) * R + k);
      *reinterpret_cast<uint2 *>(slab + n * XL_LD + 2 * k) =
          make_uint2(bf16_rne(rv.x) | ((unsigned)bf16_rne(rv.y) << 16), bf16_rne(rv.z) | ((unsigned)bf16_rne(rv.w) << 16));
      if (slot == 0) *reinterpret_cast<float4 *>(a.rr + (size_t)(s0 + n) * R + k) = rv;
    }
    // ---- operands: step-1 rows (natural W_gifo_r, this wave's half of K = R), then the resident rows of W_rm ----
    // operand row of this lane: tile row i16 = 4 cell' + gate  ->  natural row gate C + (32 slot + 4 ti + cell')
    const size_t arow = (size_t)(i16 & 3) * C + (owner ? 32 * slot + 4 * ti + (i16 >> 2) : 0);
    const int nchU = R / 32, cwU = (nchU + 1) / 2;                      // chunks of 32 over R, per K half
    xl_bf16x8 uf[8], af[16];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int cu = kp * cwU + j;
      const bool in = owner && j < cwU && cu < nchU;
      if (a.wrb) uf[j] = in ? *reinterpret_cast<const xl_bf16x8 *>(a.wrb + arow * R + 32 * cu + 8 * kg) : (xl_bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      else uf[j] = xl_load8(a.wr + arow * R + 32 * cu + 8 * kg, in);
    }
    const unsigned char *brow = slab + (i16 < 4 ? i16 : 4) * XL_LD + 16 * kg;   // B operand row of this lane (columns >= 4: the zero row)
    // projection: rows 16 slot + i16 of W_r_m (R <= 512), this wave's sixteenth of K = C (2 chunks of 32)
    const int prow = 16 * slot + i16;
    const bool projw = 16 * slot < R;
    xl_bf16x8 rf[2];
    auto run_step = [&](int t, bool first) -> bool {
      float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);                      // x(t) W_gifo_x^T + bias of this lane's (cell, stream)
      if (on && t <= T) {
        const float *gp = a.gifo + ((size_t)t * S + strm) * 4 * C + cell;
        xg = make_float4(gp[0], gp[C], gp[2 * C], gp[3 * C]);
      }
      if (!first) {
        // ---- sweep m(t-1) of the group: thread = (stream n = tid >> 8, cells 4 (tid & 255) .. + 3): ONE 16-byte sc1 load = two granules ----
        // (polling starts once this workgroup's own cell waves have issued their publishes of step t-1: klstm_persist.hip)
        if (owner) {
          const long long w0 = wall_clock64();
          for (unsigned spins = 0; __hip_atomic_load(pubcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2 * (t - 1); spins++) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 1023) == 1023 && wall_clock64() - w0 > a.spin_limit) break;
          }
        }
        const int n = tid >> 8, c4 = tid & 255;
        const bool live = n < sxl && 4 * c4 < C;
        const unsigned tag = epoch + (unsigned)(t - 1);
        const int off = ((((t - 1) & 1) * 4 + (live ? n : 0)) * (XL_C / 2) + 2 * (live ? c4 : 0)) * 8;
        u32x4 q0;
        bool ok = false;
        const long long t0 = wall_clock64();
        for (unsigned spins = 0;; spins++) {
          q0 = __builtin_amdgcn_raw_buffer_load_b128(rs_gr, off, 0, 16);         // aux 16 = sc1
          ok = !live | ((q0.y == tag) & (q0.w == tag));
          if (__all(ok)) break;
          if ((spins & 31) == 31) {
            if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            if (wall_clock64() - t0 > a.spin_limit) break;
          }
        }
        if (!__all(ok)) {
          __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (lane == 0) {
            atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
            atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t);
            if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        // (the slab is free: every wave of this workgroup passed barrier (2) of step t-1 behind its reads)
        if (live) *reinterpret_cast<uint2 *>(slab + n * XL_LD + 8 * c4) = make_uint2(q0.x, q0.z);      // (bf16 already: rounded by the publisher)
      }
      lds_barrier();                                                    // (1) slab of step t ready
      if (*abortf) return false;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accr = {0.f, 0.f, 0.f, 0.f};
      if (first) {
#pragma unroll
        for (int j = 0; j < 8; j++) {                                   // (chunks past the operand: zero weights against the last chunk's columns)
          const int ch = kp * cwU + j < nchU ? kp * cwU + j : nchU - 1;
          const xl_bf16x8 bv = *reinterpret_cast<const xl_bf16x8 *>(brow + 64 * ch);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf[j], bv, acc, 0, 0, 0);
        }
      } else {
        if (t <= T && owner) {
#pragma unroll
          for (int hf = 0; hf < 8; hf++) {                              // (two chunks' operands in flight at a time: the resident rows leave few registers)
            xl_bf16x8 bv[2];
#pragma unroll
            for (int j = 0; j < 2; j++) bv[j] = *reinterpret_cast<const xl_bf16x8 *>(brow + 64 * (kh * kp + 2 * hf + j));
#pragma unroll
            for (int j = 0; j < 2; j++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2 * hf + j], bv[j], acc, 0, 0, 0);
          }
        }
        if (projw) {
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const xl_bf16x8 bv = *reinterpret_cast<const xl_bf16x8 *>(brow + 64 * (2 * wave + j));
            accr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf[j], bv, accr, 0, 0, 0);
          }
          partr[wave * 64 + lane] = accr;
        }
      }
      part[(kp * 8 + ti) * 64 + lane] = acc;
      lds_barrier();                                                    // (2) the partial tiles are in LDS
      if (!first && projw && wave == 2) {
        // ---- r(t-1): lane (stream n = i16, rows 16 slot + 4 kg .. + 3): the sixteen K parts in fixed order ----
        f32x4 v = partr[lane];
#pragma unroll
        for (int w = 1; w < 16; w++) v = v + partr[w * 64 + lane];
        const int f = t - 1, col = 16 * slot + 4 * kg;
        if (i16 < sxl && col < R) {
          const int sr = s0 + i16;
          *reinterpret_cast<float4 *>(a.rr + ((size_t)f * S + sr) * R + col) = make_float4(v.x, v.y, v.z, v.w);
          float *op = a.out + ((size_t)(f - 1) * S + sr) * a.out_stride + col;
          op[0] = v.x; op[1] = v.y; op[2] = v.z; op[3] = v.w;
          if (f == T) *reinterpret_cast<float4 *>(a.next_r + (size_t)sr * R + col) = make_float4(v.x, v.y, v.z, v.w);
        }
      }
      if (cellw && t <= T) {
        // ---- cell update of (cell, stream): the two K halves in fixed order, then :278-309 ----
        const f32x4 v = part[psrc] + part[8 * 64 + psrc];
        float ai = v.y + xg.y, af_ = v.z + xg.z, ao = v.w + xg.w;
        const float ag = v.x + xg.x;
        ai += wpi * cp;                                                 // :278
        af_ += wpf * cp;                                                // :281
        const float gi = k_sigmoid(ai), gf = k_sigmoid(af_), gg = k_tanh(ag);   // :284-288
        float c = gg * gi;                                              // :291
        c = c + cp * gf;                                                // :294
        c = c < -50.f ? -50.f : c;                                      // :296
        c = c > 50.f ? 50.f : c;                                        // :297
        const float h = k_tanh(c);                                      // :300
        ao += wpo * c;                                                  // :303
        const float go = k_sigmoid(ao);                                 // :306
        const float m = h * go;                                         // :309
        // publish m(t) (m(T) travels too: r(T)) as bf16 (what every consumer rounds it to), two adjacent cells per granule: the lane of
        // the even cell takes its neighbour's value (the next lane) and issues ONE plain 8-byte store -- the line stays in this XCC's L2
        const unsigned mb = bf16_rne(m), mb_up = (unsigned)__shfl_down((int)mb, 1);
        if (on && (cl & 1) == 0 && !(a.test_stall == t && slot == 0 && grp == 0))
          gr[((size_t)(t & 1) * 4 + n2) * (XL_C / 2) + (cell >> 1)] = ((unsigned long long)(epoch + (unsigned)t) << 32) | (mb | (mb_up << 16));
        if (on) {
          float *gp = a.gifo + ((size_t)t * S + strm) * 4 * C + cell;
          gp[0] = gg; gp[C] = gi; gp[2 * C] = gf; gp[3 * C] = go;
          const size_t pc = ((size_t)t * S + strm) * C + cell;
          a.cc[pc] = c; a.hh[pc] = h; a.mm[pc] = m;
          if (t == T) a.next_c[(size_t)strm * C + cell] = c;            // :331 (c columns)
        }
        cp = c;
        if (lane == 0) __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // publishes of this step issued
      }
      return true;
    };
    if (run_step(1, true)) {
#pragma unroll
      for (int j = 0; j < 16; j++) {                                    // (C = 1024: kh = 16, every chunk live)
        const int ch = kh * kp + j;
        const bool in = owner && j < kh && ch < nchC;
        af[j] = *reinterpret_cast<const xl_bf16x8 *>(a.wrm + (size_t)(in ? 128 * slot + 16 * ti + i16 : 0) * C + 32 * (in ? ch : 0) + 8 * kg);
        if (!in) af[j] = (xl_bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int k0 = 64 * wave + 32 * j;
        rf[j] = xl_load8(a.wm + (size_t)(prow < R ? prow : 0) * C + (k0 < C ? k0 : 0) + 8 * kg, projw && prow < R && k0 < C);
      }
      for (int t = 2; t <= T + 1; t++)                                   // (t = T + 1: r(T) only)
        if (!run_step(t, false)) break;
    }
  }
  // ---- end of launch: the last workgroup moves the epoch on and puts the per-XCC counters back ----
  __syncthreads();
  if (threadIdx.x == 0) {
    if (*abortf) {
      atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
      atomicMax(&a.ctrl[2], 0x80000000u | 0x7fffu);
      if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned old = atomicAdd(&a.ctrl[1], 1u);
    if (old == gridDim.x - 1) {
      for (int i = 0; i < 8; i++) __hip_atomic_store(&a.xcnt[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.ctrl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.ctrl[0], epoch + (unsigned)(T + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.guard) __hip_atomic_fetch_add(a.guard + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------
// BACKWARD chain, the same eight machines: d_m(t) = P(t) + dgifo(t+1) W_rm (...streams.h:391 substituted into :408; P = out_diff W_r_m,
// the batched product in front of the launch), then the elementwise BPTT of the workgroup's OWN cells (:411-440) -- nothing is
// replicated: what travels is dgifo(t) of the group's streams, 4 gates x 1024 cells x <= 4 streams as bf16 (the operand every
// consumer rounds it to), inside the XCD.  A workgroup owns 32 output cells = 32 rows of W_rm^T over K = 4C (the fold product writes
// the transpose next to W_rm: klstm_fold3.hip wlT): wave w holds both row tiles (2 x 16 rows) x the K sixteenth [256 w, 256 w + 256): the operand slab is read once per k.
// d_r(1..T) = out_diff + dgifo(2..T+1) W_gifo_r (:391: the W_r_m gradient's operand) and in_diff = dgifo W_gifo_x (:457) are batched
// products behind the launch.  Granule = 16 bytes {tag, (d_g, d_i), (d_f, d_o)} per (cell, stream): 8 contiguous bytes of the slab row
// [k = 4 cell + gate].
// -------------------------------------------------------------------------------------------------------------------
struct PersistXlBwdArgs {
  int C, S, T, sx;                // C % 32 == 0, C <= XL_C (round 6): the first C / 32 slots of every XCC group own cells, the others idle
  const unsigned short *wrmT;     // [C][4C] bf16: row c = column c of W_rm over the logical rows k = 4 cell + gate
  const float *P;                 // out_diff W_r_m [T*S x C]
  const float *pi, *pf, *po;
  const float *gifo, *cc, *hh;    // forward planes
  float *dgifo, *dc;
  unsigned short *dgifo_h;        // (or null) the same rows rounded to bf16 (the values that travel in the granules): operand copy of the batched d_r / in_diff products
  uint4 *gran;                    // [8 groups][2 parities][4 streams][C] granules
  unsigned *xcnt;
  unsigned *ctrl;                 // the backward direction's control words
  unsigned *guard;
  unsigned *hstat;
  long long spin_limit;
  int test_stall;
};

constexpr int XL_LDB = 2 * 4 * XL_C + 16;            // backward slab row bytes: [4 cell + gate] bf16 + 16

template <bool FULL>
__global__ __launch_bounds__(1024) void k_bwd_persist_xl(PersistXlBwdArgs a) {
  const int C = FULL ? XL_C : a.C, K = 4 * C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *slab = smem;                                           // [5][XL_LDB]: rows 0..3 = dgifo(t+1) of the group's streams, row 4 = zeros
  f32x4 *part = reinterpret_cast<f32x4 *>(smem + 5 * XL_LDB);           // [16 K parts][2 row tiles][64]
  unsigned *abortf = reinterpret_cast<unsigned *>(part + 32 * 64);
  int *pubcnt = reinterpret_cast<int *>(abortf + 1);
  unsigned *place = abortf + 2;
  const int S = a.S, T = a.T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned behind_giveup = 0u;
  if (a.guard) behind_giveup = __hip_atomic_load(&a.guard[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                               __hip_atomic_load(&a.guard[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = tid * 16; i < 5 * XL_LDB; i += 1024 * 16) *reinterpret_cast<uint4 *>(slab + i) = make_uint4(0u, 0u, 0u, 0u);
  if (tid == 0) {
    *abortf = 0u; *pubcnt = 0;
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;          // HW_REG_XCC_ID
    place[0] = xcc;
    place[1] = xcc < 8 ? atomicAdd(&a.xcnt[xcc], 1u) : 0xffffffffu;
  }
  __syncthreads();
  const int grp = (int)place[0], slot = (int)place[1];
  const int s0 = grp * a.sx, sxl = grp >= 8 ? 0 : (S - s0 < 0 ? 0 : (S - s0 < a.sx ? S - s0 : a.sx));
  const bool misplaced = grp >= 8 || slot >= 32 || slot < 0;
  const bool skip = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;
  if (misplaced && !skip && tid == 0) {
    atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
    atomicMax(&a.ctrl[2], 0x80000000u | 0x7ffeu);
    if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const bool idle = misplaced || skip || sxl == 0 || 32 * slot >= C;   // (C < XL_C: slots without cells have nothing to do here)

  if (!idle) {
    const int i16 = lane & 15, kg = lane >> 4;
    const int kp = wave;                                                // K sixteenth of BOTH row tiles: the slab is read once per k
    const int nchK = K / 32, cw = (nchK + 15) / 16;                     // chunks of 32 over K = 4C, per wave (C = 1024: 8 = 256 k)
    const int ti = wave & 1;                                            // (elementwise waves 0 and 1: the row tile whose cells they update)
    uint4 *gr = a.gran + (size_t)grp * 2 * 4 * XL_C;                     // the group's granules: [2 parities][4 streams][XL_C cells]
    const __amdgpu_buffer_rsrc_t rs_gr = buf_rsrc(gr, 2 * 4 * XL_C * 16);
    // ---- elementwise lanes: the two waves with kp = 0, lane = (cell cl = lane & 15 of the wave's row tile, stream n2 = lane >> 4):
    //      one (cell, stream) pair per lane (four pairs per lane in the MFMA result layout cost 56 registers of state and operands)
    const bool cellw = wave < 2;
    const int cl = lane & 15, n2 = lane >> 4;
    const bool on = cellw && n2 < sxl;
    const int cell = 32 * slot + 16 * ti + cl, strm = s0 + (n2 < sxl ? n2 : 0);
    const float wpi = a.pi[cell], wpf = a.pf[cell], wpo = a.po[cell];
    float dcn = 0.f, fn = 0.f, din = 0.f, dfn = 0.f;
    if (on) {                                                           // dgifo(T+1) = 0 (:351): operand rows of the batched d_r product
      float *zp = a.dgifo + ((size_t)(T + 1) * S + strm) * K + cell;
      zp[0] = 0.f; zp[C] = 0.f; zp[2 * C] = 0.f; zp[3 * C] = 0.f;
      if (a.dgifo_h) {
        unsigned short *zh = a.dgifo_h + ((size_t)(T + 1) * S + strm) * K + cell;
        zh[0] = 0; zh[C] = 0; zh[2 * C] = 0; zh[3 * C] = 0;
      }
    }
    // ---- the resident operand: 2 x 16 rows of W_rm^T (output cells 32 slot + 16 tile + i16) x this wave's sixteenth of K ----
    xl_bf16x8 af[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int ch = cw * kp + (j & 7);
      const bool in = (j & 7) < cw && ch < nchK;
      af[j] = *reinterpret_cast<const xl_bf16x8 *>(a.wrmT + (size_t)(32 * slot + 16 * (j >> 3) + i16) * K + 32 * (in ? ch : 0) + 8 * kg);
      if (!in) af[j] = (xl_bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    const unsigned char *brow = slab + (i16 < 4 ? i16 : 4) * XL_LDB + 16 * kg;
    bool dead = false;
    for (int t = T; t >= 1 && !dead; t--) {
      if (t < T) {
        // ---- sweep dgifo(t+1) of the group: thread = (stream n = tid >> 8, cells (tid & 255) + 256 e, e = 0..3): four 16-byte sc1 loads, each
        //      one contiguous KB per wave (lane-strided 64 bytes instead: 32 lines per instruction x 4, 83 -> ... us at T = 20) ----
        {
          const long long w0 = wall_clock64();
          for (unsigned spins = 0; __hip_atomic_load(pubcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2 * (T - t); spins++) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 1023) == 1023 && wall_clock64() - w0 > a.spin_limit) break;
          }
        }
        const int n = tid >> 8, c4 = tid & 255;
        const bool live = n < sxl;
        const unsigned tag = epoch + (unsigned)(t + 1);
        const int off = ((((t + 1) & 1) * 4 + (live ? n : 0)) * XL_C + c4) * 16;
        const bool le[4] = {c4 < C, c4 + 256 < C, c4 + 512 < C, c4 + 768 < C};     // (cells past C: no granule, nothing to wait for)
        u32x4 q[4];
        bool ok = false;
        const long long t0 = wall_clock64();
        for (unsigned spins = 0;; spins++) {
#pragma unroll
          for (int e = 0; e < 4; e++) q[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_gr, off + 4096 * e, 0, 16);   // aux 16 = sc1
          ok = !live | ((!le[0] | (q[0].x == tag)) & (!le[1] | (q[1].x == tag)) & (!le[2] | (q[2].x == tag)) & (!le[3] | (q[3].x == tag)));
          if (__all(ok)) break;
          if ((spins & 31) == 31) {
            if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            if (wall_clock64() - t0 > a.spin_limit) break;
          }
        }
        if (!__all(ok)) {
          __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (lane == 0) {
            atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
            atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t);
            if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        if (live) {
          uint2 *sp = reinterpret_cast<uint2 *>(slab + n * XL_LDB + 8 * c4);     // slab row [k = 4 cell + gate] bf16: 8 bytes per cell
#pragma unroll
          for (int e = 0; e < 4; e++) if (le[e]) sp[256 * e] = make_uint2(q[e].y, q[e].z);
        }
      }
      // the own pair's operands of frame t: requested here, consumed behind the two barriers
      float Pv = 0.f, yg = 0.f, yi = 0.f, yf = 0.f, yo = 0.f, yh = 0.f, cpv = 0.f;
      if (on) {
        Pv = a.P[((size_t)(t - 1) * S + strm) * C + cell];
        const float *gp = a.gifo + ((size_t)t * S + strm) * K + cell;
        yg = gp[0]; yi = gp[C]; yf = gp[2 * C]; yo = gp[3 * C];
        yh = a.hh[((size_t)t * S + strm) * C + cell];
        cpv = a.cc[((size_t)(t - 1) * S + strm) * C + cell];
      }
      lds_barrier();                                                    // (1) slab of dgifo(t+1) ready
      if (*abortf) { dead = true; break; }
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      if (t < T) {
#pragma unroll
        for (int hf = 0; hf < 4; hf++) {
          xl_bf16x8 bv[2];
#pragma unroll
          for (int j = 0; j < 2; j++) bv[j] = *reinterpret_cast<const xl_bf16x8 *>(brow + 64 * (cw * kp + 2 * hf + j));
#pragma unroll
          for (int j = 0; j < 2; j++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2 * hf + j], bv[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[8 + 2 * hf + j], bv[j], acc1, 0, 0, 0);
          }
        }
      }
      part[(kp * 2 + 0) * 64 + lane] = acc0;
      part[(kp * 2 + 1) * 64 + lane] = acc1;
      lds_barrier();                                                    // (2) the partial tiles are in LDS
      if (cellw) {
        // d_m of this lane's (cell, stream): element (row cl, column n2) of the tile = component cl & 3 of result lane 16 (cl >> 2) + n2;
        // the sixteen K parts in fixed order, then :408 with :391 substituted
        const float *pf32 = reinterpret_cast<const float *>(part);
        const int pe = ((cl >> 2) * 16 + n2) * 4 + (cl & 3);
        float dm = pf32[(ti * 64) * 4 + pe];
#pragma unroll
        for (int w = 1; w < 16; w++) dm += pf32[((w * 2 + ti) * 64) * 4 + pe];
        dm += Pv;
        const float d_h = k_diff_tanh(dm * yo, yh);                     // :411-412
        const float d_o = k_diff_sigmoid(dm * yh, yo);                  // :415-416
        float d_c = d_h;                                                // :424
        d_c = d_c + dcn * fn;                                           // :425
        d_c = d_c + wpi * din;                                          // :426
        d_c = d_c + wpf * dfn;                                          // :427
        d_c = d_c + wpo * d_o;                                          // :428
        const float dg = k_diff_tanh(d_c * yi, yg);                     // :439-440
        const float di = k_diff_sigmoid(d_c * yg, yi);                  // :435-436
        const float df = k_diff_sigmoid(d_c * cpv, yf);                 // :431-432
        dcn = d_c; fn = yf; din = di; dfn = df;                         // what frame t - 1 needs of frame t
        if (on) {
          const unsigned short hg = bf16_rne(dg), hi = bf16_rne(di), hf = bf16_rne(df), ho = bf16_rne(d_o);
          if (t > 1 && !(a.test_stall == t && slot == 0 && grp == 0))   // publish dgifo(t): one plain 16-byte store -- the line stays in this XCC's L2
            gr[((size_t)(t & 1) * 4 + n2) * XL_C + cell] = make_uint4(epoch + (unsigned)t, hg | ((unsigned)hi << 16), hf | ((unsigned)ho << 16), 0u);
          float *dp = a.dgifo + ((size_t)t * S + strm) * K + cell;
          dp[0] = dg; dp[C] = di; dp[2 * C] = df; dp[3 * C] = d_o;
          if (a.dgifo_h) {                                              // (2-byte stores in 32-byte runs: sixteen cells of a tile are sixteen lanes)
            unsigned short *dh = a.dgifo_h + ((size_t)t * S + strm) * K + cell;
            dh[0] = hg; dh[C] = hi; dh[2 * C] = hf; dh[3 * C] = ho;
          }
          a.dc[((size_t)t * S + strm) * C + cell] = d_c;
        }
        if (lane == 0) __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (*abortf) {
      atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
      atomicMax(&a.ctrl[2], 0x80000000u | 0x7fffu);
      if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned old = atomicAdd(&a.ctrl[1], 1u);
    if (old == gridDim.x - 1) {
      for (int i = 0; i < 8; i++) __hip_atomic_store(&a.xcnt[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.ctrl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.ctrl[0], epoch + (unsigned)(T + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.guard) __hip_atomic_fetch_add(a.guard + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------
// launcher
// -------------------------------------------------------------------------------------------------------------------
bool persist_xl_supported(const Dims &d, const PersistOpts &o) {
  // C: any multiple of 32 from 512 up to XL_C (round 6; smaller layers keep the one-copy-over-all-CUs launch: half of an XCC's workgroups
  // would idle, and the tests of that kernel live there)
  return o.xl != 0 && d.C % 32 == 0 && d.C >= 512 && d.C <= XL_C && d.S >= 9 && d.S <= 32 && d.R % 32 == 0 && d.R >= 32 && d.R <= 512 && d.T >= 3 &&
         d.T * d.S >= 256;
}
size_t persist_xl_gran_bytes() { return (size_t)8 * 2 * 4 * (XL_C / 2) * 8 + 64; }   // granules + the eight per-XCC counters

hipError_t launch_fwd_persist_xl(const Dims &d, const FwdPtrs &p, const unsigned short *wrm, float *out, int out_stride, void *gran, unsigned *ctrl,
                                 const PersistOpts &o, hipStream_t st, LaunchProbe pr) {
  if (!persist_xl_supported(d, o) || !wrm || !gran || !out) return hipErrorInvalidValue;
  PersistXlArgs a;
  a.C = d.C; a.R = d.R; a.S = d.S; a.T = d.T; a.sx = (d.S + 7) / 8;
  a.wrm = wrm; a.wr = p.wr; a.wrb = p.wr_bf16; a.wm = p.wm; a.out = out; a.out_stride = out_stride; a.next_r = p.next_r;
  a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.mm = p.mm; a.rr = p.rr;
  a.prev_c = p.prev_c; a.prev_r = p.prev_r; a.next_c = p.next_c;
  a.gran = static_cast<unsigned long long *>(gran);
  a.xcnt = reinterpret_cast<unsigned *>(static_cast<unsigned char *>(gran) + (size_t)8 * 2 * 4 * (XL_C / 2) * 8);
  a.ctrl = ctrl; a.guard = o.guard; a.hstat = o.hstat;
  a.spin_limit = o.spin_limit > 0 ? o.spin_limit : SPIN_LIMIT_DEFAULT;
  a.test_stall = o.test_stall_fwd;
  const size_t shm = (size_t)5 * XL_LD + (size_t)2 * 16 * 64 * 16 + 32;
  auto kern = d.C == XL_C ? k_fwd_persist_xl<true> : k_fwd_persist_xl<false>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (pr.start) hipExtLaunchKernelGGL(kern, dim3(256), dim3(1024), shm, st, pr.start, pr.stop, 0, a);
  else hipLaunchKernelGGL(kern, dim3(256), dim3(1024), shm, st, a);
  return hipGetLastError();
}

size_t persist_xl_bwd_gran_bytes() { return (size_t)8 * 2 * 4 * XL_C * 16 + 64; }

hipError_t launch_bwd_persist_xl(const Dims &d, const BwdPtrs &p, const unsigned short *wrmT, const float *P, void *gran, unsigned *ctrl,
                                 const PersistOpts &o, hipStream_t st, LaunchProbe pr) {
  if (!persist_xl_supported(d, o) || !wrmT || !P || !gran) return hipErrorInvalidValue;
  PersistXlBwdArgs a;
  a.C = d.C; a.S = d.S; a.T = d.T; a.sx = (d.S + 7) / 8;
  a.wrmT = wrmT; a.P = P; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.dgifo = p.dgifo; a.dc = p.dc; a.dgifo_h = p.dgifo_h;
  a.gran = static_cast<uint4 *>(gran);
  a.xcnt = reinterpret_cast<unsigned *>(static_cast<unsigned char *>(gran) + (size_t)8 * 2 * 4 * XL_C * 16);
  a.ctrl = ctrl; a.guard = o.guard; a.hstat = o.hstat;
  a.spin_limit = o.spin_limit > 0 ? o.spin_limit : SPIN_LIMIT_DEFAULT;
  a.test_stall = o.test_stall_bwd;
  const size_t shm = (size_t)5 * XL_LDB + (size_t)32 * 64 * 16 + 32;
  auto kern = d.C == XL_C ? k_bwd_persist_xl<true> : k_bwd_persist_xl<false>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (pr.start) hipExtLaunchKernelGGL(kern, dim3(256), dim3(1024), shm, st, pr.start, pr.stop, 0, a);
  else hipLaunchKernelGGL(kern, dim3(256), dim3(1024), shm, st, a);
  return hipGetLastError();
}

}  // namespace klstm
