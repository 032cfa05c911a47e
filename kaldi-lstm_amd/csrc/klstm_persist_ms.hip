// kaldi-lstm_amd/csrc/klstm_persist_ms.hip -- weights-RESIDENT forward chain for MANY streams (9 .. 32 per GPU) in bf16 operand
// mode (engine option "bf16"; BASELINE.json configs[4]: 3 x LstmProjectedStreams 1024 / 512, 32 streams per GPU).  VERDICT r03
// next #4: "build the many-stream weights-resident forward chain -- stop sizing it".
//
// The launch-per-step chain spends 4.2-5.0 us per step KERNEL at 32 streams -- the price of a dependent launch that touches
// memory, not of the step's work (docs/DESIGN_rounds_1-4.md 9 item 1): two of them per forward step (gates :275-309, projection :312).  Here ONE
// launch runs all T steps of the folded recurrence
//     a(t) = [x(t) W_gifo_x^T + b] + W_rm m(t-1),   W_rm = W_gifo_r W_r_m        (:275 with r(t-1) = W_r_m m(t-1), :312)
// for up to 32 streams: the x term is the batched product of the reference (:246, :259; already in the gifo plane when the launch
// starts), r(t) = W_r_m m(t) (:312; the recurrence no longer needs it) rides along on the first R / 16 workgroups, and W_rm is one
// bf16 product per Update (4C x C over K = R: both operands rounded to bf16 like every other operand of this mode).
//   * a workgroup owns 4 cells = 16 rows of W_rm (row 4 cell + gate), 32 KB of bf16 at C = 1024, RESIDENT in the registers of its
//     four contraction waves (a quarter of K each) for the whole minibatch: C / 4 <= 256 workgroups, one per CU;
//   * v_mfma_f32_16x16x32_bf16 with the weights on the M side and 16 streams on the N side: the result lane (stream, cell) holds
//     g, i, f, o of ITS (cell, stream) -- the cell update (:278-309) is lane-local, as in the step kernels (klstm_math.h);
//   * the per-step all-to-all (every workgroup needs all of m(t-1): S x C bf16) inside the launch, the transport of
//     klstm_persist.hip: data-tagged granules, here 16 bytes = {tag, 6 x bf16} (tools/xchg_probe "wide": 4.0 us per step for 32
//     streams x 1024 cells between 256 workgroups), one sc1 store per granule, swept with 16-byte sc1 loads until every tag matches;
//     two parity slots; tags = epoch + t; every wait bounded; a give-up is recorded and answered like the small chain's
//     (klstm_engine.hip recover());
//   * the B operand of a step = m(t-1) of all cells as bf16, [stream][cell] in LDS (rows padded by 16 bytes: conflict-free
//     ds_read_b128), written by the sweeper waves straight from the granules.
// Step 1 closes over the CARRIED r (possibly produced under older weights): contracted against the natural W_gifo_r rows
// (K = R), which die after it.
// Rounding = that of the bf16 operand mode: weights and the staged activations (m, r(0)) to bf16 (RNE), fp32 accumulate, planes
// fp32; W_rm itself is a bf16 product with fp32 accumulation, rounded to bf16 when it is stored (tests/bf16_emul.py fold = True).
#include "klstm_kernels.h"
#include "klstm_math.h"
#include "klstm_persist_dev.h"

#include <hip/hip_ext.h>

namespace klstm {

#pragma clang fp contract(off)

typedef __bf16 ms_bf16x8 __attribute__((ext_vector_type(8)));

struct PersistMsArgs {
  int C, R, S, T;
  int ldrow;                      // bytes per LDS slab row: 2 max(C, R) + 16
  const unsigned short *wrm;      // folded W_rm as bf16, LOGICAL rows (4 cell + gate) x C: the 16 rows of workgroup wg are rows 16 wg .. 16 wg + 15 (launch_fold_ms)
  const float *wr;                // natural W_gifo_r [4C x R] (step 1)
  const float *wm;                // natural W_r_m [R x C]: r(t) = W_r_m m(t) (:312) is contracted here too, 16 rows per workgroup (the first R / 16)
  float *out; int out_stride;     // output rows [T*S x R] (:328)
  float *next_r;                  // r(T) (:331)
  const float *pi, *pf, *po;
  float *gifo, *cc, *hh, *mm, *rr; // planes; gifo rows of frames 1..T hold x W_gifo_x^T + bias on entry
  const float *prev_c, *prev_r;   // carried state the minibatch starts from
  float *next_c;                  // c(T) (:331)
  uint4 *gran;                    // [2 parities][workgroups][NT][11] granules {tag, 6 x bf16}
  unsigned *ctrl;                 // [0] epoch, [1] finished workgroups, [2] status, [3] ordinal of the launch that gave up
  unsigned *guard;                // the engine's control words (klstm_kernels.h PersistOpts)
  unsigned *hstat;
  long long spin_limit;
  int test_stall;                 // test hook: workgroup 0 withholds its publishes of this step
};

constexpr int MS_NG = 11;         // granules per (workgroup, 16-stream tile): 64 values at 6 per granule

__device__ __forceinline__ ms_bf16x8 ms_load8(const float *p, bool on) {
  const float4 lo = on ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 hi = on ? *reinterpret_cast<const float4 *>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  return (ms_bf16x8){(__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w, (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w};
}

// NT: 16-stream tiles (S <= 16 NT); NSW: sweeper waves; PG: granules per sweeper thread (>= workgroups * NT * 11 / (64 NSW))
template <int NT, int NSW, int PG>
__global__ __launch_bounds__((4 + NSW) * 64) void k_fwd_persist_ms(PersistMsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int C = a.C, R = a.R, S = a.S, T = a.T, ldrow = a.ldrow;
  unsigned char *slab = smem;                                         // [16 NT][ldrow]: row s = m(t-1)[s][0..C) (step 1: r(0)[s][0..R)) as bf16
  f32x4 *part = reinterpret_cast<f32x4 *>(smem + 16 * NT * ldrow);    // [4 waves][NT][64]: partial gate tiles
  f32x4 *partr = part + 4 * NT * 64;                                  // [4 waves][NT][64]: partial projection tiles
  unsigned short *mst = reinterpret_cast<unsigned short *>(partr + 4 * NT * 64);   // [NT][72]: m(t) of the own 4 cells x 16 streams, value v = 4 stream + cell
  unsigned *abortf = reinterpret_cast<unsigned *>(mst + NT * 72);
  int *pubcnt = reinterpret_cast<int *>(abortf + 1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x, nwg = gridDim.x;
  const unsigned epoch = __hip_atomic_load(&a.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // queued behind a launch that gave up: do nothing (klstm_persist.hip; looked at behind each role's own prologue loads)
  unsigned behind_giveup = 0u;
  if (a.guard) behind_giveup = __hip_atomic_load(&a.guard[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                               __hip_atomic_load(&a.guard[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = tid * 16; i < 16 * NT * ldrow; i += (4 + NSW) * 64 * 16) *reinterpret_cast<uint4 *>(slab + i) = make_uint4(0u, 0u, 0u, 0u);
  if (tid < NT * 72) mst[tid] = 0;
  if (tid == 0) { *abortf = 0u; *pubcnt = 0; }
  __syncthreads();
  const int granules_per_step = nwg * NT * MS_NG;

  if (wave < 4) {
    // =========================== contraction waves (waves 0 .. NT-1 also own the cell update of stream tile `wave`) ===========================
    const int i16 = lane & 15, kg = lane >> 4;
    // operand row of this lane: tile row i16 = 4 cell + gate  ->  stored row gate C + 4 wg + cell
    const size_t arow = (size_t)(i16 & 3) * C + 4 * wg + (i16 >> 2);
    const int nchU = R / 32, cwU = (nchU + 3) / 4, nch = C / 32, cw = (nch + 3) / 4;
    // epilogue lane = (stream 16 wave + i16, cell kg)
    const int s = 16 * wave + i16, cellg = 4 * wg + kg;
    const bool on = wave < NT && s < S;
    const int sc = on ? s : 0;
    float cp = on ? a.prev_c[(size_t)sc * C + cellg] : 0.f;            // carried c(0) (:231)
    if (on) a.cc[(size_t)sc * C + cellg] = cp;                         // time block 0 of the c plane: BPTT reads it
    const float wpi = a.pi[cellg], wpf = a.pf[cellg], wpo = a.po[cellg];
    const bool skip = __builtin_amdgcn_readfirstlane(behind_giveup) != 0u;   // (behind the prev_c load: it is here; in front of the weight requests)
    ms_bf16x8 uf[8], af[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int cu = wave * cwU + j;
      uf[j] = ms_load8(a.wr + arow * R + 32 * cu + 8 * kg, j < cwU && cu < nchU);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int cf = wave * cw + j;
      af[j] = (j < cw && cf < nch) ? *reinterpret_cast<const ms_bf16x8 *>(a.wrm + (size_t)(16 * wg + i16) * C + 32 * cf + 8 * kg)
                                   : (ms_bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    const unsigned char *brow = slab + i16 * ldrow + 16 * kg;
    // r(t-1) = W_r_m m(t-1) (:312) rides along from step 2 on: the slab of step t IS m(t-1) of all cells.  The first R / 16
    // workgroups hold 16 rows of W_r_m each (row 16 wg + i16, this wave's K quarter: the registers the step-1 rows leave behind),
    // 2 NT more MFMAs per chunk; wave NT + ti adds the four partial tiles of stream tile ti and writes the r plane rows, the output
    // rows (:328) and, for frame T, the carried r (:331).  One more exchange (m(T)) and one more pass (t = T + 1) for r(T).
    const bool projw = 16 * wg < R;
    const int prow = projw ? 16 * wg + i16 : 0;
    ms_bf16x8 rf[8];
    // one step: barrier (1), contraction of this wave's K quarter against the slab, barrier (2), cell update on waves < NT.
    // Returns false when the launch is over (a wait expired somewhere, or this launch sits behind one that gave up).
    auto run_step = [&](int t, const ms_bf16x8 (&w)[8], int c0, int clast) -> bool {
      float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);                     // x(t) W_gifo_x^T + b of this lane's (cell, stream): the batched product left it in the plane
      if (on && t <= T) {
        const float *gp = a.gifo + ((size_t)t * S + s) * 4 * C + cellg;
        xg = make_float4(gp[0], gp[C], gp[2 * C], gp[3 * C]);
      }
      lds_barrier();                                                   // (1) slab of step t ready
      if (*abortf || skip) return false;
      f32x4 acc[NT], accr[NT];
#pragma unroll
      for (int ti = 0; ti < NT; ti++) { acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f}; accr[ti] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      const bool do_r = projw && t >= 2;
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {                                 // (two halves of four chunks: 4 NT operand registers in flight instead of 8 NT)
        ms_bf16x8 bv[4][NT];
#pragma unroll
        for (int j = 0; j < 4; j++) {                                  // (chunks past the operand: zero weights against the last chunk's columns)
          const int ch = c0 + 4 * hf + j < clast ? c0 + 4 * hf + j : clast;
#pragma unroll
          for (int ti = 0; ti < NT; ti++) bv[j][ti] = *reinterpret_cast<const ms_bf16x8 *>(brow + 16 * ti * ldrow + 64 * ch);
        }
        __builtin_amdgcn_sched_barrier(0);                             // (all reads of the half issued before its first MFMA)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int ti = 0; ti < NT; ti++) acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[4 * hf + j], bv[j][ti], acc[ti], 0, 0, 0);
        if (do_r) {
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int ti = 0; ti < NT; ti++) accr[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf[4 * hf + j], bv[j][ti], accr[ti], 0, 0, 0);
        }
      }
#pragma unroll
      for (int ti = 0; ti < NT; ti++) {
        part[(wave * NT + ti) * 64 + lane] = acc[ti];
        if (do_r) partr[(wave * NT + ti) * 64 + lane] = accr[ti];
      }
      lds_barrier();                                                   // (2) the four partial tiles are in LDS
      if (do_r && wave >= NT && wave < 2 * NT) {
        // ---- r(t-1) of stream tile ti: lane (stream 16 ti + i16, rows 16 wg + 4 kg .. + 3) ----
        const int ti = wave - NT, sr = 16 * ti + i16, f = t - 1;
        const f32x4 v = ((partr[(0 * NT + ti) * 64 + lane] + partr[(1 * NT + ti) * 64 + lane]) + partr[(2 * NT + ti) * 64 + lane]) +
                        partr[(3 * NT + ti) * 64 + lane];
        if (sr < S) {
          const int col = 16 * wg + 4 * kg;
          *reinterpret_cast<float4 *>(a.rr + ((size_t)f * S + sr) * R + col) = make_float4(v.x, v.y, v.z, v.w);
          float *op = a.out + ((size_t)(f - 1) * S + sr) * a.out_stride + col;
          op[0] = v.x; op[1] = v.y; op[2] = v.z; op[3] = v.w;
          if (f == T) *reinterpret_cast<float4 *>(a.next_r + (size_t)sr * R + col) = make_float4(v.x, v.y, v.z, v.w);
        }
      }
      if (wave < NT && t <= T) {
        // ---- cell update of (cell kg, stream s): the four K quarters in fixed order, then :278-309 ----
        const f32x4 v = ((part[(0 * NT + wave) * 64 + lane] + part[(1 * NT + wave) * 64 + lane]) + part[(2 * NT + wave) * 64 + lane]) +
                        part[(3 * NT + wave) * 64 + lane];
        float ai = v.y + xg.y, af_ = v.z + xg.z, ao = v.w + xg.w;
        const float ag = v.x + xg.x;
        ai += wpi * cp;                                                // :278
        af_ += wpf * cp;                                               // :281
        const float gi = k_sigmoid(ai), gf = k_sigmoid(af_), gg = k_tanh(ag);   // :284-288
        float c = gg * gi;                                             // :291
        c = c + cp * gf;                                               // :294
        c = c < -50.f ? -50.f : c;                                     // :296
        c = c > 50.f ? 50.f : c;                                       // :297
        const float h = k_tanh(c);                                     // :300
        ao += wpo * c;                                                 // :303
        const float go = k_sigmoid(ao);                                // :306
        const float m = h * go;                                        // :309
        // ---- publish m(t): the tile's 64 values (v = 4 stream + cell) as bf16 through LDS into 11 granules of 6 ----
        {                                                              // (m(T) travels too: r(T))
          mst[wave * 72 + 4 * i16 + kg] = on ? bf16_rne(m) : (unsigned short)0;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (same wave: the LDS queue keeps the order; the wait keeps the compiler from hoisting the reads)
          if (lane < MS_NG && !(a.test_stall == t && wg == 0)) {
            const unsigned *wp = reinterpret_cast<const unsigned *>(mst + wave * 72 + 6 * lane);
            const u32x4 g = {epoch + (unsigned)t, wp[0], wp[1], wp[2]};
            const __amdgpu_buffer_rsrc_t rs = buf_rsrc(a.gran, 2 * granules_per_step * 16);
            __builtin_amdgcn_raw_buffer_store_b128(g, rs, (((t & 1) * nwg + wg) * NT + wave) * MS_NG * 16 + lane * 16, 0, 16);   // one 16-byte sc1 store
          }
        }
        if (on) {
          float *gp = a.gifo + ((size_t)t * S + s) * 4 * C + cellg;
          gp[0] = gg; gp[C] = gi; gp[2 * C] = gf; gp[3 * C] = go;
          const size_t pc = ((size_t)t * S + s) * C + cellg;
          a.cc[pc] = c; a.hh[pc] = h; a.mm[pc] = m;
          if (t == T) a.next_c[(size_t)s * C + cellg] = c;             // :331 (c columns)
        }
        cp = c;
        if (lane == 0) __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // publishes of this step issued
      }
      return true;
    };
    if (run_step(1, uf, wave * cwU, nchU - 1)) {                        // step 1: the carried r against the natural W_gifo_r rows, dead afterwards
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int cf = wave * cw + j;
        rf[j] = ms_load8(a.wm + (size_t)prow * C + 32 * cf + 8 * kg, projw && j < cw && cf < nch);
      }
      for (int t = 2; t <= T + 1; t++)                                   // (t = T + 1: r(T) only)
        if (!run_step(t, af, wave * cw, nch - 1)) break;
    }
  } else {
    // =========================== sweepers: the B operand of every step into the slab ===========================
    constexpr int NST = NSW * 64;
    const int sidx = (wave - 4) * 64 + lane;
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(a.gran, 2 * granules_per_step * 16);
    for (int t = 1; t <= T + 1; t++) {                  // (t = T + 1: the slab of m(T) for r(T))
      if (t == 1) {
        // step 1: the carried r(0) (:231, :275), rounded like every staged activation of this mode; time block 0 of the r plane
        for (int i = sidx; i < S * (R / 4); i += NST) {
          const int s = i / (R / 4), k = (i % (R / 4)) * 4;
          const float4 rv = *reinterpret_cast<const float4 *>(a.prev_r + (size_t)s * R + k);
          *reinterpret_cast<uint2 *>(slab + s * ldrow + 2 * k) =
              make_uint2(bf16_rne(rv.x) | ((unsigned)bf16_rne(rv.y) << 16), bf16_rne(rv.z) | ((unsigned)bf16_rne(rv.w) << 16));
          if (wg == 0) *reinterpret_cast<float4 *>(a.rr + (size_t)s * R + k) = rv;
        }
      } else {
        // polling starts once this workgroup's OWN cell waves have issued their publishes of step t-1 (klstm_persist.hip)
        {
          const long long w0 = wall_clock64();
          for (unsigned spins = 0; __hip_atomic_load(pubcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NT * (t - 1); spins++) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 1023) == 1023 && wall_clock64() - w0 > a.spin_limit) break;
          }
        }
        const unsigned tag = epoch + (unsigned)(t - 1);
        const int base = ((t - 1) & 1) * granules_per_step;
        u32x4 q[PG];
        bool ok = false;
        const long long t0 = wall_clock64();
        // (every pass asks for ALL of the thread's granules again: asking only for the missing ones -- predicated loads -- measured
        //  slower, 100.6 vs 94.8 us per launch at 32 streams: the loads of a pass then leave one by one)
        for (unsigned spins = 0;; spins++) {
#pragma unroll
          for (int i = 0; i < PG; i++) {
            const int g = sidx + i * NST;
            q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base + (g < granules_per_step ? g : 0)) * 16, 0, 16);   // aux 16 = sc1
          }
          ok = true;
#pragma unroll
          for (int i = 0; i < PG; i++) ok &= (q[i].x == tag) | (sidx + i * NST >= granules_per_step);
          if (ok) break;
          if ((spins & 31) == 31 && wall_clock64() - t0 > a.spin_limit) break;
        }
        if (!ok) {
          *abortf = 1u;
          if (lane == 0) {
            atomicCAS(&a.ctrl[3], 0u, launch_ordinal(a.guard));
            atomicMax(&a.ctrl[2], 0x80000000u | (unsigned)t);
            if (a.hstat) __hip_atomic_store(a.hstat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        // (the slab is free: every contraction wave of this workgroup passed barrier (2) of step t-1 behind its reads)
#pragma unroll
        for (int i = 0; i < PG; i++) {
          const int g = sidx + i * NST;
          if (g >= granules_per_step) continue;
          const int gw = g / (NT * MS_NG), rem = g - gw * (NT * MS_NG), ti = rem / MS_NG, j = rem - ti * MS_NG;
          const unsigned w[3] = {q[i].y, q[i].z, q[i].w};
#pragma unroll
          for (int wi = 0; wi < 3; wi++) {
            const int vv = 6 * j + 2 * wi;                             // values vv, vv + 1 = (stream vv >> 2, cells vv & 3 and + 1)
            if (vv < 64) *reinterpret_cast<unsigned *>(slab + (16 * ti + (vv >> 2)) * ldrow + 2 * (4 * gw + (vv & 3))) = w[wi];
          }
        }
      }
      lds_barrier();                                                   // (1)
      if (*abortf || __builtin_amdgcn_readfirstlane(behind_giveup) != 0u) break;
      lds_barrier();                                                   // (2)
    }
  }
  finish(a.ctrl, epoch, T + 2, a.guard ? a.guard + 8 : nullptr, a.guard, a.hstat ? a.hstat + 1 : nullptr);
}

// -------------------------------------------------------------------------------------------------------------------
// launcher
// -------------------------------------------------------------------------------------------------------------------
static inline int msdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int MS_NSW = 8;

bool persist_ms_supported(const Dims &d) {
  return d.S >= 9 && d.S <= 32 && d.C % 32 == 0 && d.C <= 1024 && d.R % 32 == 0 && d.R <= 1024 && d.C >= 64 && d.R / 16 <= d.C / 4 &&
         d.T >= 3 && d.T * d.S >= 256;                                  // (the batched products of this mode run on the bf16 tiles from 256 frames on)
}
int persist_ms_grid(const Dims &d) { return d.C / 4; }
size_t persist_ms_gran_bytes(const Dims &d) {      // (the XCD-local kernel of klstm_persist_xl.hip uses the same buffer: the larger of the two)
  const size_t ms = (size_t)2 * (d.C / 4) * 2 * MS_NG * 16, xl = persist_xl_gran_bytes();
  return ms > xl ? ms : xl;
}

template <int NT, int PG>
static hipError_t ms_launch(const PersistMsArgs &a, int grid, size_t shm, hipStream_t st, LaunchProbe pr) {
  auto kern = k_fwd_persist_ms<NT, MS_NSW, PG>;
  if (shm > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (pr.start) hipExtLaunchKernelGGL(kern, dim3(grid), dim3((4 + MS_NSW) * 64), shm, st, pr.start, pr.stop, 0, a);
  else hipLaunchKernelGGL(kern, dim3(grid), dim3((4 + MS_NSW) * 64), shm, st, a);
  return hipGetLastError();
}

hipError_t launch_fwd_persist_ms(const Dims &d, const FwdPtrs &p, const unsigned short *wrm, float *out, int out_stride, uint4 *gran, unsigned *ctrl,
                                 const PersistOpts &o, hipStream_t st, LaunchProbe pr) {
  if (!persist_ms_supported(d) || !wrm || !gran || !out) return hipErrorInvalidValue;
  if (persist_xl_supported(d, o)) return launch_fwd_persist_xl(d, p, wrm, out, out_stride, gran, ctrl, o, st, pr);   // C = 1024: one chain per XCD
  PersistMsArgs a;
  a.C = d.C; a.R = d.R; a.S = d.S; a.T = d.T;
  a.ldrow = 2 * (d.C > d.R ? d.C : d.R) + 16;
  a.wrm = wrm; a.wr = p.wr; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.mm = p.mm; a.rr = p.rr;
  a.prev_c = p.prev_c; a.prev_r = p.prev_r; a.next_c = p.next_c;
  a.wm = p.wm; a.out = out; a.out_stride = out_stride; a.next_r = p.next_r;
  a.gran = gran; a.ctrl = ctrl; a.guard = o.guard; a.hstat = o.hstat;
  a.spin_limit = o.spin_limit > 0 ? o.spin_limit : SPIN_LIMIT_DEFAULT;
  a.test_stall = o.test_stall_fwd;
  const int nt = d.S > 16 ? 2 : 1, grid = persist_ms_grid(d);
  const int pg = msdiv(grid * nt * MS_NG, MS_NSW * 64);
  const size_t shm = (size_t)16 * nt * a.ldrow + (size_t)2 * 4 * nt * 64 * 16 + (size_t)nt * 72 * 2 + 16;
  if (nt == 2) {
    if (pg <= 4) return ms_launch<2, 4>(a, grid, shm, st, pr);
    if (pg <= 8) return ms_launch<2, 8>(a, grid, shm, st, pr);
    if (pg <= 11) return ms_launch<2, 11>(a, grid, shm, st, pr);
  } else {
    if (pg <= 3) return ms_launch<1, 3>(a, grid, shm, st, pr);
    if (pg <= 6) return ms_launch<1, 6>(a, grid, shm, st, pr);
  }
  return hipErrorInvalidValue;
}

}  // namespace klstm
