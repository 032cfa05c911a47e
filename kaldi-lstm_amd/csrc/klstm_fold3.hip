// kaldi-lstm_amd/csrc/klstm_fold3.hip -- the fold product W_rm = W_gifo_r W_r_m on the 16-bit matrix cores at fp32 accuracy.
//
// Two operand formats (option "fold_bf16x3"): 2 (default) = two fp16 planes, three products; 1 = three bf16 planes, six products.
//
// klstm_fold.hip runs the product on v_mfma_f32_16x16x4_f32 (256 FLOP/clk/CU): 2.6 GFLOP at 800/512 = 16.7 us at that peak,
// 32 us measured.  The bf16 MFMA (v_mfma_f32_16x16x32_bf16) is 16x faster per instruction-clock, enough to pay for a
// three-way split of both operands:
//     a = a1 + a2 + a3,  a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)      (8 + 8 + 8 = 24 mantissa bits)
//     a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2)                        (terms below 2^-24 |a b| dropped)
// six bf16 products accumulated in fp32, smallest first: every partial product of two bf16 numbers is EXACT in fp32, so
// the only roundings are the fp32 accumulations (as in the fp32 MFMA) and the three dropped terms (~3 * 2^-24 relative to
// |a||b|, the size of one fp32 rounding).  tests/test_engine_gpu.py holds the result to the tolerance of the fp32 kernel.
//
// fp16 x 2 (NPL = 2, the default): a = a1 + a2 / 2048, a1 = fp16(a), a2 = fp16((a - a1) * 2048) (klstm_math.h f16_split2: 11 + 11
// mantissa bits, the residual scaled so that it stays a normal fp16 number), a b ~= a1 b1 + (a1 b2 + a2 b1) / 2048: THREE products,
// the cross terms in an accumulator set of their own (every partial product still exact in fp32), dropped: a2 b2 and the two
// truncations, ~7e-7 |a||b| per term -- half of what the fp32 accumulation of a 512-term sum rounds away anyway.  Measured: every
// parity test at unchanged tolerances, the 50-chunk drift run identical to the bf16 x 3 form (out 4.3e-6, in_diff 4.9e-6), the
// product 21.5 -> 15.2 us (K loop 26.9 k -> 15.9 k clocks: MFMA-bound, 576 instead of 1152 MFMAs per wave), planes 8 MB instead
// of 12.  Range: fp16 -- parameters beyond +-65504 overflow the first plane (W_rm then carries Inf / NaN, loudly); bf16 x 3 has the
// fp32 range.
//
//   k_split3        both operands -> three bf16 planes each (reads 8 MB, writes 12 MB, 4.5 us) -- only when the parameters changed
//                   outside an Update (set_params ...): both Update kernels write the planes from the tile they have just updated
//                   (klstm_kernels.hip: GradsUpdate::a3 / b3, bf16_split3_store4)
//   k_fold_bf16x3   128 x 96 tiles (2 x 2 MFMA waves of 64 x 48 = 4 x 3 MFMA blocks x 6 products), K in stages of 32:
//                   one stage = 3 planes x (128 + 96) rows x 64 B = 42 KB, brought in by LDS-DMA
//                   (global_load_lds_dwordx4: 16 rows x 64 B per wave instruction, no registers) issued by FOUR LOADER WAVES,
//                   three buffers, ONE barrier per stage; the MFMA waves read the next stage's operands into a second register
//                   set under the current stage's MFMAs (measurements: DESIGN.md 4b, tools/fold3_probe.hip).
//                   LDS rows are 64 B; the 16-byte k-group of row r sits in slot kg ^ ((-(r >> 2)) & 3)
//                   (swizzle applied on the SOURCE side of the DMA, whose destination is lane-linear): the ds_read_b128 of
//                   an MFMA operand (16 rows x 4 k-groups) touches every bank quad once per 16-lane group.
//                   800/512: 25 x 9 = 225 workgroups, one round of the 256 CUs, 88 % of them busy.
// Epilogue as in k_fold_direct: rows are read in gates-packed order (logical row 4*cell + gate <- stored row gate*C + cell)
// and the tile goes straight into the two packed operands of the folded chain (pk1 / pk2, layouts in klstm_fold.hip).
#include "klstm_kernels.h"
#include "klstm_math.h"

#include <hip/hip_ext.h>

#include <cstring>
#include <mutex>

namespace klstm {

#pragma clang fp contract(off)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

struct Split3Args {
  const float *src[2];
  unsigned short *dst[2];
  size_t plane[2];       // elements per plane
  size_t n8[2];          // groups of 8 elements
  int mode;              // 1: three bf16 planes, 2: two fp16 planes (klstm_math.h f16_split2), 3: one bf16 plane
};

__global__ __launch_bounds__(256) void k_split3(Split3Args a) {
  const size_t tot = a.n8[0] + a.n8[1];
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < tot; g += (size_t)gridDim.x * blockDim.x) {
    const int w = g >= a.n8[0];
    const size_t i = (w ? g - a.n8[0] : g) * 8;
    const float4 lo = *reinterpret_cast<const float4 *>(a.src[w] + i), hi = *reinterpret_cast<const float4 *>(a.src[w] + i + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u16x8_t h1, h2, h3;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      unsigned short b1, b2, b3 = 0;
      if (a.mode == 2) f16_split2(v[e], b1, b2);
      else if (a.mode == 3) { b1 = bf16_rne(v[e]); b2 = 0; }
      else bf16_split3(v[e], b1, b2, b3);
      h1[e] = b1; h2[e] = b2; h3[e] = b3;
    }
    *reinterpret_cast<u16x8_t *>(a.dst[w] + i) = h1;
    if (a.mode != 3) *reinterpret_cast<u16x8_t *>(a.dst[w] + a.plane[w] + i) = h2;
    if (a.mode == 1) *reinterpret_cast<u16x8_t *>(a.dst[w] + 2 * a.plane[w] + i) = h3;
  }
}

struct Fold3Args {
  int C, R;
  const float *wr, *wmT;      // the fp32 operands themselves (W_gifo_r [4C x R], W_r_m^T [C x R]): the range guard's redo path reads them
  unsigned *redo;             // host-mapped event counter of that path (or null)
  const unsigned short *a3;   // W_gifo_r split: [3][4C x R] bf16, rows in g,i,f,o blocks of C
  const unsigned short *b3;   // W_r_m^T split:  [3][C x R]
  size_t a_plane, b_plane;
  float4 *pk1; int nch1;
  float4 *pk2; int nch2;
  unsigned short *wl;         // (NPL = 1) instead of pk1 / pk2: W_rm as bf16, LOGICAL rows (4 cell + gate) x C columns -- klstm_persist_ms.hip reads 16 consecutive rows per workgroup
  unsigned short *wlT;        // (with wl, or null) the TRANSPOSE as well: [C columns of W_rm][4C logical rows] -- the backward chain's operand
  int nbn, nwg;
#ifdef KLSTM_FOLD3_TIMING
  long long *dbg;             // per workgroup: shader clocks at entry / first operands / end of the K loop / exit (tools/fold3_probe.hip)
#endif
};

// LW: four more waves (4..7, one per SIMD) issue all the LDS-DMA requests, the MFMA waves none.
// NODMA: timing experiment (tools/fold3_probe.hip), stages past the first NBUF - 1 are not requested.
// NPL: planes per operand: 3 = bf16 x 3 (six products), 2 = fp16 x 2 (three products, two accumulator sets; klstm_math.h f16_split2)
template <int MI, int NI, int NBUF, bool NODMA = false, bool LW = false, int NPL = 3>
__global__ __launch_bounds__(LW ? 512 : 256) void k_fold_bf16x3(Fold3Args a) {
  constexpr int BM = 32 * MI, BN = 32 * NI, WM = 16 * MI, WN = 16 * NI;
  constexpr int APL = BM * 64, BPL = BN * 64, STG = NPL * (APL + BPL);   // bytes: one plane tile of A / B, one stage
  constexpr int NA = NPL * (BM / 16), NQ = NA + NPL * (BN / 16);           // DMA instructions of a stage (1 KB each)
  constexpr int NJ = (NQ + 3) / 4;                                     // per wave
  constexpr int FLD = WM + 4;                                          // epilogue transpose: floats per column
  static_assert(NPL == 1 || 4 * WN * FLD * 4 <= NBUF * STG, "epilogue transpose does not fit the staging buffers");   // (NPL = 1: the launcher sizes the LDS for the transpose)
  static_assert(NBUF >= 2 && NBUF <= 4 && (NBUF - 2) * NJ <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kg = lane >> 4;
  const int C = a.C, R = a.R, nstage = R / 32;
  // XCD-aware order (workgroup w lands on XCD w % 8): XCD x gets a contiguous m-major range -- the 9 column tiles of a row
  // panel of A share one L2, all of B (2.4 MB) is in every L2
  const int cpx = (a.nwg + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * cpx + (int)(blockIdx.x >> 3);
  if (b >= a.nwg) return;
  const int m0w = (b / a.nbn) * BM, n0w = (b % a.nbn) * BN;
#ifdef KLSTM_FOLD3_TIMING
  const long long t_c0 = clock64(), t_w0 = wall_clock64();
#endif

  // ---- DMA sources: instruction q of a stage fills LDS bytes [q KB, q KB + 1 KB) = 16 rows of one plane; lane l brings
  // the 16 bytes of (row l >> 2, slot l & 3) = k-group slot ^ swizzle(row)
  const char *src[NJ];
  int qd[NJ];
  {
    const int r16 = lane >> 2, kgs = (lane & 3) ^ ((-(r16 >> 2)) & 3);
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      int q = (wave & 3) + 4 * j;
      if (q > NQ - 1) q = NQ - 1;                                       // (a duplicate of the last one: same bytes, same place)
      qd[j] = q;
      if (q < NA) {
        const int p = q / (BM / 16), x = m0w + (q % (BM / 16)) * 16 + r16;       // logical row 4*cell + gate
        const size_t row = (x >> 2) < C ? (size_t)(x & 3) * C + (x >> 2) : 0;
        src[j] = reinterpret_cast<const char *>(a.a3 + p * a.a_plane + row * R + kgs * 8);
      } else {
        const int p = (q - NA) / (BN / 16), n = n0w + ((q - NA) % (BN / 16)) * 16 + r16;
        src[j] = reinterpret_cast<const char *>(a.b3 + p * a.b_plane + (size_t)(n < C ? n : 0) * R + kgs * 8);
      }
    }
  }
  auto issue = [&](int s, int buf) {
    char *db = smem + buf * STG;
#pragma unroll
    for (int j = 0; j < NJ; j++)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + (size_t)s * 64), (lptr_t)(db + qd[j] * 1024), 16, 0, 0);
  };

  if (LW && wave >= 4) {                                               // loader waves: same barrier sequence as the MFMA waves
#pragma unroll
    for (int s = 0; s < NBUF - 1; s++)
      if (s < nstage) issue(s, s);
    int ib = NBUF - 1;
    for (int k = 0; k < nstage; k++) {
      const int ahead = nstage - 1 - k < NBUF - 2 ? nstage - 1 - k : NBUF - 2;
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NBUF >= 4 ? 2 * NJ : 0) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NBUF >= 3 ? NJ : 0) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (!NODMA && k + NBUF - 1 < nstage) issue(k + NBUF - 1, ib);
      ib = ib + 1 == NBUF ? 0 : ib + 1;
    }
    __syncthreads();                                                   // (the MFMA waves' barrier below)
  }
  const bool loader = LW && wave >= 4;
  const int tw = wave & 3, wr = tw >> 1, wc = tw & 1;                  // loader wave w + 4 shares the epilogue of MFMA wave w
  const int sw = (kg ^ ((-(i16 >> 2)) & 3)) * 16;
  const int aoff = (wr * WM + i16) * 64 + sw, boff = NPL * APL + (wc * WN + i16) * 64 + sw;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++) acc[mi][ni] = (f32x4){0, 0, 0, 0};

  // Stage k lives in buffer k % NBUF and is brought in NBUF - 1 stages ahead.  The operands of a stage are read from LDS
  // into one of two register sets while the MFMAs of the stage before run on the other (one wave per SIMD: nobody else
  // would cover the LDS reads -- measured without this, tools/fold3_probe.hip: 20.9 us with the refills removed, 2.7x the
  // MFMA time).
  // advance(k): this wave's requests for stage k have landed (those of later stages may still be in flight: vmcnt counts
  // in order) and its LDS reads of stage k-1 are complete; the barrier makes both true for everyone, so buffer (k-1) % NBUF
  // can take stage k + NBUF - 1; then the reads of stage k are issued.
  int rb = 0, ib = NBUF - 1;
  auto advance = [&](int k, bf16x8_t (&af)[NPL][MI], bf16x8_t (&bf)[NPL][NI]) {
    const int ahead = nstage - 1 - k < NBUF - 2 ? nstage - 1 - k : NBUF - 2;
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NBUF >= 4 ? 2 * NJ : 0) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NBUF >= 3 ? NJ : 0) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!LW && !NODMA && k + NBUF - 1 < nstage) issue(k + NBUF - 1, ib);
    const char *sb = smem + rb * STG;
    rb = rb + 1 == NBUF ? 0 : rb + 1;
    ib = ib + 1 == NBUF ? 0 : ib + 1;
#pragma unroll
    for (int p = 0; p < NPL; p++) {
#pragma unroll
      for (int mi = 0; mi < MI; mi++) af[p][mi] = *reinterpret_cast<const bf16x8_t *>(sb + p * APL + aoff + mi * 1024);
#pragma unroll
      for (int ni = 0; ni < NI; ni++) bf[p][ni] = *reinterpret_cast<const bf16x8_t *>(sb + p * BPL + boff + ni * 1024);
    }
  };
  f32x4 accx[NPL == 2 ? MI : 1][NPL == 2 ? NI : 1];   // fp16 x 2: the cross terms a1 b2 + a2 b1 (their planes carry a factor 2^11)
#pragma unroll
  for (int mi = 0; mi < (NPL == 2 ? MI : 1); mi++)
#pragma unroll
    for (int ni = 0; ni < (NPL == 2 ? NI : 1); ni++) accx[mi][ni] = (f32x4){0, 0, 0, 0};
  auto multiply = [&](const bf16x8_t (&af)[NPL][MI], const bf16x8_t (&bf)[NPL][NI]) {
    if constexpr (NPL == 1) {
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mi], bf[0][ni], acc[mi][ni], 0, 0, 0);
    } else if constexpr (NPL == 3) {
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest products first
#pragma unroll
      for (int t = 0; t < 6; t++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA[t]][mi], bf[PB[t]][ni], acc[mi][ni], 0, 0, 0);
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
          const f16x8_t a0 = __builtin_bit_cast(f16x8_t, af[0][mi]), a1 = __builtin_bit_cast(f16x8_t, af[1][mi]);
          const f16x8_t b0 = __builtin_bit_cast(f16x8_t, bf[0][ni]), b1 = __builtin_bit_cast(f16x8_t, bf[1][ni]);
          accx[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, accx[mi][ni], 0, 0, 0);
          accx[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, accx[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[mi][ni], 0, 0, 0);
        }
    }
  };
#ifdef KLSTM_FOLD3_TIMING
  long long t_c1 = 0, t_c2 = 0;
#endif
  // ---- epilogue (k_fold_direct's, for WM rows): acc[mi][ni] of lane (i16, kg) = rows 16mi + 4kg + (0..3) at column 16ni + i16
  const int m0 = m0w + wr * WM, n0 = n0w + wc * WN;
  float *cs = reinterpret_cast<float *>(smem) + tw * (WN * FLD);
  if (!loader) {
    if (!LW) {
#pragma unroll
      for (int s = 0; s < NBUF - 1; s++)
        if (s < nstage) issue(s, s);
    }
    bf16x8_t af0[NPL][MI], bf0[NPL][NI], af1[NPL][MI], bf1[NPL][NI];
    advance(0, af0, bf0);
#ifdef KLSTM_FOLD3_TIMING
    t_c1 = clock64();
#endif
    for (int s = 0; s < nstage; s += 2) {
      if (s + 1 < nstage) advance(s + 1, af1, bf1);
      multiply(af0, bf0);
      if (s + 1 < nstage) {
        if (s + 2 < nstage) advance(s + 2, af0, bf0);
        multiply(af1, bf1);
      }
    }
#ifdef KLSTM_FOLD3_TIMING
    t_c2 = clock64();
#endif
    __syncthreads();                                                   // the staging buffers become the transpose buffers
    if constexpr (NPL == 2) {
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) acc[mi][ni] = acc[mi][ni] + accx[mi][ni] * (1.f / 2048.f);   // (exact scaling, one rounding)
    }
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
        *reinterpret_cast<float4 *>(cs + (16 * ni + i16) * FLD + 16 * mi + 4 * kg) =
            make_float4(acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w);
    if constexpr (NPL == 2) {
      // range guard (klstm_math.h): a parameter beyond the fp16 range left Inf / NaN in this wave's tile -> the tile again in plain
      // fp32 from the fp32 matrices, written over the transpose buffer (same wave: the LDS queue keeps the order; no barrier in
      // here, the other waves go on to the workgroup barrier below)
      float probe = 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
          for (int r = 0; r < 4; r++) probe = nonfinite_probe(probe, acc[mi][ni][r]);
      if (wave_any(probe != probe)) {
        redo_note(a.redo);
#pragma unroll 1
        for (int q = 0; q < MI * NI * 4; q++) {
          const int mi = q / (NI * 4), ni = (q >> 2) % NI, r = q & 3;
          const int n = n0 + 16 * ni + i16, x = m0 + 16 * mi + 4 * kg + r;   // x: logical row 4*cell + gate
          cs[(16 * ni + i16) * FLD + 16 * mi + 4 * kg + r] =
              ((x >> 2) < C && n < C) ? redo_dot(a.wr + ((size_t)(x & 3) * C + (x >> 2)) * R, 1, a.wmT + (size_t)n * R, 1, R) : 0.f;
        }
      }
    }
  }
  if (LW) __syncthreads();                                             // with loader waves the two packed operands are written by different waves
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int cell0 = m0 >> 2;
  if (a.wl) {
    // bf16 operand mode: rows in LOGICAL order (row m0 + 16 tl + i = 4 cell + gate), four consecutive columns per lane -> 8 bytes of bf16
    if (!loader)
#pragma unroll
      for (int it = 0; it < MI * NI; it++) {
        const int tl = it / NI, nq = (it % NI) * 4 + kg, i = i16;
        const int nl = 4 * nq, n = n0 + nl, x = m0 + tl * 16 + i;
        const float *cp = cs + nl * FLD + tl * 16 + i;
        if ((x >> 2) < C && n < C)
          *reinterpret_cast<uint2 *>(a.wl + (size_t)x * C + n) = make_uint2(bf16_rne(cp[0]) | ((unsigned)bf16_rne(cp[FLD]) << 16),
                                                                           bf16_rne(cp[2 * FLD]) | ((unsigned)bf16_rne(cp[3 * FLD]) << 16));
      }
    if (!loader && a.wlT) {
      // the transpose from the same tile: piece = (column, four consecutive logical rows) = 16 bytes of the column-major tile -> 8 bytes of bf16
#pragma unroll
      for (int it = 0; it < WM * WN / 4 / 64; it++) {
        const int p = it * 64 + lane, col = p / (WM / 4), rq = p % (WM / 4);
        const float4 v = *reinterpret_cast<const float4 *>(cs + col * FLD + 4 * rq);
        const int n = n0 + col, x = m0 + 4 * rq;
        if ((x >> 2) < C && n < C)
          *reinterpret_cast<uint2 *>(a.wlT + (size_t)n * 4 * C + x) = make_uint2(bf16_rne(v.x) | ((unsigned)bf16_rne(v.y) << 16),
                                                                                bf16_rne(v.z) | ((unsigned)bf16_rne(v.w) << 16));
      }
    }
    return;
  }
  // gates operand: piece = (16-row tile tl, column quad nq, row i): 16 consecutive float4 (256 B) per (tl, nq)
  if (!loader)
#pragma unroll
  for (int it = 0; it < MI * NI; it++) {
    const int tl = it / NI, nq = (it % NI) * 4 + kg, i = i16;
    const int nl = 4 * nq, n = n0 + nl, cell = cell0 + tl * 4 + (i >> 2);
    const float *cp = cs + nl * FLD + tl * 16 + i;
    const float4 v = make_float4(cp[0], cp[FLD], cp[2 * FLD], cp[3 * FLD]);
    if (cell < C && n < C)
      a.pk1[(((size_t)(cell >> 2) * a.nch1 + (n >> 5)) * 2 + ((n & 7) >> 2)) * 64 + ((n & 31) >> 3) * 16 + i] = v;
  }
  // d_m operand: piece = (column quad ct, gate, cell quad kq, cq = column % 4): 4 cells of one gate at one column
  constexpr int NKQ = WM / 16, CTS = 4 / NKQ;                         // cell quads of the wave's rows; column quads per pass
  if (!LW || loader)
#pragma unroll
  for (int it = 0; it < WN / 4 / CTS; it++) {
    const int cq = lane & 3, kq = (lane >> 2) % NKQ, gate = (lane / (4 * NKQ)) & 3, ct = it * CTS + lane / (16 * NKQ);
    const int c = n0 + ct * 4 + cq, cell = cell0 + kq * 4;
    const float *cp = cs + (ct * 4 + cq) * FLD + kq * 16 + gate;
    const float4 v = make_float4(cp[0], cp[4], cp[8], cp[12]);
    const int k = gate * C + cell;
    if (a.pk2 && c < C && cell < C)
      a.pk2[(((size_t)(c >> 2) * a.nch2 + (k >> 7)) * 2 + ((k >> 6) & 1)) * 64 + ((k & 63) >> 2) * 4 + cq] = v;
  }
#ifdef KLSTM_FOLD3_TIMING
  if (tid == 0) {
    long long *q = a.dbg + (size_t)blockIdx.x * 8;
    q[0] = t_c1 - t_c0; q[1] = t_c2 - t_c1; q[2] = clock64() - t_c2; q[3] = wall_clock64() - t_w0; q[4] = t_w0;
  }
#endif
}

#ifdef KLSTM_FOLD3_TIMING
static long long *g_fold3_dbg = nullptr;
#endif
static int g_fold_bf16x3 = 2;      // default mode of new engines: 0: fp32 MFMA kernel (klstm_fold.hip), 1: bf16 x 3, 2: fp16 x 2
void set_fold_bf16x3(int v) { g_fold_bf16x3 = v < 0 ? 0 : v > 2 ? 2 : v; }
int fold_default_mode() { return g_fold_bf16x3; }
bool fold_bf16x3_supported(const Dims &d, int mode) { return mode != 0 && d.C % 4 == 0 && d.R % 32 == 0; }

// ---- the range guard's state (klstm_kernels.h): per guard REDO_WORDS host-mapped event words (portable across devices), what has been
// answered of them, and the cool-down of every product family ----
struct RangeGuard {
  unsigned *host = nullptr, *dev = nullptr;
  unsigned seen[REDO_WORDS] = {0};
  long events[REDO_WORDS] = {0};
  long cool[REDO_WORDS] = {0};          // looks left on the fp32-range kernel
  long cool_len[REDO_WORDS] = {0};      // length of the cool-down in progress / the last one
  long clean[REDO_WORDS] = {0};         // fp16-plane looks since the last re-arm
  bool enabled = true;
  std::mutex mu;
};
static long guard_base_len(int which) { return which == REDO_FOLD ? 64 : 2048; }   // looks: one per fold product (= per Update); a few per stateless call
static void (*g_guard_note)(const char *) = nullptr;
void range_guard_set_note(void (*fn)(const char *)) { g_guard_note = fn; }
RangeGuard *range_guard_create() {
  RangeGuard *g = new RangeGuard();
  void *hp = nullptr, *dp = nullptr;
  if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return g; }   // (no words: nothing is counted)
  memset(hp, 0, 64);
  if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(hp); return g; }
  g->host = static_cast<unsigned *>(hp); g->dev = static_cast<unsigned *>(dp);
  return g;
}
void range_guard_destroy(RangeGuard *g) {
  if (!g) return;
  if (g->host) (void)hipHostFree(g->host);
  delete g;
}
static thread_local RangeGuard *t_guard = nullptr;
RangeGuard *range_guard_exchange(RangeGuard *g) { RangeGuard *p = t_guard; t_guard = g; return p; }
static RangeGuard *device_default_guard() {
  static std::mutex mu;
  static RangeGuard *per_device[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
  std::lock_guard<std::mutex> lk(mu);
  if (!per_device[dev]) per_device[dev] = range_guard_create();
  return per_device[dev];
}
static RangeGuard *current_guard() { return t_guard ? t_guard : device_default_guard(); }
unsigned *redo_counters() { return current_guard()->dev; }
void range_guard_reset(RangeGuard *g, bool enabled) {
  if (!g) g = device_default_guard();
  std::lock_guard<std::mutex> lk(g->mu);
  for (int i = 0; i < REDO_WORDS; i++) {
    if (g->host) reinterpret_cast<volatile unsigned *>(g->host)[i] = 0u;
    g->seen[i] = 0; g->events[i] = 0; g->cool[i] = 0; g->cool_len[i] = 0; g->clean[i] = 0;
  }
  g->enabled = enabled;
}
long range_guard_events(RangeGuard *g, int which) {
  if (!g) g = device_default_guard();
  if (which < 0 || which >= REDO_WORDS) return 0;
  std::lock_guard<std::mutex> lk(g->mu);
  const unsigned c = g->host ? reinterpret_cast<volatile unsigned *>(g->host)[which] : 0u;
  return g->events[which] + (long)(c - g->seen[which]);
}
unsigned redo_count(int which) {
  RangeGuard *g = current_guard();
  if (which < 0 || which >= REDO_WORDS) return 0u;
  static const char *names[REDO_WORDS] = {"the fold product W_gifo_r W_r_m", "the wide output-layer product (klstm_affine_propagate)",
                                          "the wide gradient product (klstm_affine_gradient / _update)",
                                          "the skinny products (klstm_affine_backpropagate, d_r / in_diff of a wide layer)", "?", "?", "?", "?"};
  char msg[320];
  msg[0] = 0;
  unsigned ret;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (!g->enabled) return 1u;
    const unsigned c = g->host ? reinterpret_cast<volatile unsigned *>(g->host)[which] : 0u;
    if (c != g->seen[which]) {                       // new events: (re-)start the cool-down
      g->events[which] += (long)(c - g->seen[which]);
      g->seen[which] = c;
      const long base = guard_base_len(which);
      if (g->cool_len[which] < base || g->clean[which] >= g->cool_len[which]) g->cool_len[which] = base;
      else g->cool_len[which] = g->cool_len[which] >= (1L << 19) ? (1L << 20) : 2 * g->cool_len[which];
      g->cool[which] = g->cool_len[which];
      g->clean[which] = 0;
      snprintf(msg, sizeof(msg), "range guard: %s met an operand beyond the fp16 range (recomputed in fp32 where it mattered); it stays on its "
               "fp32-range kernel for the next %ld looks, then the fp16 planes are tried again", names[which], g->cool[which]);
    }
    if (g->cool[which] > 0) {
      g->cool[which]--;
      ret = 1u;
      if (g->cool[which] == 0 && !msg[0]) snprintf(msg, sizeof(msg), "range guard: %s is back on its fp16 planes", names[which]);
    } else { g->clean[which]++; ret = 0u; }
  }
  if (msg[0] && g_guard_note) g_guard_note(msg);
  return ret;
}
size_t fold_bf16x3_scratch_bytes(const Dims &d) { return (size_t)3 * 5 * d.C * d.R * sizeof(unsigned short); }

void fold_bf16x3_planes(const Dims &d, void *scratch, unsigned short **a3, long *a_plane, unsigned short **b3, long *b_plane) {
  *a3 = static_cast<unsigned short *>(scratch);
  *a_plane = (long)4 * d.C * d.R; *b_plane = (long)d.C * d.R;
  *b3 = *a3 + 3 * *a_plane;
}

template <int NPL>
static hipError_t launch_fold_planes(const Fold3Args &a, hipStream_t st, LaunchProbe pr) {
  constexpr int MI = 4, NI = 3, NBUF = 3;
  constexpr int shm = NBUF * NPL * (32 * MI + 32 * NI) * 64 > 4 * 16 * NI * (16 * MI + 4) * 4 ? NBUF * NPL * (32 * MI + 32 * NI) * 64
                                                                                              : 4 * 16 * NI * (16 * MI + 4) * 4;
  {   // (per launch, like every other kernel here: the attribute belongs to the current DEVICE, engines may sit on several)
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fold_bf16x3<MI, NI, NBUF, false, true, NPL>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    if (err != hipSuccess) return err;
  }
  const dim3 grid((a.nwg + 7) / 8 * 8), block(512);
  if (pr.start) hipExtLaunchKernelGGL((k_fold_bf16x3<MI, NI, NBUF, false, true, NPL>), grid, block, shm, st, pr.start, pr.stop, 0, a);
  else hipLaunchKernelGGL((k_fold_bf16x3<MI, NI, NBUF, false, true, NPL>), grid, block, shm, st, a);
  return hipGetLastError();
}

// The fold product of the bf16 operand mode: W_rm = bf16(W_gifo_r) bf16(W_r_m) with fp32 accumulation, stored as bf16 in logical-row
// order (wl, [4C x C]).  scratch: plane 0 of each operand (mode 3 of the split; written by the Update when planes_fresh).
// 128 x 128 tiles: 256 workgroups at 1024 / 512 = one round of the chip.
hipError_t launch_fold_ms(const Dims &d, const float *wr, const float *wmT, void *scratch, unsigned short *wl, hipStream_t st,
                          LaunchProbe pr_split, LaunchProbe pr, bool planes_fresh, unsigned short *wlT) {
  constexpr int MI = 4, NI = 4, NBUF = 3;
  unsigned short *a3 = static_cast<unsigned short *>(scratch);
  const size_t apl = (size_t)4 * d.C * d.R, bpl = (size_t)d.C * d.R;
  unsigned short *b3 = a3 + 3 * apl;
  if (!planes_fresh) {
    Split3Args s;
    s.src[0] = wr; s.src[1] = wmT; s.dst[0] = a3; s.dst[1] = b3; s.plane[0] = apl; s.plane[1] = bpl;
    s.n8[0] = apl / 8; s.n8[1] = bpl / 8;
    s.mode = 3;
    const unsigned sgrid = (unsigned)std::min<size_t>((s.n8[0] + s.n8[1] + 255) / 256, 2048);
    if (pr_split.start) hipExtLaunchKernelGGL(k_split3, dim3(sgrid), dim3(256), 0, st, pr_split.start, pr_split.stop, 0, s);
    else hipLaunchKernelGGL(k_split3, dim3(sgrid), dim3(256), 0, st, s);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
  }
  Fold3Args a;
#ifdef KLSTM_FOLD3_TIMING
  a.dbg = g_fold3_dbg;
#endif
  a.C = d.C; a.R = d.R; a.wr = wr; a.wmT = wmT; a.redo = nullptr;
  a.a3 = a3; a.b3 = b3; a.a_plane = apl; a.b_plane = bpl;
  a.pk1 = nullptr; a.nch1 = 0; a.pk2 = nullptr; a.nch2 = 0; a.wl = wl; a.wlT = wlT;
  a.nbn = (d.C + 32 * NI - 1) / (32 * NI);
  a.nwg = ((4 * d.C + 32 * MI - 1) / (32 * MI)) * a.nbn;
  constexpr int stage = NBUF * 1 * (32 * MI + 32 * NI) * 64, transp = 4 * 16 * NI * (16 * MI + 4) * 4;
  constexpr int shm = stage > transp ? stage : transp;
  auto kern = k_fold_bf16x3<MI, NI, NBUF, false, true, 1>;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, shm);
  if (err != hipSuccess) return err;
  const dim3 grid((a.nwg + 7) / 8 * 8), block(512);
  if (pr.start) hipExtLaunchKernelGGL(kern, grid, block, shm, st, pr.start, pr.stop, 0, a);
  else hipLaunchKernelGGL(kern, grid, block, shm, st, a);
  return hipGetLastError();
}

hipError_t launch_fold_bf16x3(const Dims &d, int mode, const float *wr, const float *wmT, void *scratch, float *pk_fold[2], int nch1,
                              int nch2, hipStream_t st, LaunchProbe pr_split, LaunchProbe pr, bool planes_fresh) {
  constexpr int MI = 4, NI = 3;
  unsigned short *a3 = static_cast<unsigned short *>(scratch);
  const size_t apl = (size_t)4 * d.C * d.R, bpl = (size_t)d.C * d.R;
  unsigned short *b3 = a3 + 3 * apl;
  Split3Args s;
  s.src[0] = wr; s.src[1] = wmT; s.dst[0] = a3; s.dst[1] = b3; s.plane[0] = apl; s.plane[1] = bpl;
  s.n8[0] = apl / 8; s.n8[1] = bpl / 8;
  s.mode = mode == 2 ? 2 : 1;
  const unsigned sgrid = (unsigned)std::min<size_t>((s.n8[0] + s.n8[1] + 255) / 256, 2048);
  if (!planes_fresh) {
    if (pr_split.start) hipExtLaunchKernelGGL(k_split3, dim3(sgrid), dim3(256), 0, st, pr_split.start, pr_split.stop, 0, s);
    else hipLaunchKernelGGL(k_split3, dim3(sgrid), dim3(256), 0, st, s);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;

  Fold3Args a;
#ifdef KLSTM_FOLD3_TIMING
  a.dbg = g_fold3_dbg;
#endif
  a.C = d.C; a.R = d.R; a.wr = wr; a.wmT = wmT; a.redo = redo_counters() ? redo_counters() + REDO_FOLD : nullptr;
  a.a3 = a3; a.b3 = b3; a.a_plane = apl; a.b_plane = bpl;
  a.pk1 = reinterpret_cast<float4 *>(pk_fold[0]); a.nch1 = nch1;
  a.pk2 = reinterpret_cast<float4 *>(pk_fold[1]); a.nch2 = nch2; a.wl = nullptr; a.wlT = nullptr;
  a.nbn = (d.C + 32 * NI - 1) / (32 * NI);
  a.nwg = ((4 * d.C + 32 * MI - 1) / (32 * MI)) * a.nbn;
  return s.mode == 2 ? launch_fold_planes<2>(a, st, pr) : launch_fold_planes<3>(a, st, pr);
}

}  // namespace klstm
