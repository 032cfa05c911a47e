// kaldi-lstm_amd/csrc/klstm_fold.hip -- the fold product W_rm = W_gifo_r W_r_m (4C x C over K = R, fp32 MFMA), once per Update.
//
// The recurrence of ...streams.h substitutes r(t-1) = W_r_m m(t-1) (:312) into the gates pre-activation (:275):
// W_gifo_r r(t-1) = (W_gifo_r W_r_m) m(t-1).  This product is the largest single piece of work outside the chain
// (2.6 GFLOP at 800/512: 16.7 us at the fp32 MFMA peak), and the generic 64x64-tile kernel (klstm_kernels.hip k_gemm)
// loses a third of it to tile quantisation alone: 650 tiles over 256 CUs = 3 tiles on the slowest CU against 2.54 on
// average, on top of a single-buffered LDS pipeline (tools/fold_probe.hip: 4.0 us per 64-wide K tile against 3.1 us for
// three tiles' MFMAs).
//
// Here the tile is shaped to the chip instead: one WAVE owns 32 x 16*NI outputs (2 x NI blocks of v_mfma_f32_16x16x4_f32),
// a workgroup is 2 x 2 waves, and NI is picked so that the grid is a whole number of rounds of 256 workgroups
// (800/512: NI = 5 -> 50 x 5 = 250 workgroups of 64 x 160, ONE wave per SIMD, 10 blocks each -- 97.7 % balance).
// No LDS in the K loop and no barrier anywhere: both operands are k-contiguous in memory (A = rows of W_gifo_r,
// B = rows of W_r_m^T), so lane (i16, kg) loads the 32 bytes at k = 32c + 8kg of ITS row and feeds MFMA steps e = 0..7 with
// the eight components -- the same k bijection for A and B, every k exactly once.  The four k-groups of a row together
// consume one whole 128-byte line per chunk: with 16-byte pieces (chunks of 16 k) every line was fetched by two different
// chunks, the 57 KB in flight per CU do not survive in a 32 KB L1, and the kernel ran at the L2's pace (35 us).
// 2*(2 + NI) loads per 8*2*NI MFMAs, one between every two groups of NI MFMAs; a register ring of FD chunks keeps >= 2
// chunks (5k MFMA cycles) of loads in flight; the two waves that share rows / columns meet in the CU's L1.
// Measured (tools/fold_probe.hip, 800/512): K loop 24.1 us = 42.5 cycles per MFMA against 33.0 with the refills removed
// and 32.0 in a bare MFMA loop (tools/mfma_peak.hip: 140 TFLOP/s at 2.22 GHz on random data) -- every 1 KB load
// instruction costs ~57 cycles of the SIMD's MFMA issue wherever it is placed, wherever its data comes from (re-reading
// chunk 0 out of L1: 42.0), with one or two waves per SIMD, with the K walk rotated per workgroup against L2 channel
// hot spots.  Whole launch 33.8 us against 38.6 us for the tiled kernel.
// Epilogue: rows are read in gates-packed order (logical row 4*cell + gate <- stored row gate*C + cell) and the tile goes
// straight into the two packed operands of the folded chain (through a wave-private LDS transpose, no workgroup barrier):
//   pk1  [W_rm | W_x] gates operand   [C/4 tiles][nch1 chunks of 32][2][64] float4, float4 = 4 consecutive k of one row
//   pk2  W_rm^T, 4-row geometry        [C/4 tiles][nch2 chunks of 128][2][64] float4, float4 = 4 consecutive cells of one gate
// (layouts: klstm_kernels.hip k_gates_v / k_dmf_v; klstm_persist.hip reads the same arrays).
#include "klstm_kernels.h"
#include "klstm_math.h"

#include <hip/hip_ext.h>

namespace klstm {

#pragma clang fp contract(off)

struct FoldArgs {
  int C, R;
  const float *wr;     // W_gifo_r [4C x R], rows in g,i,f,o blocks of C
  const float *wmT;    // W_r_m^T  [C x R]
  float4 *pk1; int nch1;
  float4 *pk2; int nch2;
  int nbn;             // workgroups along N
  int nwg;
#ifdef KLSTM_FOLD_TIMING
  int kscale;          // 1; 0 = every refill re-reads chunk 0 (timing experiment: same instruction stream, no new lines)
  long long *dbg;      // per workgroup: shader clocks / wall ticks of the K loop and of the whole kernel (tools/fold_probe.hip)
#endif
};

constexpr int FD = 4;          // chunks of 32 k in the register ring (R % (32*FD) == 0: the K loop has no conditional loads --
                               // with them hipcc drains the whole ring, s_waitcnt vmcnt(0), at the top of every iteration)

// The K loop shared by the direct kernels: lane (i16, kg) holds the row pointers ap[mi] / bp[ni] (already advanced by 8*kg);
// K = 32 * nchunk, nchunk a multiple of FD.  acc[mi][ni] += A rows x B rows over all of K.
template <int MI, int NI>
__device__ __forceinline__ void direct_kloop(const float *const (&ap)[MI], const float *const (&bp)[NI], int nchunk,
                                             f32x4 (&acc)[MI][NI], int kscale = 1) {
  float4 ra[FD][MI][2], rb[FD][NI][2];
  auto load = [&](int d, int c) {
    const int co = 32 * c;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
      ra[d][mi][0] = *reinterpret_cast<const float4 *>(ap[mi] + co);
      ra[d][mi][1] = *reinterpret_cast<const float4 *>(ap[mi] + co + 4);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
      rb[d][ni][0] = *reinterpret_cast<const float4 *>(bp[ni] + co);
      rb[d][ni][1] = *reinterpret_cast<const float4 *>(bp[ni] + co + 4);
    }
  };
  // One stage = the 8*2*NI MFMAs of the chunk in ring slot d, with the 2*(2+NI) loads that refill slot dl (the chunk
  // consumed by the PREVIOUS stage) spread between them, one load per NI MFMAs.  A wave issues in order and this kernel
  // runs one wave per SIMD: 14 loads in a burst hold the issue port for ~800 cycles per chunk while the MFMA pipe drains
  // (measured: 41.8 cycles per MFMA instead of 32); one load between two groups of MFMAs disappears in their shadow.
  auto stage = [&](int d, int dl, int cl, bool refill) {
    float av[MI][8], bv[NI][8];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        av[mi][4 * h] = ra[d][mi][h].x; av[mi][4 * h + 1] = ra[d][mi][h].y; av[mi][4 * h + 2] = ra[d][mi][h].z; av[mi][4 * h + 3] = ra[d][mi][h].w;
      }
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        bv[ni][4 * h] = rb[d][ni][h].x; bv[ni][4 * h + 1] = rb[d][ni][h].y; bv[ni][4 * h + 2] = rb[d][ni][h].z; bv[ni][4 * h + 3] = rb[d][ni][h].w;
      }
#pragma unroll
    for (int e = 0; e < 8; e++)
#pragma unroll
      for (int mi = 0; mi < MI; mi++) {
#pragma unroll
        for (int ni = 0; ni < NI; ni++) acc[mi][ni] = MFMA16(av[mi][e], bv[ni][e], acc[mi][ni]);
        // the 2 * (MI + NI) refill loads, spread evenly over the 8 * MI groups of NI MFMAs
        constexpr int NLOAD = 2 * (MI + NI), NSLOT = 8 * MI;
        const int slot = e * MI + mi;
#pragma unroll
        for (int l = slot * NLOAD / NSLOT; l < (slot + 1) * NLOAD / NSLOT; l++) {
          if (!refill) break;
          const int row = l >> 1, h = l & 1;
          const int co = 32 * cl * kscale;                 // (kscale 0: every refill re-reads chunk 0, timing experiment)
          __builtin_amdgcn_sched_barrier(0);
          if (row < MI) ra[dl][row][h] = *reinterpret_cast<const float4 *>(ap[row] + co + 4 * h);
          else rb[dl][row - MI][h] = *reinterpret_cast<const float4 *>(bp[row - MI] + co + 4 * h);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };
#pragma unroll
  for (int d = 0; d < FD - 1; d++) load(d, d);
  // chunk c lives in slot c % FD; stage c refills slot (c - 1) % FD with chunk c + FD - 1
  stage(0, FD - 1, FD - 1, true);
  int c0 = 1;
  for (; c0 + 2 * FD - 1 <= nchunk; c0 += FD) {        // stages c0 .. c0+FD-1, all of them with a refill (chunk <= nchunk - 1)
#pragma unroll
    for (int j = 0; j < FD; j++) stage((1 + j) % FD, j % FD, c0 + j + FD - 1, true);
  }
  // the last FD - 1 stages (R % (32*FD) == 0: c0 == nchunk - FD + 1 here): nothing left to request
#pragma unroll
  for (int j = 0; j < FD - 1; j++) stage((1 + j) % FD, 0, 0, false);
}

template <int NI>
__global__ __launch_bounds__(256) void k_fold_direct(FoldArgs a) {
  constexpr int MI = 2, WN = 16 * NI, NWAVE = 4;       // wave tile 32 x WN, 2 x 2 waves
  constexpr int FLD = 16 * MI + 4;                     // LDS row stride of the epilogue transpose (floats): 16-byte aligned rows + pad
  __shared__ __attribute__((aligned(16))) float Cs[NWAVE][WN * FLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int C = a.C, R = a.R, nchunk = R / 32;
  // XCD-aware order (workgroup w lands on XCD w % 8): XCD x gets a contiguous m-major range, i.e. a few row panels of A
  // and all of B in its own L2
  const int cpx = (a.nwg + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * cpx + (int)(blockIdx.x >> 3);
  if (b >= a.nwg) return;
#ifdef KLSTM_FOLD_TIMING
  const long long t_c0 = clock64(), t_w0 = wall_clock64();
#endif
  const int m0 = (b / a.nbn) * 64 + (wave >> 1) * 32, n0 = (b % a.nbn) * (2 * WN) + (wave & 1) * WN;

  const float *ap[MI], *bp[NI];
  bool aok[MI], bok[NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
    const int x = m0 + 16 * mi + i16;                  // logical row 4*cell + gate
    aok[mi] = (x >> 2) < C;
    ap[mi] = a.wr + (size_t)(aok[mi] ? (x & 3) * C + (x >> 2) : 0) * R + 8 * kg;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
    const int n = n0 + 16 * ni + i16;
    bok[ni] = n < C;
    bp[ni] = a.wmT + (size_t)(bok[ni] ? n : 0) * R + 8 * kg;
  }
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++) acc[mi][ni] = (f32x4){0, 0, 0, 0};

#ifdef KLSTM_FOLD_TIMING
  const long long t_c1 = clock64(), t_w1 = wall_clock64();
  direct_kloop<MI, NI>(ap, bp, nchunk, acc, a.kscale != 0);
#else
  direct_kloop<MI, NI>(ap, bp, nchunk, acc);
#endif
#ifdef KLSTM_FOLD_TIMING
  const long long t_c2 = clock64(), t_w2 = wall_clock64();
#endif

  // ---- epilogue: wave-private transpose through LDS.  acc[mi][ni] of lane (i16, kg) = rows 16mi + 4kg + (0..3) at
  // column 16ni + i16; Cs[column][row] makes that one 16-byte store.  Rows / columns past the operand were fed row 0 /
  // column 0 of the inputs and are simply not written out.
  float *cs = Cs[wave];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
      *reinterpret_cast<float4 *>(cs + (16 * ni + i16) * FLD + 16 * mi + 4 * kg) =
          make_float4(acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (same wave reads it back: program order on the LDS queue is enough, the
                                                       //  wait only keeps the compiler from hoisting the reads)
  const int cell0 = m0 >> 2;
  // gates operand: piece = (16-row tile tl, column quad nq, row i): 16 consecutive float4 (256 B) per (tl, nq)
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
#pragma unroll
  for (int it = 0; it < 2 * NI; it++) {
    const int cq = lane & 3, kq = (lane >> 2) & 1, gate = (lane >> 3) & 3, ct = it * 2 + (lane >> 5);
    const int c = n0 + ct * 4 + cq, cell = cell0 + kq * 4;
    const float *cp = cs + (ct * 4 + cq) * FLD + kq * 16 + gate;
    const float4 v = make_float4(cp[0], cp[4], cp[8], cp[12]);
    const int k = gate * C + cell;
    if (c < C && cell < C)
      a.pk2[(((size_t)(c >> 2) * a.nch2 + (k >> 7)) * 2 + ((k >> 6) & 1)) * 64 + ((k & 63) >> 2) * 4 + cq] = v;
  }
#ifdef KLSTM_FOLD_TIMING
  if (tid == 0) {
    long long *q = a.dbg + (size_t)blockIdx.x * 8;
    q[0] = t_c2 - t_c1; q[1] = t_w2 - t_w1; q[2] = clock64() - t_c0; q[3] = wall_clock64() - t_w0; q[4] = t_w0;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// C[M x N] = A[M x K] B[N x K]^T + bias for FEW rows and MANY columns (AffineTransform::PropagateFnc of the output layer:
// 80 x 16624 over K = 512, nnet.proto:4): the same register-direct K loop, one wave = all M rows (MI = M/16 blocks) x 16
// (NI) columns, one-wave workgroups: every wave streams its rows of B exactly once; the M rows of A are re-read by every wave
// (160 KB each, out of L2: that traffic, 83 MB at NI = 2, is what bounds the kernel -- wider waves leave SIMDs idle, narrower
// ones double it.  Sharing A through LDS between the 4 waves of a workgroup, one barrier per 32-k chunk, was measured at
// 48 us with one chunk of prefetch lead: at 40 MFMAs per chunk and wave the loads need >= 4 chunks in flight, which the
// register ring of the direct form has and a 3-slot LDS ring does not).  The 64x64-tile kernel spends
// half of its second row tile on padding at M = 80: 37.9 us against 29.5 us here.
// ---------------------------------------------------------------------------------------------------------------------
struct DirectNtArgs {
  int M, N, K;
  const float *A; int lda;
  const float *B; int ldb;
  float *Cm; int ldc;
  const float *bias;
};

template <int MI, int NI>
__global__ __launch_bounds__(256) void k_direct_nt(DirectNtArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int n0 = ((int)blockIdx.x * (int)(blockDim.x >> 6) + wave) * 16 * NI;
  if (n0 >= a.N) return;
  const float *ap[MI], *bp[NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
    const int m = 16 * mi + i16;
    ap[mi] = a.A + (size_t)(m < a.M ? m : 0) * a.lda + 8 * kg;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
    const int n = n0 + 16 * ni + i16;
    bp[ni] = a.B + (size_t)(n < a.N ? n : 0) * a.ldb + 8 * kg;
  }
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++) acc[mi][ni] = (f32x4){0, 0, 0, 0};
  direct_kloop<MI, NI>(ap, bp, a.K / 32, acc);
  // lane (i16, kg): rows 16mi + 4kg + (0..3) at column n0 + 16ni + i16: 16 lanes = 64 contiguous bytes per row
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
    const int n = n0 + 16 * ni + i16;
    if (n >= a.N) continue;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
      const float e[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = 16 * mi + 4 * kg + r;
        if (m < a.M) a.Cm[(size_t)m * a.ldc + n] = e[r] + bv;
      }
    }
  }
}

#ifdef KLSTM_FOLD_TIMING
static long long *g_fold_dbg = nullptr;
static int g_fold_kscale = 1;
#endif
static int g_fold_direct = 1;
void set_fold_direct(int v) { g_fold_direct = v; }

bool fold_direct_supported(const Dims &d) { return g_fold_direct != 0 && d.C % 4 == 0 && d.R % (32 * FD) == 0; }

hipError_t launch_fold_direct(const Dims &d, const float *wr, const float *wmT, float *pk_fold[2], int nch1, int nch2,
                              hipStream_t st, LaunchProbe pr) {
  FoldArgs a;
  a.C = d.C; a.R = d.R; a.wr = wr; a.wmT = wmT;
  a.pk1 = reinterpret_cast<float4 *>(pk_fold[0]); a.nch1 = nch1;
  a.pk2 = reinterpret_cast<float4 *>(pk_fold[1]); a.nch2 = nch2;
#ifdef KLSTM_FOLD_TIMING
  a.dbg = g_fold_dbg; a.kscale = g_fold_kscale;
#endif
  const int nbm = (4 * d.C + 63) / 64;
  // columns per workgroup: the choice with the fewest rounds of 256 workgroups x blocks per wave
  int best = 0, best_cost = 0;
  for (int ni : {5, 4}) {
    const int nbn = (d.C + 32 * ni - 1) / (32 * ni), cost = ((nbm * nbn + 255) / 256) * ni;
    if (!best || cost < best_cost) { best = ni; best_cost = cost; }
  }
  a.nbn = (d.C + 32 * best - 1) / (32 * best);
  a.nwg = nbm * a.nbn;
  const dim3 grid((a.nwg + 7) / 8 * 8), block(256);
  if (best == 5) {
    if (pr.start) hipExtLaunchKernelGGL(k_fold_direct<5>, grid, block, 0, st, pr.start, pr.stop, 0, a);
    else hipLaunchKernelGGL(k_fold_direct<5>, grid, block, 0, st, a);
  } else {
    if (pr.start) hipExtLaunchKernelGGL(k_fold_direct<4>, grid, block, 0, st, pr.start, pr.stop, 0, a);
    else hipLaunchKernelGGL(k_fold_direct<4>, grid, block, 0, st, a);
  }
  return hipGetLastError();
}

// The same product for a NARROW result (the batched x-projection of a wide-input layer at few frames: 80 x 3200 over K = 512): one
// 16-column block per WORKGROUP, its four waves take a quarter of K each (K % 512 == 0) and wave 0 adds the partial tiles in
// fixed order through LDS -- 200 workgroups x 4 waves instead of 100 one-wave workgroups (16.6 -> ~5 us inside configs[3]).
template <int MI>
__global__ __launch_bounds__(256) void k_direct_nt_ks(DirectNtArgs a) {
  __shared__ __attribute__((aligned(16))) f32x4 part[3][MI][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int n0 = (int)blockIdx.x * 16, kq = a.K / 4;
  const float *ap[MI], *bp[1];
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
    const int m = 16 * mi + i16;
    ap[mi] = a.A + (size_t)(m < a.M ? m : 0) * a.lda + wave * kq + 8 * kg;
  }
  const int nb = n0 + i16;
  bp[0] = a.B + (size_t)(nb < a.N ? nb : 0) * a.ldb + wave * kq + 8 * kg;
  f32x4 acc[MI][1];
#pragma unroll
  for (int mi = 0; mi < MI; mi++) acc[mi][0] = (f32x4){0, 0, 0, 0};
  direct_kloop<MI, 1>(ap, bp, kq / 32, acc);
  if (wave > 0) {
#pragma unroll
    for (int mi = 0; mi < MI; mi++) part[wave - 1][mi][lane] = acc[mi][0];
  }
  __syncthreads();
  if (wave > 0 || nb >= a.N) return;
  const float bv = a.bias ? a.bias[nb] : 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
    const f32x4 v = ((acc[mi][0] + part[0][mi][lane]) + part[1][mi][lane]) + part[2][mi][lane];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int m = 16 * mi + 4 * kg + r;
      if (m < a.M) a.Cm[(size_t)m * a.ldc + nb] = e[r] + bv;
    }
  }
}

static int g_nt_ni = 2, g_nt_waves = 1;   //    // measured at 80 x 16624 x 512 (tools/nt_sweep.py): 29.5 us; 1x1 36.7, 2x2 34.0, 4x1 49.4; tiled kernel 37.9
void set_direct_nt_shape(int ni, int waves) { g_nt_ni = ni == 1 || ni == 2 ? ni : 4; g_nt_waves = waves >= 1 && waves <= 4 ? waves : 2; }

bool direct_nt_supported(int M, int N, int K, const float *A, int lda, const float *B, int ldb) {
  return g_fold_direct != 0 && M >= 1 && M <= 80 && N >= 64 && K % (32 * FD) == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
         (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
}

hipError_t launch_direct_nt(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *Cm, int ldc,
                            const float *bias, hipStream_t st, LaunchProbe pr) {
#define KS_LAUNCH(MI_) do { if (pr.start) hipExtLaunchKernelGGL((k_direct_nt_ks<MI_>), grid, block, 0, st, pr.start, pr.stop, 0, a); \
                            else hipLaunchKernelGGL((k_direct_nt_ks<MI_>), grid, block, 0, st, a); } while (0)
  DirectNtArgs a{M, N, K, A, lda, B, ldb, Cm, ldc, bias};
  if (N <= 8192 && K % (4 * 32 * FD) == 0) {                 // narrow result: K split over the four waves of a workgroup
    const dim3 grid((N + 15) / 16), block(256);
    switch ((M + 15) / 16) {
      case 1: KS_LAUNCH(1); break;
      case 2: KS_LAUNCH(2); break;
      case 3: KS_LAUNCH(3); break;
      case 4: KS_LAUNCH(4); break;
      case 5: KS_LAUNCH(5); break;
      default: return hipErrorInvalidValue;
    }
#undef KS_LAUNCH
    return hipGetLastError();
  }
  // columns per wave (16*NI) and waves per workgroup: experiment knobs (g_nt_ni, g_nt_waves)
  const int ni = g_nt_ni, nw = g_nt_waves;
  const dim3 grid((N + 16 * ni * nw - 1) / (16 * ni * nw)), block(64 * nw);
#define NT_GO(MI_, NI_) do { if (pr.start) hipExtLaunchKernelGGL((k_direct_nt<MI_, NI_>), grid, block, 0, st, pr.start, pr.stop, 0, a); \
                             else hipLaunchKernelGGL((k_direct_nt<MI_, NI_>), grid, block, 0, st, a); } while (0)
#define NT_CASE(MI_)                                                                                   \
  case MI_:                                                                                            \
    if (ni == 1) NT_GO(MI_, 1);                                                                        \
    else if (ni == 2) NT_GO(MI_, 2);                                                                   \
    else NT_GO(MI_, 4);                                                                                \
    break;
  switch ((M + 15) / 16) {
    NT_CASE(1) NT_CASE(2) NT_CASE(3) NT_CASE(4) NT_CASE(5)
    default: return hipErrorInvalidValue;
  }
#undef NT_CASE
#undef NT_GO
  return hipGetLastError();
}

}  // namespace klstm
