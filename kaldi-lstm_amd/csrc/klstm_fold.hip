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
#include <mutex>
#include <set>
#include <type_traits>

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
    if (a.pk2 && c < C && cell < C)
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
  unsigned *redo;         // range guard of the f16 form (k_nt_shared_a16): host-mapped event counter, or null
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

// ---------------------------------------------------------------------------------------------------------------------
// C[M x N] = A[M x K] B[K x N] for FEW rows, a narrow result and a LONG contraction (AffineTransform::BackpropagateFnc of the
// output layer: in_diff[80 x 512] = out_diff[80 x 16624] W[16624 x 512], nnet.proto:4): all the parallelism is along K.
//   workgroup (n-range of 128 columns, K slice of 256 rows of B): its four waves take 64 rows each, both 32-row chunks
//   requested up front, operands straight into MFMA registers, no LDS in the loop:
//     A  lane (row i16, k-group kg) loads the 32 bytes at k = k0 + 8kg of its row (as in direct_kloop)
//     B  is n-contiguous: lane (i16, kg) loads, for MFMA step e, the 16 bytes B[k0 + 8kg + e][n0 + 64jj + 4 i16 .. +3] --
//        4 rows x 256 contiguous bytes per instruction.  Component c of that load is column i16 of the VIRTUAL 16-column
//        block {n0 + 64jj + 4i + c : i = 0..15}: any 16 columns can be an MFMA block, and this choice leaves every lane with
//        four CONSECUTIVE result columns (one 16-byte store per row).
//   The four partial tiles of a workgroup are added through LDS (each wave owns a quarter of the blocks and adds the other
//   three waves' copies in fixed order), the workgroup's partial goes to ws[K slice][M][N]; k_skinny_nn_reduce adds the K
//   slices in order (deterministic, like every other reduction here).
// 80 x 512 x 16624: split-K pair of the tiled kernel 34 us -> this pair ~17 us (1.36 GFLOP = 9.7 us at the fp32 MFMA peak).
// ---------------------------------------------------------------------------------------------------------------------
struct SkinnyNnArgs {
  int M, N, K;
  const float *A; int lda;
  const float *B; int ldb;
  float *ws; int nks;
  float *Cm; int ldc;
  int rem;                // the first `rem` wave slots (K slice * 4 + wave) take a ninth group of 8 rows
  unsigned *redo;         // range guard of the f16 form (k_skinny_nn16): host-mapped event counter, or null
#ifdef KLSTM_SKINNY_TIMING
  long long *dbg;         // per workgroup: shader clocks entry -> loads issued -> MFMAs done -> exit (tools/skinny_probe.hip)
#endif
};
__device__ __forceinline__ float4 keep4(const float4 &v, bool c) {      // zero without a select the compiler could turn into a branch around the load
  const unsigned m = c ? 0xffffffffu : 0u;
  return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                     __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
}
template <int MI>
__global__ __launch_bounds__(256) void k_skinny_nn(SkinnyNnArgs a) {
  constexpr int NB = 8, OWN = MI * NB / 4;           // blocks per wave tile; blocks a wave owns in the reduction (MI * 8 % 4 == 0)
  extern __shared__ __attribute__((aligned(16))) f32x4 part[];   // [owner][source slot 0..2][OWN][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int nr = a.N / 128, n0 = ((int)blockIdx.x % nr) * 128, ks = (int)blockIdx.x / nr;
#ifdef KLSTM_SKINNY_TIMING
  const long long t_c0 = clock64(), t_w0 = wall_clock64();
#endif
  // rows of B: wave slot s = 4 ks + wave takes 64 rows (two chunks of 32) from row 8 (8 s + min(s, rem)), the first `rem` slots 8
  // more -- so that a contraction that is not a multiple of 256 x 64 slices (16624 = 2078 groups of 8 over 256 slots) still
  // fits ONE round of workgroups: with 65 slices of 256 rows the 4 workgroups of the 65th ran alone behind the other 256
  const int slot = ks * 4 + wave, kw = 8 * (8 * slot + min(slot, a.rem));
  const bool extra = slot < a.rem;
  float4 ra[2][MI][2], rb[2][8][2];
  float2 xa[MI];
  float4 xb[2][2];
  {
    const int kx = kw + 64 + 2 * kg;                 // ninth group: k = kx + e, e = 0, 1
    const bool xin = extra && kx + 2 <= a.K;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
      const int m = 16 * mi + i16;
      xa[mi] = *reinterpret_cast<const float2 *>(a.A + (size_t)(m < a.M ? m : 0) * a.lda + (xin ? kx : 0));
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const float *p = a.B + (size_t)(xin ? kx + e : 0) * a.ldb + n0 + 4 * i16;
      xb[e][0] = *reinterpret_cast<const float4 *>(p);
      xb[e][1] = *reinterpret_cast<const float4 *>(p + 64);
    }
  }
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int k0 = kw + 32 * c + 8 * kg;
    const bool kin = k0 + 8 <= a.K;                  // K % 8 == 0
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
      const int m = 16 * mi + i16;
      const float *p = a.A + (size_t)(m < a.M ? m : 0) * a.lda + (kin ? k0 : 0);
      ra[c][mi][0] = *reinterpret_cast<const float4 *>(p);          // (rows past M and k past K are zeroed at use: touching a loaded
      ra[c][mi][1] = *reinterpret_cast<const float4 *>(p + 4);      //  register here would make the wave wait before the next request)
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float *p = a.B + (size_t)(kin ? k0 + e : 0) * a.ldb + n0 + 4 * i16;
      rb[c][e][0] = *reinterpret_cast<const float4 *>(p);          // (rows past K meet zeros of A)
      rb[c][e][1] = *reinterpret_cast<const float4 *>(p + 64);
    }
  }
#ifdef KLSTM_SKINNY_TIMING
  const long long t_c1 = clock64();
#endif
  __builtin_amdgcn_sched_barrier(0);                 // ALL 2 x (2 MI + 16) loads are in flight before the first MFMA (left alone, hipcc sinks
                                                     //  them between the MFMAs two at a time: one memory latency per pair, 39 us)
  f32x4 acc[MI][NB];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) acc[mi][nb] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float av[MI];
      const unsigned kmask = kw + 32 * c + 8 * kg + 8 <= a.K ? 0xffffffffu : 0u;
#pragma unroll
      for (int mi = 0; mi < MI; mi++) {
        const float4 &q = ra[c][mi][e >> 2];
        const float raw = (e & 3) == 0 ? q.x : (e & 3) == 1 ? q.y : (e & 3) == 2 ? q.z : q.w;
        av[mi] = __uint_as_float(__float_as_uint(raw) & (16 * mi + i16 < a.M ? kmask : 0u));
      }
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
        const float bv[4] = {rb[c][e][jj].x, rb[c][e][jj].y, rb[c][e][jj].z, rb[c][e][jj].w};
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
          for (int mi = 0; mi < MI; mi++)
            acc[mi][jj * 4 + cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[cc], acc[mi][jj * 4 + cc], 0, 0, 0);
      }
    }
  if (extra) {
#pragma unroll
    for (int e = 0; e < 2; e++) {
      float av[MI];
#pragma unroll
      for (int mi = 0; mi < MI; mi++) av[mi] = 16 * mi + i16 < a.M ? (e ? xa[mi].y : xa[mi].x) : 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
        const float bv[4] = {xb[e][jj].x, xb[e][jj].y, xb[e][jj].z, xb[e][jj].w};
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
          for (int mi = 0; mi < MI; mi++)
            acc[mi][jj * 4 + cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[cc], acc[mi][jj * 4 + cc], 0, 0, 0);
      }
    }
  }
#ifdef KLSTM_SKINNY_TIMING
  __builtin_amdgcn_sched_barrier(0);
  const long long t_c2 = clock64();
#endif
  // ---- the four waves' tiles -> one: block q = mi * 8 + nb belongs to wave q & 3; source s hands it to owner o in slot s - (s > o)
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const int q = mi * NB + nb, o = q & 3;
      if (o != wave) part[((o * 3 + (wave - (wave > o ? 1 : 0))) * OWN + (q >> 2)) * 64 + lane] = acc[mi][nb];
    }
  __syncthreads();
  float *wp = a.ws + (size_t)ks * a.M * a.N;
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const int q = mi * NB + nb;
      if ((q & 3) != wave) continue;
      f32x4 v[4];                                    // contributions in source order 0, 1, 2, 3 (own in its place)
#pragma unroll
      for (int s = 0; s < 4; s++)
        v[s] = s == wave ? acc[mi][nb] : part[((wave * 3 + (s - (s > wave ? 1 : 0))) * OWN + (q >> 2)) * 64 + lane];
      acc[mi][nb] = ((v[0] + v[1]) + v[2]) + v[3];
    }
  // lane (i16, kg) of block (mi, jj, cc): rows 16mi + 4kg + r at column n0 + 64jj + 4 i16 + cc: the four cc blocks of a lane
  // are one 16-byte piece -- but they belong to four different owner waves (q & 3 = cc): back through LDS, then row-wise stores
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
      if (((mi * NB + nb) & 3) == wave) part[(mi * NB + nb) * 64 + lane] = acc[mi][nb];
  __syncthreads();
  // 256 threads: thread = (block pair (mi, jj) = tid >> 6 ... ): walk (mi, jj) pairs, 64 lanes each
  for (int pr = wave; pr < MI * 2; pr += 4) {
    const int mi = pr >> 1, jj = pr & 1;
    f32x4 c4[4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++) c4[cc] = part[(mi * NB + jj * 4 + cc) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int m = 16 * mi + 4 * kg + r;
      if (m < a.M)
        *reinterpret_cast<float4 *>(wp + (size_t)m * a.N + n0 + 64 * jj + 4 * i16) = make_float4(c4[0][r], c4[1][r], c4[2][r], c4[3][r]);
    }
  }
#ifdef KLSTM_SKINNY_TIMING
  if (lane == 0) {
    long long *q = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
    q[0] = t_c1 - t_c0; q[1] = t_c2 - t_c1; q[2] = clock64() - t_c2; q[3] = wall_clock64() - t_w0; q[4] = t_w0;
  }
#endif
}
// C = sum over K slices of ws, in slice order.  One thread per float4 of C and group of slices; groups combined through LDS.
__global__ __launch_bounds__(256) void k_skinny_nn_reduce(SkinnyNnArgs a) {
  __shared__ float4 red[8][32];
  const int tid = threadIdx.x, el = tid & 31, grp = tid >> 5;        // 32 float4 of C per workgroup, 8 slice groups
  const size_t e4 = (size_t)blockIdx.x * 32 + el, tot4 = (size_t)a.M * a.N / 4;
  const int per = (a.nks + 7) / 8, s0 = grp * per, s1 = min(a.nks, s0 + per);
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e4 < tot4)
    for (int sb = s0; sb < s1; sb += 8) {              // eight slices' loads in flight, added in slice order
      float4 v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const float4 *>(a.ws + ((size_t)min(sb + q, s1 - 1) * tot4 + e4) * 4);
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (sb + q < s1) { sum.x += v[q].x; sum.y += v[q].y; sum.z += v[q].z; sum.w += v[q].w; }
    }
  red[grp][el] = sum;
  __syncthreads();
  if (grp == 0 && e4 < tot4) {
    float4 t = red[0][el];
#pragma unroll
    for (int g = 1; g < 8; g++) { t.x += red[g][el].x; t.y += red[g][el].y; t.z += red[g][el].z; t.w += red[g][el].w; }
    const size_t m = e4 * 4 / a.N, n = e4 * 4 % a.N;
    *reinterpret_cast<float4 *>(a.Cm + m * a.ldc + n) = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same product on the f16 matrix cores at fp32 accuracy (klstm_math.h f16_split2_pair, three products, cross terms in their own
// accumulators -- klstm_fold3.hip / klstm_outer.hip), both operands split in registers: A rows are k-contiguous (8 consecutive k
// of a lane's row = two 16-byte loads = one operand), B is the virtual-column layout above (component cn of the eight rows a
// lane loads = the operand of column block cn).  Workgroup (64-column tile t, K group g): ntile x G = one round of the chip;
// the 32-row chunks of K are dealt evenly to the 4 G wave slots (16624 = 519.5 chunks over 128 slots: 4 or 5 each), the raw rows
// of the next chunk in flight under the 12 MI MFMAs of the current one; the four waves' tiles meet in LDS in wave order and
// go to ws[g][M][N]; k_skinny_nn_reduce adds the G groups in order.  80 x 512 x 16624: 30.6 us (k_skinny_nn: 7 us until the
// loads are issued, then 10 us of fp32 MFMAs, then the reduction, nothing overlapped) -> see tools/t_indiff.py.
// ---------------------------------------------------------------------------------------------------------------------
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 sk_f16x8 __attribute__((ext_vector_type(8)));
// (two products in one launch: blocks [0, nb1) work on a, the rest on b -- d_r and in_diff of the folded BPTT tail, launch_bwd_tail)
template <int MI>
__global__ __launch_bounds__(256) void k_skinny_nn16(SkinnyNnArgs a_, SkinnyNnArgs b_, int nb1) {
  extern __shared__ __attribute__((aligned(16))) float part16[];    // [4 waves][16 MI rows][64]
  const bool second = (int)blockIdx.x >= nb1;
  const SkinnyNnArgs &a = second ? b_ : a_;
  const int bid = second ? (int)blockIdx.x - nb1 : (int)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int ntile = (a.N + 63) / 64, t = bid % ntile, g = bid / ntile;
  const int nc = 64 * t + 4 * i16;
  const bool n_in = nc < a.N;
  const int nchunk = (a.K + 31) / 32, nslot = 4 * a.nks, slot = 4 * g + wave;
  const int c0 = (int)((long)slot * nchunk / nslot), c1 = (int)((long)(slot + 1) * nchunk / nslot);
  const float *ap[MI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++) ap[mi] = a.A + (size_t)min(16 * mi + i16, a.M - 1) * a.lda;
  const float *bp = a.B + (n_in ? nc : 0);
  sk_f32x4 ra[3][MI][2], rb[3][8];
  auto loadA = [&](sk_f32x4 (&da)[MI][2], int c_) {
    const int c = c_ < c1 ? c_ : c1 - 1;                 // (unconditional: past the slot's last chunk the same rows again, never used)
    const int k = 32 * c + 8 * kg, ka = k + 8 <= a.K ? k : 0;        // K % 8 == 0
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
      da[mi][0] = *reinterpret_cast<const sk_f32x4 *>(ap[mi] + ka);
      da[mi][1] = *reinterpret_cast<const sk_f32x4 *>(ap[mi] + ka + 4);
    }
  };
  auto loadB = [&](sk_f32x4 (&db)[8], int c_) {
    const int c = c_ < c1 ? c_ : c1 - 1;
    const int k = 32 * c + 8 * kg;
#pragma unroll
    for (int e = 0; e < 8; e++) db[e] = *reinterpret_cast<const sk_f32x4 *>(bp + (size_t)(k + e < a.K ? k + e : 0) * a.ldb);
  };
  sk_f32x4 acc[MI][4], accx[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int cn = 0; cn < 4; cn++) { acc[mi][cn] = (sk_f32x4){0, 0, 0, 0}; accx[mi][cn] = (sk_f32x4){0, 0, 0, 0}; }
  // one chunk: B planes first and the rows of B three chunks on requested into the registers just freed (W streams from HBM), then row
  // block by row block: split A's eight values, 12 MFMAs; A's rows three chunks on requested at the end
  auto body = [&](int c, sk_f32x4 (&ca)[MI][2], sk_f32x4 (&cb)[8]) {
    const bool kin = 32 * c + 8 * kg + 8 <= a.K;
    sk_f16x8 b1[4], b2[4];
#pragma unroll
    for (int cn = 0; cn < 4; cn++) {
      uint4 u1, u2;
      f16_split2_pair(cb[0][cn], cb[1][cn], u1.x, u2.x);
      f16_split2_pair(cb[2][cn], cb[3][cn], u1.y, u2.y);
      f16_split2_pair(cb[4][cn], cb[5][cn], u1.z, u2.z);
      f16_split2_pair(cb[6][cn], cb[7][cn], u1.w, u2.w);
      b1[cn] = __builtin_bit_cast(sk_f16x8, u1); b2[cn] = __builtin_bit_cast(sk_f16x8, u2);   // (rows past K meet zeros of A)
    }
    loadB(cb, c + 3);
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
      const bool on = kin && 16 * mi + i16 < a.M;
      const sk_f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const sk_f32x4 v0 = (on ? ca[mi][0] : z) * DERIV_SCALE, v1 = (on ? ca[mi][1] : z) * DERIV_SCALE;   // (A is a derivative: klstm_math.h)
      uint4 u1, u2;
      f16_split2_pair(v0[0], v0[1], u1.x, u2.x);
      f16_split2_pair(v0[2], v0[3], u1.y, u2.y);
      f16_split2_pair(v1[0], v1[1], u1.z, u2.z);
      f16_split2_pair(v1[2], v1[3], u1.w, u2.w);
      const sk_f16x8 a1 = __builtin_bit_cast(sk_f16x8, u1), a2 = __builtin_bit_cast(sk_f16x8, u2);
#pragma unroll
      for (int cn = 0; cn < 4; cn++) {
        accx[mi][cn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2[cn], accx[mi][cn], 0, 0, 0);
        accx[mi][cn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1[cn], accx[mi][cn], 0, 0, 0);
        acc[mi][cn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1[cn], acc[mi][cn], 0, 0, 0);
      }
    }
    loadA(ca, c + 3);
    __builtin_amdgcn_sched_barrier(0);
  };
  if (c0 < c1) {
    loadB(rb[0], c0); loadA(ra[0], c0);
    loadB(rb[1], c0 + 1); loadA(ra[1], c0 + 1);
    loadB(rb[2], c0 + 2); loadA(ra[2], c0 + 2);
    for (int c = c0; c < c1; c += 3) {                   // three chunks in flight: 24 KB per wave
      body(c, ra[0], rb[0]);
      if (c + 1 < c1) body(c + 1, ra[1], rb[1]);
      if (c + 2 < c1) body(c + 2, ra[2], rb[2]);
    }
  }
  // ---- the four waves' tiles -> one, in wave order; accumulator (mi, cn)[r] = C[16 mi + 4 kg + r][nc + cn] over this wave's
  //      chunks: the two accumulator sets into one, the derivative scale out again ----
  float *mine = part16 + (size_t)wave * (16 * MI) * 64;
  float probe = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      sk_f32x4 v;
#pragma unroll
      for (int cn = 0; cn < 4; cn++) {
        v[cn] = (acc[mi][cn][r] + accx[mi][cn][r] * (1.f / 2048.f)) * DERIV_UNSCALE;
        probe = nonfinite_probe(probe, v[cn]);
      }
      *reinterpret_cast<sk_f32x4 *>(mine + (16 * mi + 4 * kg + r) * 64 + 4 * i16) = v;
    }
  // range guard (klstm_math.h): an operand beyond the fp16 range -> the wave's partial tile again in plain fp32, written over its
  // LDS copy (same wave: the LDS queue keeps the order)
  if (wave_any(probe != probe)) {
    redo_note(a.redo);
    const int k0 = 32 * c0, k1 = min(32 * c1, a.K);
#pragma unroll 1
    for (int q = 0; q < MI * 16; q++) {
      const int mi = q >> 4, cn = (q >> 2) & 3, r = q & 3, m = 16 * mi + 4 * kg + r;
      mine[m * 64 + 4 * i16 + cn] =
          (m < a.M && n_in && k1 > k0) ? redo_dot(a.A + (size_t)m * a.lda + k0, 1, a.B + (size_t)k0 * a.ldb + nc + cn, a.ldb, k1 - k0) : 0.f;
    }
  }
  __syncthreads();
  float *wp = a.ws + (size_t)g * a.M * a.N;
  for (int q = tid; q < a.M * 16; q += 256) {            // 16 pieces of 16 bytes per row
    const int row = q >> 4, col = 4 * (q & 15);
    if (64 * t + col >= a.N) continue;
    const float *p = part16 + row * 64 + col;
    const sk_f32x4 v = ((*reinterpret_cast<const sk_f32x4 *>(p) + *reinterpret_cast<const sk_f32x4 *>(p + 16 * MI * 64)) +
                        *reinterpret_cast<const sk_f32x4 *>(p + 2 * 16 * MI * 64)) + *reinterpret_cast<const sk_f32x4 *>(p + 3 * 16 * MI * 64);
    *reinterpret_cast<sk_f32x4 *>(wp + (size_t)row * a.N + 64 * t + col) = v;
  }
}
static int g_skinny16 = 1;
void set_skinny_f16(int on) { g_skinny16 = on; }
static int g_skinny16_pair = 1;
void set_skinny_f16_pair(int on) { g_skinny16_pair = on; }
static int skinny16_groups(int N, int K) {
  const int ntile = (N + 63) / 64, nchunk = (K + 31) / 32;
  int G = 256 / ntile;
  if (G > nchunk / 4) G = nchunk / 4;
  return G < 1 ? 1 : G;
}

#ifdef KLSTM_SKINNY_TIMING
static long long *g_skinny_dbg = nullptr;
#endif
// K slices: groups of 8 rows over wave slots of 8 groups (+ 1 for the first `rem` slots), at most 256 workgroups
static bool skinny_nn_plan(int N, int K, int *nks, int *rem) {
  const int noct = K / 8, nr = N / 128, max_slots = 4 * (256 / nr);
  int slots = (noct + 7) / 8;
  *rem = 0;
  if (slots > max_slots) { slots = max_slots; *rem = noct - 8 * slots; }
  *nks = (slots + 3) / 4;
  return *rem <= slots;
}
bool skinny_nn_supported(int M, int N, int K, const float *A, int lda, const float *B, int ldb, const float *Cm, int ldc) {
  int nks, rem;
  if (g_skinny16 && redo_count(REDO_SKINNY) == 0 && g_fold_direct != 0 && M >= 1 && M <= 80 && N % 4 == 0 && N >= 64 && N <= 1024 && K >= 4096 && K % 8 == 0 && lda % 4 == 0 &&
      ldb % 4 == 0 && ldc % 4 == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(Cm)) & 15) == 0)
    return true;
  return g_fold_direct != 0 && M >= 1 && M <= 80 && N >= 128 && N % 128 == 0 && N <= 1024 && K >= 4096 && K % 8 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
         ldc % 4 == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(Cm)) & 15) == 0 &&
         skinny_nn_plan(N, K, &nks, &rem);
}
size_t skinny_nn_workspace_floats(int M, int N, int K) {
  int nks, rem;
  if (N % 128 != 0 || N < 128 || !skinny_nn_plan(N, K, &nks, &rem)) nks = 0;
  const int g16 = skinny16_groups(N, K);
  return (size_t)(nks > g16 ? nks : g16) * M * N;
}
hipError_t launch_skinny_nn(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *Cm, int ldc, float *ws,
                            hipStream_t st) {
  int nks, rem;
  if (g_skinny16 && redo_count(REDO_SKINNY) == 0) {
    SkinnyNnArgs a{};
    a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.ws = ws; a.nks = skinny16_groups(N, K); a.Cm = Cm; a.ldc = ldc;
    a.redo = redo_counters() ? redo_counters() + REDO_SKINNY : nullptr;
    const int mi = (M + 15) / 16;
    const dim3 grid(((N + 63) / 64) * a.nks), block(256);
    const size_t shm = (size_t)4 * 16 * mi * 64 * sizeof(float);
#define SK16_GO(MI_) do { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_skinny_nn16<MI_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
                          hipLaunchKernelGGL((k_skinny_nn16<MI_>), grid, block, shm, st, a, a, (int)grid.x); } while (0)
    switch (mi) {
      case 1: SK16_GO(1); break;
      case 2: SK16_GO(2); break;
      case 3: SK16_GO(3); break;
      case 4: SK16_GO(4); break;
      case 5: SK16_GO(5); break;
      default: return hipErrorInvalidValue;
    }
#undef SK16_GO
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_skinny_nn_reduce, dim3((unsigned)(((size_t)M * N / 4 + 31) / 32)), dim3(256), 0, st, a);
    return hipGetLastError();
  }
  if (N % 128 != 0 || N < 128 || !skinny_nn_plan(N, K, &nks, &rem)) return hipErrorInvalidValue;
#ifdef KLSTM_SKINNY_TIMING
  SkinnyNnArgs a{M, N, K, A, lda, B, ldb, ws, nks, Cm, ldc, rem, nullptr, g_skinny_dbg};
#else
  SkinnyNnArgs a{M, N, K, A, lda, B, ldb, ws, nks, Cm, ldc, rem, nullptr};
#endif
  const int mi = (M + 15) / 16;
  const dim3 grid((N / 128) * a.nks), block(256);
  const size_t shm = (size_t)4 * 3 * (mi * 8 / 4) * 64 * sizeof(f32x4) > (size_t)mi * 8 * 64 * sizeof(f32x4)
                         ? (size_t)4 * 3 * (mi * 8 / 4) * 64 * sizeof(f32x4) : (size_t)mi * 8 * 64 * sizeof(f32x4);
#define SK_GO(MI_) do { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_skinny_nn<MI_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
                        hipLaunchKernelGGL((k_skinny_nn<MI_>), grid, block, shm, st, a); } while (0)
  switch (mi) {
    case 1: SK_GO(1); break;
    case 2: SK_GO(2); break;
    case 3: SK_GO(3); break;
    case 4: SK_GO(4); break;
    case 5: SK_GO(5); break;
    default: return hipErrorInvalidValue;
  }
#undef SK_GO
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(k_skinny_nn_reduce, dim3((unsigned)(((size_t)M * N / 4 + 31) / 32)), dim3(256), 0, st, a);
  return hipGetLastError();
}

// Two products over the same rows of A in one launch (the tail of the folded BPTT: d_r = DGIFO[2..] W_gifo_r and in_diff =
// DGIFO[1..] W_gifo_x, 80 x 512 over K = 3200 each): partials to ws1 / ws2 as [G][M][N]; the caller's reduction adds the G slices.
// Returns G (0: not applicable -- the caller keeps its tiled split-K pair).
int skinny16_pair_groups(int M, int N1, int N2, int K, int max_groups) {
  if (!g_skinny16 || !g_skinny16_pair || redo_count(REDO_SKINNY) != 0 || g_fold_direct == 0 || M < 1 || M > 80 || K < 1024 || K % 8 != 0 || N1 % 4 != 0 || N2 % 4 != 0 || N1 < 64 ||
      (N2 != 0 && N2 < 64))
    return 0;
  const int nt = (N1 + 63) / 64 + (N2 + 63) / 64, nchunk = (K + 31) / 32;
  if (nt > 64) return 0;
  int G = 256 / nt;
  if (G > nchunk / 4) G = nchunk / 4;
  if (G > max_groups) G = max_groups;
  return G < 1 ? 0 : G;
}
hipError_t launch_skinny16_pair(int M, int K, const float *A1, const float *A2, int lda, const float *B1, int N1, const float *B2, int N2,
                                float *ws1, float *ws2, int G, hipStream_t st, LaunchProbe pr) {
  SkinnyNnArgs a{}, b{};
  a.M = M; a.N = N1; a.K = K; a.A = A1; a.lda = lda; a.B = B1; a.ldb = N1; a.ws = ws1; a.nks = G;
  a.redo = redo_counters() ? redo_counters() + REDO_SKINNY : nullptr;
  b = a; b.N = N2 > 0 ? N2 : N1; b.A = A2; b.B = B2; b.ldb = b.N; b.ws = ws2;
  const int nb1 = ((N1 + 63) / 64) * G, nb2 = N2 > 0 ? ((N2 + 63) / 64) * G : 0;
  const int mi = (M + 15) / 16;
  const dim3 grid(nb1 + nb2), block(256);
  const size_t shm = (size_t)4 * 16 * mi * 64 * sizeof(float);
#define SK16P_GO(MI_) do { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_skinny_nn16<MI_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
                           if (pr.start) hipExtLaunchKernelGGL((k_skinny_nn16<MI_>), grid, block, shm, st, pr.start, pr.stop, 0, a, b, nb1); \
                           else hipLaunchKernelGGL((k_skinny_nn16<MI_>), grid, block, shm, st, a, b, nb1); } while (0)
  switch (mi) {
    case 1: SK16P_GO(1); break;
    case 2: SK16P_GO(2); break;
    case 3: SK16P_GO(3); break;
    case 4: SK16P_GO(4); break;
    case 5: SK16P_GO(5); break;
    default: return hipErrorInvalidValue;
  }
#undef SK16P_GO
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// The same NT product with the M rows of A SHARED by the four waves of a workgroup through LDS (AffineTransform::PropagateFnc of
// the output layer, 80 x 16624 over K = 512): k_direct_nt makes every wave fetch all of A itself, so either half the SIMDs idle
// (two 16-column blocks per wave: 520 waves, 41 k MFMA clocks each) or every CU pulls 650 KB (one block per wave) -- and a CU
// ingests 30-55 GB/s.  Here a workgroup = 4 waves x one 16-column block; per 32-k chunk the 256 threads
// stage the 10 KB of A once (global -> registers two chunks ahead -> LDS, rows padded to 36 floats: the operand reads of 16
// rows x 4 k-groups are conflict-free), ONE barrier per chunk, two LDS buffers; B stays register-direct with a ring of 4 chunks.
// Per CU: 160 KB of A + 128 KB of B instead of 448-650 KB.  (Round 2 tried this with one chunk of lead and measured 48 us: the
// loads need the depth they have here.)
// ---------------------------------------------------------------------------------------------------------------------
template <int MI>
__global__ __launch_bounds__(256) void k_nt_shared_a(DirectNtArgs a) {
  constexpr int LDA = 36, ROWS = 16 * MI, UNITS = ROWS * 8, NH = (UNITS + 255) / 256;
  __shared__ __attribute__((aligned(16))) float As[4][ROWS * LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int nchunk = a.K / 32;
  // A staging units of this thread: (row, 16-byte piece of the 128-byte chunk row)
  const float *gsrc[NH];
  int ldst[NH];
  bool gon[NH];
#pragma unroll
  for (int h = 0; h < NH; h++) {
    const int u = tid + 256 * h, row = u >> 3, pc = u & 7;
    gon[h] = u < UNITS && row < a.M;
    gsrc[h] = a.A + (size_t)(gon[h] ? row : 0) * a.lda + 4 * pc;
    ldst[h] = u < UNITS ? row * LDA + 4 * pc : -1;
  }
  // One pass over K for this wave's column block nbk (16 columns from 16 * nbk) and the row blocks [m_lo, m_lo + NM) of A.
  // All four waves run the same barrier sequence; a wave without work (nbk < 0) stages A and multiplies nothing.
  auto kpass = [&](int nbk, int m_lo, auto NMC) {
    constexpr int NM = decltype(NMC)::value;
    const int nb = 16 * (nbk < 0 ? 0 : nbk) + i16;
    const float *bp = a.B + (size_t)(nb < a.N ? nb : 0) * a.ldb + 8 * kg;
    float4 ga[4][NH], rb[8][2];                       // A: 3 chunks of lead (+ the one in LDS), B: 7
    auto loadA = [&](int slot, int c) {
#pragma unroll
      for (int h = 0; h < NH; h++) ga[slot][h] = *reinterpret_cast<const float4 *>(gsrc[h] + 32 * c);
    };
    auto stashA = [&](int slot, int buf) {
#pragma unroll
      for (int h = 0; h < NH; h++)
        if (ldst[h] >= 0) *reinterpret_cast<float4 *>(&As[buf][ldst[h]]) = keep4(ga[slot][h], gon[h]);
    };
    auto loadB = [&](int slot, int c) {
      rb[slot][0] = *reinterpret_cast<const float4 *>(bp + 32 * c);
      rb[slot][1] = *reinterpret_cast<const float4 *>(bp + 32 * c + 4);
    };
    f32x4 acc[NM];
#pragma unroll
    for (int mi = 0; mi < NM; mi++) acc[mi] = (f32x4){0, 0, 0, 0};
    // Pipeline (chunk = 32 k): global request of A four chunks ahead and of B seven; A chunk c + 2 goes into LDS buffer (c + 2) & 3
    // during step c, the operand registers of chunk c + 1 are read from LDS during step c (before the MFMAs of chunk c, which
    // run on the other register set), ONE barrier per step.  (Operands read at the top of their own step: 20 us at 141
    // workgroups -- an LDS round trip and a barrier in front of every 40 MFMAs.)
#pragma unroll
    for (int d = 0; d < 4; d++) loadA(d, d < nchunk ? d : nchunk - 1);
#pragma unroll
    for (int d = 0; d < 7; d++) loadB(d, d < nchunk ? d : nchunk - 1);
    stashA(0, 0);
    stashA(1, 1);
    __syncthreads();
    float4 av0[NM][2], av1[NM][2];
    auto readA = [&](float4 (&av)[NM][2], int buf) {
      const float *ab = As[buf] + (m_lo * 16 + i16) * LDA + 8 * kg;
#pragma unroll
      for (int mi = 0; mi < NM; mi++) {
        av[mi][0] = *reinterpret_cast<const float4 *>(ab + mi * 16 * LDA);
        av[mi][1] = *reinterpret_cast<const float4 *>(ab + mi * 16 * LDA + 4);
      }
    };
    readA(av0, 0);
    auto step = [&](int c, auto JC, float4 (&cur)[NM][2], float4 (&nxt)[NM][2]) {   // J = c & 7 at compile time: register ring slots must
      constexpr int J = decltype(JC)::value;                                        // be static (runtime slots: rings in scratch, 61 us)
      // (unconditional requests, the chunk index clamped: a request under a condition makes hipcc drain the whole queue at its
      //  next use; the last ones re-read the last chunk and are never used)
      loadA(J & 3, c + 4 < nchunk ? c + 4 : nchunk - 1);
      loadB((J + 7) & 7, c + 7 < nchunk ? c + 7 : nchunk - 1);
      readA(nxt, (J + 1) & 3);
      const float bv[8] = {rb[J][0].x, rb[J][0].y, rb[J][0].z, rb[J][0].w, rb[J][1].x, rb[J][1].y, rb[J][1].z, rb[J][1].w};
      if (nbk >= 0) {
#pragma unroll
        for (int e = 0; e < 8; e++)
#pragma unroll
          for (int mi = 0; mi < NM; mi++) {
            const float4 &q = cur[mi][e >> 2];
            acc[mi] = MFMA16((e & 3) == 0 ? q.x : (e & 3) == 1 ? q.y : (e & 3) == 2 ? q.z : q.w, bv[e], acc[mi]);
          }
      }
      stashA((J + 2) & 3, (J + 2) & 3);                // chunk c + 2 (requested two steps ago); its buffer held chunk c - 2
      __syncthreads();
    };
    for (int c0 = 0; c0 < nchunk; c0 += 8) {           // K % 256 == 0
      step(c0, std::integral_constant<int, 0>(), av0, av1);
      step(c0 + 1, std::integral_constant<int, 1>(), av1, av0);
      step(c0 + 2, std::integral_constant<int, 2>(), av0, av1);
      step(c0 + 3, std::integral_constant<int, 3>(), av1, av0);
      step(c0 + 4, std::integral_constant<int, 4>(), av0, av1);
      step(c0 + 5, std::integral_constant<int, 5>(), av1, av0);
      step(c0 + 6, std::integral_constant<int, 6>(), av0, av1);
      step(c0 + 7, std::integral_constant<int, 7>(), av1, av0);
    }
    if (NM == MI) {
      // main pass: the workgroup's 16 MI x 64 result tile goes through LDS (the staging buffers are free behind the last barrier)
      // and leaves as 16-byte pieces, 256 contiguous bytes per row -- lane-per-element stores write 64-byte half lines
      float *cs = &As[0][0];                           // [ROWS][68]
      const float bias = (nbk >= 0 && nb < a.N && a.bias) ? a.bias[nb] : 0.f;
#pragma unroll
      for (int mi = 0; mi < NM; mi++) {
        const float e[4] = {acc[mi].x, acc[mi].y, acc[mi].z, acc[mi].w};
#pragma unroll
        for (int r = 0; r < 4; r++) cs[(16 * mi + 4 * kg + r) * 68 + wave * 16 + i16] = e[r] + bias;
      }
      __syncthreads();
      const int c0 = (int)blockIdx.x * 64;
      for (int u = tid; u < ROWS * 16; u += 256) {
        const int m = u >> 4, cq = (u & 15) * 4, n = c0 + cq;
        if (m >= a.M || n >= a.N) continue;
        const float4 v = *reinterpret_cast<const float4 *>(cs + m * 68 + cq);
        float *dp = a.Cm + (size_t)m * a.ldc + n;
        if (n + 4 <= a.N && (a.ldc & 3) == 0) *reinterpret_cast<float4 *>(dp) = v;
        else { const float ev[4] = {v.x, v.y, v.z, v.w}; for (int q = 0; q < 4 && n + q < a.N; q++) dp[q] = ev[q]; }
      }
      __syncthreads();                                 // (the extra pass stages into the same buffers)
      return;
    }
    if (nbk < 0 || nb >= a.N) return;
    const float bias = a.bias ? a.bias[nb] : 0.f;
#pragma unroll
    for (int mi = 0; mi < NM; mi++) {
      const float e[4] = {acc[mi].x, acc[mi].y, acc[mi].z, acc[mi].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = 16 * (m_lo + mi) + 4 * kg + r;
        if (m < a.M) a.Cm[(size_t)m * a.ldc + nb] = e[r] + bias;
      }
    }
  };
  // main pass: column block 4 * workgroup + wave, all row blocks.  The 16-column blocks beyond 4 x (number of workgroups)
  // -- 16624 columns = 1039 blocks against 1024 SIMDs: a 1025th..1039th wave would double the kernel -- are cut into
  // (block, row block) units and handed out as ONE extra unit to the first waves: 6/5 of the time instead of 2x.
  const int slot = (int)blockIdx.x * 4 + wave, nb_main = (int)gridDim.x * 4, nb_total = (a.N + 15) / 16;
  kpass(slot < nb_total ? slot : -1, 0, std::integral_constant<int, MI>());
  const int xunits = (nb_total - nb_main) * MI;
  if ((int)blockIdx.x * 4 >= xunits) return;           // (workgroup-uniform)
  kpass(slot < xunits ? nb_main + slot / MI : -1, slot < xunits ? slot % MI : 0, std::integral_constant<int, 1>());
}

// ---------------------------------------------------------------------------------------------------------------------
// The same product on the f16 matrix cores at fp32 accuracy, operands split ON THE FLY (klstm_math.h f16_split2: x = h1 + h2 / 2048,
// three products, cross terms in their own accumulators -- the scheme of the fold product, klstm_fold3.hip): the rows of A are
// split once per workgroup when they are staged into LDS (two fp16 planes, rows of 64 bytes, the 16-byte k-group of row r in slot
// kg ^ ((-(r >> 2)) & 3): conflict-free operand reads), the rows of B in registers by the wave that streams them (each element of
// B is used by exactly one wave).  15 MFMAs of 16x16x32 per 32-k chunk and wave instead of 40 of 16x16x4.
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 nt_f16x8 __attribute__((ext_vector_type(8)));
template <int MI>
__global__ __launch_bounds__(256, 2) void k_nt_shared_a16(DirectNtArgs a) {
  constexpr bool DB = true;                              // operands of A read from LDS one step ahead
  constexpr int BD = MI >= 5 ? 6 : 7, AD = MI >= 5 ? 3 : 4;   // A: loaded AD chunks ahead, in LDS two chunks ahead                    // chunks of B in flight per wave (registers: two waves per SIMD must fit)
  constexpr int ROWS = 16 * MI, UNITS = ROWS * 8, NH = (UNITS + 255) / 256, PLB = ROWS * 64;   // bytes of one plane of one buffer
  __shared__ __attribute__((aligned(16))) unsigned char As[4][2][PLB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int nchunk = a.K / 32;
  const float *gsrc[NH];
  int ldst[NH];
  bool gon[NH];
#pragma unroll
  for (int h = 0; h < NH; h++) {
    const int u = tid + 256 * h, row = u >> 3, pc = u & 7;             // piece pc = floats 4 pc .. 4 pc + 3 of the chunk row
    gon[h] = u < UNITS && row < a.M;
    gsrc[h] = a.A + (size_t)(gon[h] ? row : 0) * a.lda + 4 * pc;
    ldst[h] = u < UNITS ? row * 64 + (((pc >> 1) ^ ((-(row >> 2)) & 3)) * 16) + (pc & 1) * 8 : -1;
  }
  // (Measured and dropped, round 6: every workgroup starting at another chunk of K and wrapping around -- all workgroups read the same rows
  //  of A, whose 2 KB stride puts a chunk's row pieces into lines that differ in address bits 11 and up only; the suspicion was one L2
  //  channel per chunk.  80 x 16624 over 512: 20.3 us with and without; 8 chunks 14.3, 16 chunks 20.3: 0.75 us per chunk + 8 us per launch.)
  auto kpass = [&](int nbk, int m_lo, auto NMC) {
    constexpr int NM = decltype(NMC)::value;
    const int nb = 16 * (nbk < 0 ? 0 : nbk) + i16;
    const float *bp = a.B + (size_t)(nb < a.N ? nb : 0) * a.ldb + 8 * kg;
    float4 ga[4][NH], rb[8][2];
    auto loadA = [&](int slot, int c) {
#pragma unroll
      for (int h = 0; h < NH; h++) ga[slot][h] = *reinterpret_cast<const float4 *>(gsrc[h] + 32 * c);
    };
    auto stashA = [&](int slot, int buf) {
#pragma unroll
      for (int h = 0; h < NH; h++)
        if (ldst[h] >= 0) {
          const float4 v = keep4(ga[slot][h], gon[h]);
          uint2 h1, h2;
          f16_split2_pair(v.x, v.y, h1.x, h2.x);
          f16_split2_pair(v.z, v.w, h1.y, h2.y);
          *reinterpret_cast<uint2 *>(&As[buf][0][ldst[h]]) = h1;
          *reinterpret_cast<uint2 *>(&As[buf][1][ldst[h]]) = h2;
        }
    };
    auto loadB = [&](int slot, int c) {
      rb[slot][0] = *reinterpret_cast<const float4 *>(bp + 32 * c);
      rb[slot][1] = *reinterpret_cast<const float4 *>(bp + 32 * c + 4);
    };
    f32x4 acc[NM], accx[NM];
#pragma unroll
    for (int mi = 0; mi < NM; mi++) { acc[mi] = (f32x4){0, 0, 0, 0}; accx[mi] = (f32x4){0, 0, 0, 0}; }
#pragma unroll
    for (int d = 0; d < AD; d++) loadA(d, d < nchunk ? d : nchunk - 1);
#pragma unroll
    for (int d = 0; d < BD; d++) loadB(d, d < nchunk ? d : nchunk - 1);
    stashA(0, 0);
    stashA(1, 1);
    __syncthreads();
    const int aoff = (m_lo * 16 + i16) * 64 + ((kg ^ ((-(i16 >> 2)) & 3)) * 16);
    nt_f16x8 av0[NM][2], av1[NM][2];
    auto readA = [&](nt_f16x8 (&av)[NM][2], int buf) {
#pragma unroll
      for (int mi = 0; mi < NM; mi++) {
        av[mi][0] = *reinterpret_cast<const nt_f16x8 *>(&As[buf][0][aoff + mi * 1024]);
        av[mi][1] = *reinterpret_cast<const nt_f16x8 *>(&As[buf][1][aoff + mi * 1024]);
      }
    };
    if (DB) readA(av0, 0);
    auto step = [&](int c, auto JC, nt_f16x8 (&cur)[NM][2], nt_f16x8 (&nxt)[NM][2]) {
      constexpr int J = decltype(JC)::value;
      loadA((J + AD) & 3, c + AD < nchunk ? c + AD : nchunk - 1);
      loadB((J + BD) & 7, c + BD < nchunk ? c + BD : nchunk - 1);
      if (DB) readA(nxt, (J + 1) & 3);
      else readA(cur, J & 3);
      const float bf[8] = {rb[J][0].x, rb[J][0].y, rb[J][0].z, rb[J][0].w, rb[J][1].x, rb[J][1].y, rb[J][1].z, rb[J][1].w};
      uint4 u1, u2;
      f16_split2_pair(bf[0], bf[1], u1.x, u2.x);
      f16_split2_pair(bf[2], bf[3], u1.y, u2.y);
      f16_split2_pair(bf[4], bf[5], u1.z, u2.z);
      f16_split2_pair(bf[6], bf[7], u1.w, u2.w);
      const nt_f16x8 b1 = __builtin_bit_cast(nt_f16x8, u1), b2 = __builtin_bit_cast(nt_f16x8, u2);
      if (nbk >= 0) {
#pragma unroll
        for (int mi = 0; mi < NM; mi++) {
          accx[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[mi][0], b2, accx[mi], 0, 0, 0);
          accx[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[mi][1], b1, accx[mi], 0, 0, 0);
          acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[mi][0], b1, acc[mi], 0, 0, 0);
        }
      }
      stashA((J + 2) & 3, (J + 2) & 3);
      __syncthreads();
    };
    for (int c0 = 0; c0 < nchunk; c0 += 8) {           // K % 256 == 0
      step(c0, std::integral_constant<int, 0>(), av0, av1);
      step(c0 + 1, std::integral_constant<int, 1>(), av1, av0);
      step(c0 + 2, std::integral_constant<int, 2>(), av0, av1);
      step(c0 + 3, std::integral_constant<int, 3>(), av1, av0);
      step(c0 + 4, std::integral_constant<int, 4>(), av0, av1);
      step(c0 + 5, std::integral_constant<int, 5>(), av1, av0);
      step(c0 + 6, std::integral_constant<int, 6>(), av0, av1);
      step(c0 + 7, std::integral_constant<int, 7>(), av1, av0);
    }
    float probe = 0.f;
#pragma unroll
    for (int mi = 0; mi < NM; mi++) {
      acc[mi] = acc[mi] + accx[mi] * (1.f / 2048.f);
#pragma unroll
      for (int r = 0; r < 4; r++) probe = nonfinite_probe(probe, acc[mi][r]);
    }
    const bool live = nbk >= 0 && nb < a.N;
    const float bias = (live && a.bias) ? a.bias[nb] : 0.f;
    if (live) {
#pragma unroll
      for (int mi = 0; mi < NM; mi++) {
        const float e[4] = {acc[mi].x, acc[mi].y, acc[mi].z, acc[mi].w};
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int m = 16 * (m_lo + mi) + 4 * kg + r;
          if (m < a.M) a.Cm[(size_t)m * a.ldc + nb] = e[r] + bias;
        }
      }
    }
    // range guard (klstm_math.h): an input or weight beyond the fp16 range left Inf / NaN in this wave's column block -> the block
    // again in plain fp32, stored over what went out above (same lane, same address: ordered)
    if (wave_any(probe != probe)) {
      redo_note(a.redo);
#pragma unroll 1
      for (int q = 0; q < NM * 4; q++) {
        const int m = 16 * (m_lo + (q >> 2)) + 4 * kg + (q & 3);
        if (live && m < a.M) a.Cm[(size_t)m * a.ldc + nb] = redo_dot(a.A + (size_t)m * a.lda, 1, a.B + (size_t)nb * a.ldb, 1, a.K) + bias;
      }
    }
  };
  const int slot = (int)blockIdx.x * 4 + wave, nb_main = (int)gridDim.x * 4, nb_total = (a.N + 15) / 16;
  kpass(slot < nb_total ? slot : -1, 0, std::integral_constant<int, MI>());
  const int xunits = (nb_total - nb_main) * MI;
  if ((int)blockIdx.x * 4 >= xunits) return;
  kpass(slot < xunits ? nb_main + slot / MI : -1, slot < xunits ? slot % MI : 0, std::integral_constant<int, 1>());
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same f16 x 2 product with the input rows RESIDENT for the whole pass and no workgroup barrier inside it (VERDICT r03-r05:
// the variant that had never been built).  k_nt_shared_a16 walks K in 16 barrier-bounded chunk steps (0.75 us each, paced by the rows of A
// requested one step earlier: 20.3 us at 80 x 16624 x 512 whatever the weight stream does).  Here K is cut into four quarters, one per
// wave: wave w keeps ITS quarter of all MI row blocks of A as fp16 planes in registers (MI x KS x 2 operands of 4 registers: 160 at 80 x
// 512; one wave per SIMD), loaded and split once, and streams the same quarter of the workgroup's 64 rows of B (4 column blocks x KS
// k-steps, register ring NB_RING deep, no LDS, no barrier): 15 MFMAs per fetch.  The four waves' partial tiles meet in LDS (80 KB) behind
// ONE barrier and are added in wave order (deterministic), + bias, 256-byte row pieces out.  Sum over K = quarter by quarter: last-bit
// differences to k_nt_shared_a16, same parity bars.  Range guard as there (non-finite output -> that output again in plain fp32).
// MEASURED (tools/t_affprop.py, profiles/r06_affprop_resident.txt) and NOT the default: 80 x 16624 x 512 21.0 us against 19.8 (37 rows: 14.7
// against 16.5; K = 256: 14.5 / 13.9; 9000 columns, 141 of the 256 CUs busy: 17.4 / 14.8) -- with a ring of 6 fetches 32 us (two rounds of
// 260 one-per-CU workgroups), with every request of the pass in flight from the first instruction and the blocks dealt 4 / 5 per workgroup
// 21.0, with 64 contiguous bytes per row and load instruction 20.9.  Barriers, chunk steps and request depth are not what bounds this
// product: both forms bring the same 288 KB into every CU (its 64 rows of W and ALL of A), and a CU ingests 30-55 GB/s whatever is in
// flight (docs/DESIGN_rounds_1-4.md 3d).  Option "direct_nt_shape" = 97 runs it.
// ---------------------------------------------------------------------------------------------------------------------
template <int MI, int KS>
__global__ __launch_bounds__(256) void k_nt_resident_a16(DirectNtArgs a) {
  constexpr int NB = 5, NF = NB * KS;                  // column blocks per workgroup (at most), B fetches per wave: ALL in flight from the start
  extern __shared__ __attribute__((aligned(16))) float part_ra[];   // [4 waves][NB][16 MI rows][16]
  float *part = part_ra;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kg = lane >> 4;
  const int KQ = 32 * KS, k0 = wave * KQ + 4 * kg;     // this wave's quarter of K (K = 4 KQ); of every 32-k step this lane holds k = 4 kg .. + 3 and 16 + 4 kg .. + 3
                                                      // (the same bijection for A and B: every k once; a load instruction then reads 64 CONTIGUOUS bytes of each of
                                                      //  its 16 rows -- 16 line requests instead of 32)
  // the 16-column blocks are dealt evenly: the first `rem` workgroups take one more (16624 columns = 1039 blocks on 256 CUs: 15 x 5 + 241 x 4
  // -- one round of the chip; a 257th workgroup would run alone behind the others)
  const int nblk = (a.N + 15) >> 4, G = (int)gridDim.x, q = nblk / G, rem = nblk - q * G, wg = (int)blockIdx.x;
  const int blk0 = wg * q + (wg < rem ? wg : rem), nbw = q + (wg < rem ? 1 : 0);
  // ---- B: fetch f = (column block j = f / KS, k-step ks = f % KS): rows 16 (blk0 + j) + i16 of B, 8 consecutive k ----
  float4 rb[NF][2];
#pragma unroll
  for (int f = 0; f < NF; f++) {
    const int j = f / KS, ks = f % KS, n = 16 * (blk0 + (j < nbw ? j : 0)) + i16;
    const float *bp = a.B + (size_t)(n < a.N ? n : a.N - 1) * a.ldb + k0 + 32 * ks;
    rb[f][0] = *reinterpret_cast<const float4 *>(bp); rb[f][1] = *reinterpret_cast<const float4 *>(bp + 16);
  }
  __builtin_amdgcn_sched_barrier(0);                   // (every request of the pass is out before anything waits: the scheduler would sink them next to their uses)
  // ---- A: the wave's quarter of every row block, split once ----
  nt_f16x8 af[MI][KS][2];
  {
    float4 ra[MI][KS][2];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const int m = 16 * mi + i16;
        const float *ap = a.A + (size_t)(m < a.M ? m : a.M - 1) * a.lda + k0 + 32 * ks;
        ra[mi][ks][0] = *reinterpret_cast<const float4 *>(ap); ra[mi][ks][1] = *reinterpret_cast<const float4 *>(ap + 16);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const bool on = 16 * mi + i16 < a.M;
        const float4 v0 = keep4(ra[mi][ks][0], on), v1 = keep4(ra[mi][ks][1], on);
        uint4 u1, u2;
        f16_split2_pair(v0.x, v0.y, u1.x, u2.x);
        f16_split2_pair(v0.z, v0.w, u1.y, u2.y);
        f16_split2_pair(v1.x, v1.y, u1.z, u2.z);
        f16_split2_pair(v1.z, v1.w, u1.w, u2.w);
        af[mi][ks][0] = __builtin_bit_cast(nt_f16x8, u1); af[mi][ks][1] = __builtin_bit_cast(nt_f16x8, u2);
      }
  }
  // ---- the pass: up to NB column blocks x KS k-steps, fully unrolled ----
  f32x4 acc[MI], accx[MI];
#pragma unroll
  for (int f = 0; f < NF; f++) {
    const int j = f / KS, ks = f % KS;
    if (j >= nbw) break;                               // (workgroup-uniform; the loads of the absent fifth block went to block 0's rows)
    if (ks == 0) {
#pragma unroll
      for (int mi = 0; mi < MI; mi++) { acc[mi] = (f32x4){0, 0, 0, 0}; accx[mi] = (f32x4){0, 0, 0, 0}; }
    }
    const float4 b0 = rb[f][0], b1v = rb[f][1];
    uint4 u1, u2;
    f16_split2_pair(b0.x, b0.y, u1.x, u2.x);
    f16_split2_pair(b0.z, b0.w, u1.y, u2.y);
    f16_split2_pair(b1v.x, b1v.y, u1.z, u2.z);
    f16_split2_pair(b1v.z, b1v.w, u1.w, u2.w);
    const nt_f16x8 b1 = __builtin_bit_cast(nt_f16x8, u1), b2 = __builtin_bit_cast(nt_f16x8, u2);
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
      accx[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][ks][0], b2, accx[mi], 0, 0, 0);
      accx[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][ks][1], b1, accx[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][ks][0], b1, acc[mi], 0, 0, 0);
    }
    if (ks == KS - 1) {                                // the wave's partial tile of column block j: rows 16 mi + 4 kg + r, column i16
#pragma unroll
      for (int mi = 0; mi < MI; mi++) {
        const f32x4 v = acc[mi] + accx[mi] * (1.f / 2048.f);
#pragma unroll
        for (int r = 0; r < 4; r++) part[(((wave * NB + j) * MI + mi) * 16 + 4 * kg + r) * 16 + i16] = v[r];
      }
    }
  }
  __syncthreads();
  // ---- the four K quarters in wave order, + bias, out: thread = (row, 4 consecutive columns) ----
  bool redo = false;
#pragma unroll 1
  for (int u = tid; u < 16 * MI * 4 * nbw; u += 256) { // units of 4 columns: 4 nbw per row
    const int m = u / (4 * nbw), cq = u - m * 4 * nbw, j = cq >> 2, c4 = (cq & 3) * 4;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const float4 v = *reinterpret_cast<const float4 *>(&part[(((w * NB + j) * MI + (m >> 4)) * 16 + (m & 15)) * 16 + c4]);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const int n = 16 * (blk0 + j) + c4;
    if (m >= a.M || n >= a.N) continue;
    float o[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (n + e >= a.N) continue;
      const float bias = a.bias ? a.bias[n + e] : 0.f;
      float v = o[e];
      if (nonfinite_probe(0.f, v) != 0.f || v != v) {  // range guard: an operand beyond the fp16 range -> this output again in plain fp32
        v = redo_dot(a.A + (size_t)m * a.lda, 1, a.B + (size_t)(n + e) * a.ldb, 1, a.K);
        redo = true;
      }
      o[e] = v + bias;
    }
    float *cp = a.Cm + (size_t)m * a.ldc + n;
    if (n + 4 <= a.N && (a.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Cm) & 15) == 0) *reinterpret_cast<float4 *>(cp) = make_float4(o[0], o[1], o[2], o[3]);
    else
      for (int e = 0; e < 4 && n + e < a.N; e++) cp[e] = o[e];
  }
  if (wave_any(redo)) redo_note(a.redo);
}
bool nt_resident_supported(int M, int N, int K) { return M >= 1 && M <= 80 && N > 8192 && K % 128 == 0 && K >= 128 && K <= 512; }

// Round 6, measured and dropped: the same f16 x 2 product WITHOUT the shared staging of A (k_direct_nt16: every wave on its own -- its 16 rows
// of B and all MI row blocks of A straight from memory into a register ring, B 8 chunks ahead, A 2, loads pinned between the MFMA groups,
// no LDS, no barrier; bit-identical to k_nt_shared_a16).  80 x 16624 over K = 512: 47.8 us against 20.0 (9000 columns: 25.5 against 14.8)
// -- every wave re-reads the 160 KB of A out of L2, 164 MB per launch in 16-line gathers (16 rows x 64 bytes per instruction), and that,
// not the weight stream, sets the time; a row stride of 528 floats instead of 512 gives 31.5 us, other paddings nothing.  Two column
// blocks per wave (half the A traffic) do not fit two waves per SIMD next to a B ring deep enough for HBM.  The LDS-shared form stays.
static int g_nt_shared = 1;
static int g_nt_ni = 2, g_nt_waves = 1;   //    // measured at 80 x 16624 x 512 (tools/nt_sweep.py): 29.5 us; 1x1 36.7, 2x2 34.0, 4x1 49.4; tiled kernel 37.9
void set_direct_nt_shape(int ni, int waves) {       // option value 0: wide results on k_direct_nt again (A-B); 99: on the fp32 k_nt_shared_a;
  g_nt_shared = (ni == 0 && waves == 0) ? 0 : (ni == 9 && waves == 9) ? 2 : (ni == 9 && waves == 8) ? 3 : (ni == 9 && waves == 7) ? 4 : 1;   // 97: k_nt_resident_a16 (A-B)     // 98 (and every other value): on k_nt_shared_a16
  if (g_nt_shared != 1) return;
  g_nt_ni = ni == 1 || ni == 2 ? ni : 4; g_nt_waves = waves >= 1 && waves <= 4 ? waves : 2;
}

bool direct_nt_supported(int M, int N, int K, const float *A, int lda, const float *B, int ldb) {
  return g_fold_direct != 0 && M >= 1 && M <= 80 && N >= 64 && K % (32 * FD) == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
         (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
}

hipError_t launch_direct_nt(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *Cm, int ldc,
                            const float *bias, hipStream_t st, LaunchProbe pr) {
#define KS_LAUNCH(MI_) do { if (pr.start) hipExtLaunchKernelGGL((k_direct_nt_ks<MI_>), grid, block, 0, st, pr.start, pr.stop, 0, a); \
                            else hipLaunchKernelGGL((k_direct_nt_ks<MI_>), grid, block, 0, st, a); } while (0)
  DirectNtArgs a{M, N, K, A, lda, B, ldb, Cm, ldc, bias, redo_counters() ? redo_counters() + REDO_NT : nullptr};
  const bool f16_ok = redo_count(REDO_NT) == 0;          // (once the range guard has fired, wide results stay on the fp32 form)
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
  // wide result: A shared through LDS.  tools/t_affprop.py, 80 rows over K = 512, 16624 / 9000 columns: k_direct_nt 29.7 / 26.7 us,
  // k_nt_shared_a (fp32 MFMA, one wave per SIMD, the blocks past 1024 as a second pass) 29.1 / 18.9 us, k_nt_shared_a16 (f16 x 2
  // operands split on the fly, two waves per SIMD so that all 260 workgroups are resident at once) 20.6 / 14.5 us: the default.
  // direct_nt_shape = 99 asks for the fp32 form, 0 for k_direct_nt
  if (g_nt_shared == 4 && f16_ok && nt_resident_supported(M, N, K)) {   // round 6, option value 97 (A-B runs): A resident in registers, K in four quarters (one per wave)
    const int nblk = (N + 15) / 16, g4 = (nblk + 3) / 4, g5 = (nblk + 4) / 5;
    const dim3 grid(g4 <= 256 ? g4 : (g5 > 256 ? g5 : 256)), block(256);   // 4 column blocks per workgroup; 5 for some when that keeps the launch in one round
    const int mi_ = (M + 15) / 16, ks_ = K / 128;
    const size_t shm = (size_t)4 * 5 * mi_ * 16 * 16 * sizeof(float);
    // (the dynamic-LDS opt-in is per device and costs the host ~10 us per call: once per (device, instance))
    static std::mutex ra_mu;
    static std::set<int> ra_done;
    int ra_dev = 0;
    (void)hipGetDevice(&ra_dev);
#define RA_GO(MI_, KS_) do { if (shm > 64 * 1024) { std::lock_guard<std::mutex> lk(ra_mu); if (ra_done.insert(ra_dev * 64 + MI_ * 8 + KS_).second) \
                               (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nt_resident_a16<MI_, KS_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } \
                             if (pr.start) hipExtLaunchKernelGGL((k_nt_resident_a16<MI_, KS_>), grid, block, shm, st, pr.start, pr.stop, 0, a); \
                             else hipLaunchKernelGGL((k_nt_resident_a16<MI_, KS_>), grid, block, shm, st, a); return hipGetLastError(); } while (0)
#define RA_MI(MI_) do { switch (ks_) { case 1: RA_GO(MI_, 1); case 2: RA_GO(MI_, 2); case 3: RA_GO(MI_, 3); default: RA_GO(MI_, 4); } } while (0)
    switch (mi_) {
      case 1: RA_MI(1);
      case 2: RA_MI(2);
      case 3: RA_MI(3);
      case 4: RA_MI(4);
      default: RA_MI(5);
    }
#undef RA_MI
#undef RA_GO
  }
  if (N > 8192 && K % 256 == 0 && g_nt_shared) {
    const int nbt = (N + 15) / 16, mi_ = (M + 15) / 16;
    int wgs = (nbt + 3) / 4;
    if ((g_nt_shared == 2 || !f16_ok) && wgs > 256 && (nbt - 1024) * mi_ <= 1024) wgs = 256;      // the blocks past 1024 go out as extra (block, row block) units
    const dim3 grid(wgs), block(256);
#define SA_GO(MI_) do { if (g_nt_shared != 2 && f16_ok) { if (pr.start) hipExtLaunchKernelGGL((k_nt_shared_a16<MI_>), grid, block, 0, st, pr.start, pr.stop, 0, a); \
                                                else hipLaunchKernelGGL((k_nt_shared_a16<MI_>), grid, block, 0, st, a); } \
                        else if (pr.start) hipExtLaunchKernelGGL((k_nt_shared_a<MI_>), grid, block, 0, st, pr.start, pr.stop, 0, a); \
                        else hipLaunchKernelGGL((k_nt_shared_a<MI_>), grid, block, 0, st, a); } while (0)
    switch ((M + 15) / 16) {
      case 1: SA_GO(1); break;
      case 2: SA_GO(2); break;
      case 3: SA_GO(3); break;
      case 4: SA_GO(4); break;
      case 5: SA_GO(5); break;
      default: return hipErrorInvalidValue;
    }
#undef SA_GO
    return hipGetLastError();
  }
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
