// kaldi-lstm_amd/csrc/klstm_kernels.h -- host-callable launchers of the gfx950 kernels.
// Every launcher enqueues on `st` and returns the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>

namespace klstm {

// Dimensions + device pointers shared by the step kernels.  Planes are time-major
// (row = tb*S + s, tb = time block 0..T+1 like the reference slab rows, ...streams.h:229-243)
// but split per column group so that every kernel streams unit-stride rows:
//   gifo [(T+2)S x 4C]  activated g|i|f|o         cc, hh, mm [(T+2)S x C]     rr [(T+2)S x R]
//   dgifo[(T+2)S x 4C]  d(pre-activation) g|i|f|o dc [(T+2)S x C]             dr [(T+2)S x R]
struct Dims { int I, C, R, S, T; };

struct FwdPtrs {
  const float *wx, *wr, *bias, *pi, *pf, *po, *wm;   // canonical [N x K] row-major weights
  float *gifo, *cc, *hh, *mm, *rr;
  float *prev_c, *prev_r;                            // carried state [S x C], [S x R]
};

struct BwdPtrs {
  const float *wrT, *wmT;                            // transposed copies [R x 4C], [C x R]
  const float *pi, *pf, *po;
  const float *gifo, *cc, *hh;
  float *dgifo, *dc, *dr;
  float *dr_part;                                    // split-K slabs [KS][S][R]
  int ks;                                            // number of slabs
};

// optional per-launch timing through hipExtLaunchKernelGGL start/stop events
struct LaunchProbe { hipEvent_t start = nullptr, stop = nullptr; };

hipError_t launch_begin(const Dims &d, const FwdPtrs &p, hipStream_t st, LaunchProbe pr = {});
hipError_t launch_end(const Dims &d, const FwdPtrs &p, hipStream_t st, LaunchProbe pr = {});
hipError_t launch_gates_step(const Dims &d, const FwdPtrs &p, int t, hipStream_t st, LaunchProbe pr = {});
hipError_t launch_proj_step(const Dims &d, const FwdPtrs &p, int t, float *out, int out_stride,
                            hipStream_t st, LaunchProbe pr = {});
hipError_t launch_dr_step(const Dims &d, const BwdPtrs &p, int t, hipStream_t st, LaunchProbe pr = {});
hipError_t launch_dm_step(const Dims &d, const BwdPtrs &p, int t, const float *out_diff, int od_stride,
                          hipStream_t st, LaunchProbe pr = {});

// Generic batched GEMM  C[MxN] = beta*C + op(A)*op(B) (+ bias[n]).
//   transA: A stored [K x M] (lda)   else [M x K]
//   transB: B stored [N x K] (ldb)   else [K x N]
hipError_t launch_gemm(bool transA, bool transB, int M, int N, int K, const float *A, int lda,
                       const float *B, int ldb, float beta, float *Cm, int ldc, const float *bias,
                       hipStream_t st, LaunchProbe pr = {});

// bias / peephole gradient reductions (...streams.h:474-484), dst = beta*dst + sum
hipError_t launch_vec_grads(const Dims &d, const float *dgifo, const float *cc, float beta,
                            float *g_bias, float *g_pi, float *g_pf, float *g_po, hipStream_t st,
                            LaunchProbe pr = {});

// elementwise blob kernels
hipError_t launch_apply_momentum(float *corr, const float *grad, float mmt, long n, hipStream_t st, LaunchProbe pr = {});
hipError_t launch_update(float *param, float *corr, float lr, float clip, long n, hipStream_t st, LaunchProbe pr = {});
hipError_t launch_transpose(const float *src, int rows, int cols, float *dst, hipStream_t st, LaunchProbe pr = {});
hipError_t launch_zero_rows(float *base, int ld, const int *flags_dev, int nrows, int ncols, hipStream_t st);

int dr_split_k(const Dims &d);   // number of split-K slabs launch_dr_step writes

}  // namespace klstm
