// kaldi-lstm_amd/csrc/klstm_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the
// LstmProjectedStreams hot path.  fp32 throughout (Kaldi BaseFloat); every contraction runs on
// the f32-input MFMA  v_mfma_f32_16x16x4_f32  (exact fp32 FMA chain, 157 TF peak), 64-wide waves.
//
// What each kernel replaces in the reference (google/nnet/bd-nnet-lstm-projected-streams.h):
//   k_gates_step : per-step  AddMatMat(r(t-1),W_gifo_r^T) + 2x AddMatDiagVec + Sigmoid x3 + Tanh x2 +
//                  3x AddMatDotMat + ApplyFloor/Ceiling                      (:275-309, 14 launches)
//   k_proj_step  : per-step  y_r = y_m * W_r_m^T (:312) + the copy into `out` (:328)
//   k_dr_step    : per-step  d_r += DGIFO(t+1) * W_gifo_r (:391), split-K partial slabs
//   k_dm_step    : per-step  d_m = d_r * W_r_m (:408) + the 15 elementwise launches (:411-440)
//   k_gemm       : the batched products outside the time loop (:246 + bias :259, :457, :468, :471, :486)
//   k_vec_grads  : AddRowSumMat / AddDiagMatMat x3 (:474-484)
//   k_update, k_apply_momentum : Update (:504-512) / DP-mode momentum
//
// Skinny-GEMM layout used by the four step kernels ("weights on M, streams on N"):
//   D[16 weight rows][16 streams] += A[row][k] * B[k][stream],  A lane l: row l&15, k-group l>>4;
//   a lane loads 8 consecutive k (two dwordx4) of its weight row / stream row per 32-wide K chunk
//   and feeds them to 8 MFMAs; the 8 waves of a workgroup split K, partial tiles are summed in a
//   fixed order through LDS (deterministic), and the fused LSTM cell math runs on the summed tile
//   in registers.  D lane l holds stream l&15, rows 4*(l>>4)+{0..3}.
#include "klstm_kernels.h"

#include <hip/hip_ext.h>
#include <stdint.h>

namespace klstm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int NW = 8;        // waves per workgroup in the step kernels (K split)
constexpr int KCH = 32;      // K chunk one wave consumes per iteration (4 k-groups x 8)

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// scalar math: the overflow-safe forms Kaldi's CPU path uses, no FMA
// contraction so that the elementwise results track the CPU formulation to the last bit where possible.
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ float k_sigmoid(float x) {
  if (x > 0.f) return 1.f / (1.f + expf(-x));
  const float ex = expf(x);
  return ex / (ex + 1.f);
}
__device__ __forceinline__ float k_tanh(float x) {
  if (x > 0.f) { const float inv = expf(-x); return -1.f + 2.f / (1.f + inv * inv); }
  const float e = expf(x);
  return 1.f - 2.f / (1.f + e * e);
}
// DiffSigmoid / DiffTanh with the reference's double literal (kaldi-matrix.cc:2562-2593)
__device__ __forceinline__ float k_diff_sigmoid(float d, float y) {
  return (float)((double)(d * y) * (1.0 - (double)y));
}
__device__ __forceinline__ float k_diff_tanh(float d, float y) {
  return (float)((double)d * (1.0 - (double)(y * y)));
}

__device__ __forceinline__ void load8(const float *__restrict__ row, int k, int K, bool ok, bool vec,
                                      float (&v)[8]) {
  if (ok && vec && k + 8 <= K) {
    const float4 a = *reinterpret_cast<const float4 *>(row + k);
    const float4 b = *reinterpret_cast<const float4 *>(row + k + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (ok && k + j < K) ? row[k + j] : 0.f;
  }
}

// Sum the NW per-wave partial tiles of s-tile `nt` in fixed wave order (deterministic).
template <int NT>
__device__ __forceinline__ f32x4 reduce_tile(const f32x4 (*red)[NT][64], int nt, int lane) {
  f32x4 v = red[0][nt][lane];
#pragma unroll
  for (int w = 1; w < NW; w++) v += red[w][nt][lane];
  return v;
}

// ---------------------------------------------------------------------------------------------
// state bridge (...streams.h:231, :331): only the c and r column groups are ever consumed.
// ---------------------------------------------------------------------------------------------
__global__ void k_begin(int S, int C, int R, const float *__restrict__ prev_c, const float *__restrict__ prev_r,
                        float *__restrict__ cc, float *__restrict__ rr) {
  const int n = S * C + S * R;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (i < S * C) cc[i] = prev_c[i]; else rr[i - S * C] = prev_r[i - S * C];
  }
}
__global__ void k_end(int S, int C, int R, int T, float *__restrict__ prev_c, float *__restrict__ prev_r,
                      const float *__restrict__ cc, const float *__restrict__ rr) {
  const int n = S * C + S * R;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (i < S * C) prev_c[i] = cc[(size_t)T * S * C + i];
    else prev_r[i - S * C] = rr[(size_t)T * S * R + (i - S * C)];
  }
}

// ---------------------------------------------------------------------------------------------
// forward step 1/2: gates + cell.  One workgroup = 4 cells x 4 gates (16 weight rows) x 16*NT streams.
// tile row i -> (cell c0 + i/4, gate i%4) so that after the MFMA a lane owns g,i,f,o of ONE
// (cell, stream) pair in its four accumulator registers and the cell math is lane-local.
// ---------------------------------------------------------------------------------------------
struct GatesArgs {
  int C, R, S, t;
  const float *wr, *pi, *pf, *po;
  float *gifo, *cc, *hh, *mm;
  const float *rr;
  int vecW, vecB;
};

template <int NT>
__global__ __launch_bounds__(NW * 64) void k_gates_step(GatesArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int C = a.C, R = a.R, S = a.S, t = a.t;
  const int c0 = blockIdx.x * 4;
  const int sbase = blockIdx.y * 16 * NT;

  const int cell_a = c0 + (i16 >> 2), gate_a = i16 & 3;
  const bool row_ok = cell_a < C;
  const float *wrow = a.wr + (size_t)(gate_a * C + (row_ok ? cell_a : 0)) * R;
  const float *rprev = a.rr + (size_t)(t - 1) * S * R;

  f32x4 acc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) { acc[nt][0] = (f32x4){0, 0, 0, 0}; acc[nt][1] = (f32x4){0, 0, 0, 0}; }

  for (int ch = wave; ch * KCH < R; ch += NW) {
    const int k = ch * KCH + kg * 8;
    float av[8];
    load8(wrow, k, R, row_ok, a.vecW, av);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int s = sbase + nt * 16 + i16;
      float bv[8];
      load8(rprev + (size_t)(s < S ? s : 0) * R, k, R, s < S, a.vecB, bv);
#pragma unroll
      for (int j = 0; j < 8; j++) acc[nt][j & 1] = MFMA16(av[j], bv[j], acc[nt][j & 1]);
    }
  }

  __shared__ f32x4 red[NW][NT][64];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) red[wave][nt][lane] = acc[nt][0] + acc[nt][1];
  __syncthreads();

  for (int nt = wave; nt < NT; nt += NW) {
    const f32x4 v = reduce_tile<NT>(red, nt, lane);
    const int cell = c0 + kg;
    const int s = sbase + nt * 16 + i16;
    if (cell < C && s < S) {
      const size_t row = (size_t)t * S + s, rowp = row - S;
      float *gp = a.gifo + row * 4 * C + cell;
      const float cp = a.cc[rowp * C + cell];
      float ag = v.x + gp[0];
      float ai = v.y + gp[C];
      float af = v.z + gp[2 * C];
      float ao = v.w + gp[3 * C];
      ai += a.pi[cell] * cp;                       // :278
      af += a.pf[cell] * cp;                       // :281
      const float gi = k_sigmoid(ai), gf = k_sigmoid(af), gg = k_tanh(ag);   // :284-288
      float c = gg * gi;                           // :291
      c = c + cp * gf;                             // :294
      c = c < -50.f ? -50.f : c;                   // :296
      c = c > 50.f ? 50.f : c;                     // :297
      const float h = k_tanh(c);                   // :300
      ao += a.po[cell] * c;                        // :303
      const float go = k_sigmoid(ao);              // :306
      const float m = h * go;                      // :309
      gp[0] = gg; gp[C] = gi; gp[2 * C] = gf; gp[3 * C] = go;
      a.cc[row * C + cell] = c;
      a.hh[row * C + cell] = h;
      a.mm[row * C + cell] = m;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward step 2/2: recurrent projection r(t) = m(t) * W_r_m^T (:312), also written to `out` (:328).
// One workgroup = 16 projection rows x 16*NT streams; lane owns 4 consecutive r columns.
// ---------------------------------------------------------------------------------------------
struct ProjArgs {
  int C, R, S, t;
  const float *wm, *mm;
  float *rr, *out;
  int out_stride;
  int vecW, vecB, vecR, vecOut;
};

template <int NT>
__global__ __launch_bounds__(NW * 64) void k_proj_step(ProjArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int C = a.C, R = a.R, S = a.S, t = a.t;
  const int n0 = blockIdx.x * 16;
  const int sbase = blockIdx.y * 16 * NT;
  const bool row_ok = n0 + i16 < R;
  const float *wrow = a.wm + (size_t)(row_ok ? n0 + i16 : 0) * C;
  const float *mrow = a.mm + (size_t)t * S * C;

  f32x4 acc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) { acc[nt][0] = (f32x4){0, 0, 0, 0}; acc[nt][1] = (f32x4){0, 0, 0, 0}; }

  for (int ch = wave; ch * KCH < C; ch += NW) {
    const int k = ch * KCH + kg * 8;
    float av[8];
    load8(wrow, k, C, row_ok, a.vecW, av);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int s = sbase + nt * 16 + i16;
      float bv[8];
      load8(mrow + (size_t)(s < S ? s : 0) * C, k, C, s < S, a.vecB, bv);
#pragma unroll
      for (int j = 0; j < 8; j++) acc[nt][j & 1] = MFMA16(av[j], bv[j], acc[nt][j & 1]);
    }
  }
  __shared__ f32x4 red[NW][NT][64];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) red[wave][nt][lane] = acc[nt][0] + acc[nt][1];
  __syncthreads();

  for (int nt = wave; nt < NT; nt += NW) {
    const f32x4 v = reduce_tile<NT>(red, nt, lane);
    const int s = sbase + nt * 16 + i16;
    const int n = n0 + 4 * kg;
    if (s < S && n < R) {
      float *rp = a.rr + ((size_t)t * S + s) * R + n;
      float *op = a.out + (size_t)((t - 1) * S + s) * a.out_stride + n;
      if (n + 3 < R && a.vecR) *reinterpret_cast<float4 *>(rp) = make_float4(v.x, v.y, v.z, v.w);
      else { const float e[4] = {v.x, v.y, v.z, v.w}; for (int j = 0; j < 4 && n + j < R; j++) rp[j] = e[j]; }
      if (n + 3 < R && a.vecOut) *reinterpret_cast<float4 *>(op) = make_float4(v.x, v.y, v.z, v.w);
      else { const float e[4] = {v.x, v.y, v.z, v.w}; for (int j = 0; j < 4 && n + j < R; j++) op[j] = e[j]; }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward step 1/2: partial d_r(t) = DGIFO(t+1) * W_gifo_r (:391) over one K slice of the 4C gate
// rows.  grid (R/16, stream groups, KS); slab ks holds the partial sum of its slice.  The consumer
// (k_dm_step) adds the slabs in fixed order together with out_diff(t) (:367).
// ---------------------------------------------------------------------------------------------
struct DrArgs {
  int C, R, S, t;
  const float *wrT;      // [R x 4C]
  const float *dgifo;
  float *part;           // [KS][S][R]
  int klen;              // K slice length (multiple of KCH)
  int vecW, vecB, vecR;
};

template <int NT>
__global__ __launch_bounds__(NW * 64) void k_dr_step(DrArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int R = a.R, S = a.S, K = 4 * a.C;
  const int n0 = blockIdx.x * 16;
  const int sbase = blockIdx.y * 16 * NT;
  const int ks = blockIdx.z;
  const int kbeg = ks * a.klen;
  const int kend = min(K, kbeg + a.klen);
  const bool row_ok = n0 + i16 < R;
  const float *wrow = a.wrT + (size_t)(row_ok ? n0 + i16 : 0) * K;
  const float *drow = a.dgifo + (size_t)(a.t + 1) * S * K;

  f32x4 acc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) { acc[nt][0] = (f32x4){0, 0, 0, 0}; acc[nt][1] = (f32x4){0, 0, 0, 0}; }

  for (int kc = kbeg + wave * KCH; kc < kend; kc += NW * KCH) {
    const int k = kc + kg * 8;
    float av[8];
    load8(wrow, k, kend, row_ok, a.vecW, av);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int s = sbase + nt * 16 + i16;
      float bv[8];
      load8(drow + (size_t)(s < S ? s : 0) * K, k, kend, s < S, a.vecB, bv);
#pragma unroll
      for (int j = 0; j < 8; j++) acc[nt][j & 1] = MFMA16(av[j], bv[j], acc[nt][j & 1]);
    }
  }
  __shared__ f32x4 red[NW][NT][64];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) red[wave][nt][lane] = acc[nt][0] + acc[nt][1];
  __syncthreads();

  for (int nt = wave; nt < NT; nt += NW) {
    const f32x4 v = reduce_tile<NT>(red, nt, lane);
    const int s = sbase + nt * 16 + i16;
    const int n = n0 + 4 * kg;
    if (s < S && n < R) {
      float *pp = a.part + ((size_t)ks * S + s) * R + n;
      if (n + 3 < R && a.vecR) *reinterpret_cast<float4 *>(pp) = make_float4(v.x, v.y, v.z, v.w);
      else { const float e[4] = {v.x, v.y, v.z, v.w}; for (int j = 0; j < 4 && n + j < R; j++) pp[j] = e[j]; }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward step 2/2: d_r(t) = out_diff(t) + sum of slabs; d_m = d_r * W_r_m (:408) and the whole
// elementwise BPTT cell math (:411-440).  One workgroup = 16 cells x 16*NT streams; lane owns 4
// consecutive cells of one stream.  Workgroup x==0 also materialises d_r(t) (needed by the
// W_r_m gradient, :486).
// ---------------------------------------------------------------------------------------------
struct DmArgs {
  int C, R, S, T, t;
  const float *wmT;       // [C x R]
  const float *pi, *pf, *po;
  const float *gifo, *cc, *hh;
  float *dgifo, *dc, *dr;
  const float *part;      // [KS][S][R]
  int nslab;              // 0 at t == T (the t+1 block is all zero, :351)
  const float *out_diff;
  int od_stride;
  int vecW, vecR, vecOD, vecC;
};

__device__ __forceinline__ void load_dr8(const DmArgs &a, int s, int k, bool ok, float (&v)[8]) {
  const int R = a.R, S = a.S;
  load8(a.out_diff + (size_t)((a.t - 1) * S + (ok ? s : 0)) * a.od_stride, k, R, ok, a.vecOD, v);
  for (int ks = 0; ks < a.nslab; ks++) {
    float p[8];
    load8(a.part + ((size_t)ks * S + (ok ? s : 0)) * R, k, R, ok, a.vecR, p);
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] += p[j];
  }
}

template <int NT>
__global__ __launch_bounds__(NW * 64) void k_dm_step(DmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int C = a.C, R = a.R, S = a.S, t = a.t;
  const int c0 = blockIdx.x * 16;
  const int sbase = blockIdx.y * 16 * NT;
  const bool row_ok = c0 + i16 < C;
  const float *wrow = a.wmT + (size_t)(row_ok ? c0 + i16 : 0) * R;

  f32x4 acc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) { acc[nt][0] = (f32x4){0, 0, 0, 0}; acc[nt][1] = (f32x4){0, 0, 0, 0}; }

  for (int ch = wave; ch * KCH < R; ch += NW) {
    const int k = ch * KCH + kg * 8;
    float av[8];
    load8(wrow, k, R, row_ok, a.vecW, av);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int s = sbase + nt * 16 + i16;
      float bv[8];
      load_dr8(a, s, k, s < S, bv);
      if (blockIdx.x == 0 && s < S) {            // materialise d_r(t) exactly once
        float *dp = a.dr + ((size_t)t * S + s) * R + k;
        if (a.vecR && k + 8 <= R) {
          *reinterpret_cast<float4 *>(dp) = make_float4(bv[0], bv[1], bv[2], bv[3]);
          *reinterpret_cast<float4 *>(dp + 4) = make_float4(bv[4], bv[5], bv[6], bv[7]);
        } else {
          for (int j = 0; j < 8 && k + j < R; j++) dp[j] = bv[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) acc[nt][j & 1] = MFMA16(av[j], bv[j], acc[nt][j & 1]);
    }
  }
  __shared__ f32x4 red[NW][NT][64];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) red[wave][nt][lane] = acc[nt][0] + acc[nt][1];
  __syncthreads();

  for (int nt = wave; nt < NT; nt += NW) {
    const f32x4 v = reduce_tile<NT>(red, nt, lane);
    const int s = sbase + nt * 16 + i16;
    const int cb = c0 + 4 * kg;
    if (s >= S || cb >= C) continue;
    const size_t row = (size_t)t * S + s, rown = row + S, rowp = row - S;
    const bool last = (t == a.T);
    const float dm[4] = {v.x, v.y, v.z, v.w};
    float og[4], oi[4], of[4], oo[4], oc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = cb + j;
      if (c >= C) { og[j] = oi[j] = of[j] = oo[j] = oc[j] = 0.f; continue; }
      const float *yp = a.gifo + row * 4 * C + c;
      const float yg = yp[0], yi = yp[C], yf = yp[2 * C], yo = yp[3 * C];
      const float yh = a.hh[row * C + c];
      const float cprev = a.cc[rowp * C + c];
      float dc_n = 0.f, f_n = 0.f, di_n = 0.f, df_n = 0.f;
      if (!last) {
        dc_n = a.dc[rown * C + c];
        f_n = a.gifo[rown * 4 * C + 2 * C + c];
        di_n = a.dgifo[rown * 4 * C + C + c];
        df_n = a.dgifo[rown * 4 * C + 2 * C + c];
      }
      const float d_h = k_diff_tanh(dm[j] * yo, yh);           // :411-412
      const float d_o = k_diff_sigmoid(dm[j] * yh, yo);        // :415-416
      float d_c = d_h;                                         // :424
      d_c = d_c + dc_n * f_n;                                  // :425
      d_c = d_c + a.pi[c] * di_n;                              // :426
      d_c = d_c + a.pf[c] * df_n;                              // :427
      d_c = d_c + a.po[c] * d_o;                               // :428
      of[j] = k_diff_sigmoid(d_c * cprev, yf);                 // :431-432
      oi[j] = k_diff_sigmoid(d_c * yg, yi);                    // :435-436
      og[j] = k_diff_tanh(d_c * yi, yg);                       // :439-440
      oo[j] = d_o;
      oc[j] = d_c;
    }
    float *dp = a.dgifo + row * 4 * C + cb;
    float *dcp = a.dc + row * C + cb;
    if (a.vecC && cb + 3 < C) {
      *reinterpret_cast<float4 *>(dp) = make_float4(og[0], og[1], og[2], og[3]);
      *reinterpret_cast<float4 *>(dp + C) = make_float4(oi[0], oi[1], oi[2], oi[3]);
      *reinterpret_cast<float4 *>(dp + 2 * C) = make_float4(of[0], of[1], of[2], of[3]);
      *reinterpret_cast<float4 *>(dp + 3 * C) = make_float4(oo[0], oo[1], oo[2], oo[3]);
      *reinterpret_cast<float4 *>(dcp) = make_float4(oc[0], oc[1], oc[2], oc[3]);
    } else {
      for (int j = 0; j < 4 && cb + j < C; j++) {
        dp[j] = og[j]; dp[C + j] = oi[j]; dp[2 * C + j] = of[j]; dp[3 * C + j] = oo[j]; dcp[j] = oc[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// generic LDS-tiled MFMA GEMM for the batched products outside the time loop.
// 64x64 output tile, BK = 16, 256 threads = 4 waves (2x2), each wave 32x32 = 2x2 MFMA tiles.
// ---------------------------------------------------------------------------------------------
constexpr int GT = 64, GK = 16, GLD = 80;   // LDS row stride 80 floats: k-groups land on disjoint banks

struct GemmArgs {
  int M, N, K;
  const float *A; int lda;
  const float *B; int ldb;
  float beta;
  float *Cm; int ldc;
  const float *bias;
  int vecA, vecB;
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
  __shared__ float As[GK][GLD];
  __shared__ float Bs[GK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int M = g.M, N = g.N, K = g.K;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0, 0, 0, 0};

  for (int k0 = 0; k0 < K; k0 += GK) {
    // ---- stage A tile into As[k][m] ----
    if (TA) {   // A stored [K x M]
      const int k = tid >> 4, m4 = (tid & 15) * 4;
      const int gk = k0 + k, gm = m0 + m4;
      float v[4] = {0, 0, 0, 0};
      if (gk < K) {
        const float *p = g.A + (size_t)gk * g.lda + gm;
        if (g.vecA && gm + 3 < M) { const float4 q = *reinterpret_cast<const float4 *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else { for (int j = 0; j < 4; j++) if (gm + j < M) v[j] = p[j]; }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) As[k][m4 + j] = v[j];
    } else {    // A stored [M x K]
      const int m = tid >> 2, k4 = (tid & 3) * 4;
      const int gm = m0 + m, gk = k0 + k4;
      float v[4] = {0, 0, 0, 0};
      if (gm < M) {
        const float *p = g.A + (size_t)gm * g.lda + gk;
        if (g.vecA && gk + 3 < K) { const float4 q = *reinterpret_cast<const float4 *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else { for (int j = 0; j < 4; j++) if (gk + j < K) v[j] = p[j]; }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) As[k4 + j][m] = v[j];
    }
    // ---- stage B tile into Bs[k][n] ----
    if (TB) {   // B stored [N x K]
      const int n = tid >> 2, k4 = (tid & 3) * 4;
      const int gn = n0 + n, gk = k0 + k4;
      float v[4] = {0, 0, 0, 0};
      if (gn < N) {
        const float *p = g.B + (size_t)gn * g.ldb + gk;
        if (g.vecB && gk + 3 < K) { const float4 q = *reinterpret_cast<const float4 *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else { for (int j = 0; j < 4; j++) if (gk + j < K) v[j] = p[j]; }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) Bs[k4 + j][n] = v[j];
    } else {    // B stored [K x N]
      const int k = tid >> 4, n4 = (tid & 15) * 4;
      const int gk = k0 + k, gn = n0 + n4;
      float v[4] = {0, 0, 0, 0};
      if (gk < K) {
        const float *p = g.B + (size_t)gk * g.ldb + gn;
        if (g.vecB && gn + 3 < N) { const float4 q = *reinterpret_cast<const float4 *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else { for (int j = 0; j < 4; j++) if (gn + j < N) v[j] = p[j]; }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) Bs[k][n4 + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK / 4; kk++) {
      const int k = kk * 4 + kg;
      const float a0 = As[k][wr * 32 + i16], a1 = As[k][wr * 32 + 16 + i16];
      const float b0 = Bs[k][wc * 32 + i16], b1 = Bs[k][wc * 32 + 16 + i16];
      acc[0][0] = MFMA16(a0, b0, acc[0][0]);
      acc[0][1] = MFMA16(a0, b1, acc[0][1]);
      acc[1][0] = MFMA16(a1, b0, acc[1][0]);
      acc[1][1] = MFMA16(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
      const int n = n0 + wc * 32 + ni * 16 + i16;
      if (n >= N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
      const float e[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = m0 + wr * 32 + mi * 16 + 4 * kg + r;
        if (m >= M) continue;
        float *cp = g.Cm + (size_t)m * g.ldc + n;
        float val = e[r] + bv;
        if (g.beta != 0.f) val = g.beta * *cp + val;
        *cp = val;
      }
    }
}

// ---------------------------------------------------------------------------------------------
// bias and peephole gradients (...streams.h:474-484): column sums over the T*S frame rows.
// block (64 columns x 8 row groups); one column of the 4C gate axis per thread.x.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_vec_grads(int C, int S, int T, const float *__restrict__ dgifo,
                                                   const float *__restrict__ cc, float beta,
                                                   float *__restrict__ g_bias, float *__restrict__ g_pi,
                                                   float *__restrict__ g_pf, float *__restrict__ g_po) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int rows = T * S;
  float sb = 0.f, sp = 0.f;
  if (col < 4 * C) {
    const int gate = col / C, cell = col - gate * C;
    // DI/DF[1..T] pair with YC[0..T-1]; DO[1..T] pairs with YC[1..T]
    const float *cbase = cc + (gate == 3 ? (size_t)S * C : 0) + cell;
    for (int r = ty; r < rows; r += 8) {
      const float d = dgifo[(size_t)(S + r) * 4 * C + col];
      sb += d;
      if (gate != 0) sp += d * cbase[(size_t)r * C];
    }
  }
  __shared__ float rb[8][64], rp[8][64];
  rb[ty][tx] = sb; rp[ty][tx] = sp;
  __syncthreads();
  if (ty == 0 && col < 4 * C) {
    for (int w = 1; w < 8; w++) { sb += rb[w][tx]; sp += rp[w][tx]; }
    const int gate = col / C, cell = col - gate * C;
    g_bias[col] = (beta != 0.f ? beta * g_bias[col] : 0.f) + sb;
    float *gp = gate == 1 ? g_pi : gate == 2 ? g_pf : gate == 3 ? g_po : nullptr;
    if (gp) gp[cell] = (beta != 0.f ? beta * gp[cell] : 0.f) + sp;
  }
}

__global__ void k_apply_momentum(float *__restrict__ corr, const float *__restrict__ grad, float mmt, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    corr[i] = mmt * corr[i] + grad[i];
}
// Update (:504-512), optional in-place +-clip of corr first (standard/...:480-493)
__global__ void k_update(float *__restrict__ p, float *__restrict__ corr, float lr, float clip, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float c = corr[i];
    if (clip > 0.f) { c = c < -clip ? -clip : c; c = c > clip ? clip : c; corr[i] = c; }
    p[i] = p[i] + (-lr) * c;
  }
}
__global__ void k_transpose(const float *__restrict__ src, int rows, int cols, float *__restrict__ dst) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = by + j, c = bx + threadIdx.x;
    tile[j][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = bx + j, r = by + threadIdx.x;     // dst[c][r]
    if (c < cols && r < rows) dst[(size_t)c * rows + r] = tile[threadIdx.x][j];
  }
}
__global__ void k_zero_rows(float *base, int ld, const int *flags, int nrows, int ncols) {
  const int r = blockIdx.x;
  if (r >= nrows || flags[r] != 1) return;
  for (int j = threadIdx.x; j < ncols; j += blockDim.x) base[(size_t)r * ld + j] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
#define KLAUNCH(kern, grid, block, st, pr, ...)                                               \
  do {                                                                                        \
    if ((pr).start) hipExtLaunchKernelGGL(kern, grid, block, 0, st, (pr).start, (pr).stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, 0, st, __VA_ARGS__);                           \
    return hipGetLastError();                                                                 \
  } while (0)

// pick the number of 16-stream tiles a workgroup handles
static inline int pick_nt(int S) {
  const int tiles = cdiv(S, 16);
  return tiles >= 8 ? 8 : tiles >= 4 ? 4 : tiles >= 2 ? 2 : 1;
}
#define NT_DISPATCH(KERN, nt, grid, st, pr, args)                          \
  switch (nt) {                                                            \
    case 1: KLAUNCH(KERN<1>, grid, dim3(NW * 64), st, pr, args);           \
    case 2: KLAUNCH(KERN<2>, grid, dim3(NW * 64), st, pr, args);           \
    case 4: KLAUNCH(KERN<4>, grid, dim3(NW * 64), st, pr, args);           \
    default: KLAUNCH(KERN<8>, grid, dim3(NW * 64), st, pr, args);          \
  }

hipError_t launch_begin(const Dims &d, const FwdPtrs &p, hipStream_t st, LaunchProbe pr) {
  const int n = d.S * (d.C + d.R);
  KLAUNCH(k_begin, dim3(cdiv(n, 256)), dim3(256), st, pr, d.S, d.C, d.R, (const float *)p.prev_c,
          (const float *)p.prev_r, p.cc, p.rr);
}
hipError_t launch_end(const Dims &d, const FwdPtrs &p, hipStream_t st, LaunchProbe pr) {
  const int n = d.S * (d.C + d.R);
  KLAUNCH(k_end, dim3(cdiv(n, 256)), dim3(256), st, pr, d.S, d.C, d.R, d.T, p.prev_c, p.prev_r,
          (const float *)p.cc, (const float *)p.rr);
}

hipError_t launch_gates_step(const Dims &d, const FwdPtrs &p, int t, hipStream_t st, LaunchProbe pr) {
  GatesArgs a;
  a.C = d.C; a.R = d.R; a.S = d.S; a.t = t;
  a.wr = p.wr; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.mm = p.mm; a.rr = p.rr;
  a.vecW = aligned16(p.wr) && d.R % 4 == 0;
  a.vecB = aligned16(p.rr) && d.R % 4 == 0;
  const int nt = pick_nt(d.S);
  const dim3 grid(cdiv(d.C, 4), cdiv(d.S, 16 * nt));
  NT_DISPATCH(k_gates_step, nt, grid, st, pr, a);
}

hipError_t launch_proj_step(const Dims &d, const FwdPtrs &p, int t, float *out, int out_stride,
                            hipStream_t st, LaunchProbe pr) {
  ProjArgs a;
  a.C = d.C; a.R = d.R; a.S = d.S; a.t = t;
  a.wm = p.wm; a.mm = p.mm; a.rr = p.rr; a.out = out; a.out_stride = out_stride;
  a.vecW = aligned16(p.wm) && d.C % 4 == 0;
  a.vecB = aligned16(p.mm) && d.C % 4 == 0;
  a.vecR = aligned16(p.rr) && d.R % 4 == 0;
  a.vecOut = aligned16(out) && out_stride % 4 == 0 && d.R % 4 == 0;
  const int nt = pick_nt(d.S);
  const dim3 grid(cdiv(d.R, 16), cdiv(d.S, 16 * nt));
  NT_DISPATCH(k_proj_step, nt, grid, st, pr, a);
}

int dr_split_k(const Dims &d) {
  // enough (R/16 x KS) workgroups to cover the chip at small S; slices are multiples of KCH
  const int K = 4 * d.C;
  int ks = 8;
  while (ks > 1 && cdiv(K, ks) < NW * KCH / 2) ks >>= 1;
  return ks;
}

hipError_t launch_dr_step(const Dims &d, const BwdPtrs &p, int t, hipStream_t st, LaunchProbe pr) {
  DrArgs a;
  a.C = d.C; a.R = d.R; a.S = d.S; a.t = t;
  a.wrT = p.wrT; a.dgifo = p.dgifo; a.part = p.dr_part;
  const int K = 4 * d.C;
  a.klen = cdiv(cdiv(K, p.ks), KCH) * KCH;
  a.vecW = aligned16(p.wrT) && K % 4 == 0;
  a.vecB = aligned16(p.dgifo) && K % 4 == 0;
  a.vecR = aligned16(p.dr_part) && d.R % 4 == 0;
  const int nt = pick_nt(d.S);
  const dim3 grid(cdiv(d.R, 16), cdiv(d.S, 16 * nt), p.ks);
  NT_DISPATCH(k_dr_step, nt, grid, st, pr, a);
}

hipError_t launch_dm_step(const Dims &d, const BwdPtrs &p, int t, const float *out_diff, int od_stride,
                          hipStream_t st, LaunchProbe pr) {
  DmArgs a;
  a.C = d.C; a.R = d.R; a.S = d.S; a.T = d.T; a.t = t;
  a.wmT = p.wmT; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh;
  a.dgifo = p.dgifo; a.dc = p.dc; a.dr = p.dr;
  a.part = p.dr_part; a.nslab = (t == d.T) ? 0 : p.ks;
  a.out_diff = out_diff; a.od_stride = od_stride;
  a.vecW = aligned16(p.wmT) && d.R % 4 == 0;
  a.vecR = aligned16(p.dr_part) && aligned16(p.dr) && d.R % 4 == 0;
  a.vecOD = aligned16(out_diff) && od_stride % 4 == 0 && d.R % 4 == 0;
  a.vecC = aligned16(p.dgifo) && aligned16(p.dc) && d.C % 4 == 0;
  const int nt = pick_nt(d.S);
  const dim3 grid(cdiv(d.C, 16), cdiv(d.S, 16 * nt));
  NT_DISPATCH(k_dm_step, nt, grid, st, pr, a);
}

hipError_t launch_gemm(bool transA, bool transB, int M, int N, int K, const float *A, int lda,
                       const float *B, int ldb, float beta, float *Cm, int ldc, const float *bias,
                       hipStream_t st, LaunchProbe pr) {
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.beta = beta;
  g.Cm = Cm; g.ldc = ldc; g.bias = bias;
  g.vecA = aligned16(A) && lda % 4 == 0;
  g.vecB = aligned16(B) && ldb % 4 == 0;
  const dim3 grid(cdiv(N, GT), cdiv(M, GT)), block(256);
  if (transA && transB) KLAUNCH((k_gemm<true, true>), grid, block, st, pr, g);
  if (transA && !transB) KLAUNCH((k_gemm<true, false>), grid, block, st, pr, g);
  if (!transA && transB) KLAUNCH((k_gemm<false, true>), grid, block, st, pr, g);
  KLAUNCH((k_gemm<false, false>), grid, block, st, pr, g);
}

hipError_t launch_vec_grads(const Dims &d, const float *dgifo, const float *cc, float beta,
                            float *g_bias, float *g_pi, float *g_pf, float *g_po, hipStream_t st,
                            LaunchProbe pr) {
  KLAUNCH(k_vec_grads, dim3(cdiv(4 * d.C, 64)), dim3(512), st, pr, d.C, d.S, d.T, dgifo, cc, beta,
          g_bias, g_pi, g_pf, g_po);
}

static inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g)); }

hipError_t launch_apply_momentum(float *corr, const float *grad, float mmt, long n, hipStream_t st, LaunchProbe pr) {
  KLAUNCH(k_apply_momentum, dim3(ew_grid(n)), dim3(256), st, pr, corr, grad, mmt, n);
}
hipError_t launch_update(float *param, float *corr, float lr, float clip, long n, hipStream_t st, LaunchProbe pr) {
  KLAUNCH(k_update, dim3(ew_grid(n)), dim3(256), st, pr, param, corr, lr, clip, n);
}
hipError_t launch_transpose(const float *src, int rows, int cols, float *dst, hipStream_t st, LaunchProbe pr) {
  KLAUNCH(k_transpose, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), st, pr, src, rows, cols, dst);
}
hipError_t launch_zero_rows(float *base, int ld, const int *flags_dev, int nrows, int ncols, hipStream_t st) {
  LaunchProbe pr;
  KLAUNCH(k_zero_rows, dim3(nrows), dim3(256), st, pr, base, ld, flags_dev, nrows, ncols);
}

}  // namespace klstm
