// kaldi-lstm_amd/csrc/klstm_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the
// LstmProjectedStreams hot path.  fp32 throughout (Kaldi BaseFloat); every contraction runs on
// the f32-input MFMA  v_mfma_f32_16x16x4_f32  (exact fp32 FMA chain, 157 TF peak), 64-wide waves.
//
// What each kernel replaces in the reference (google/nnet/bd-nnet-lstm-projected-streams.h):
//   k_gates_step : per-step  AddMatMat(r(t-1),W_gifo_r^T) + 2x AddMatDiagVec + Sigmoid x3 + Tanh x2 +
//                  3x AddMatDotMat + ApplyFloor/Ceiling (:275-309, 14 launches); optionally also the
//                  x(t) W_gifo_x^T + bias term (:246,:259) and the state bridge copies (:231,:331)
//   k_proj_step  : per-step  y_r = y_m * W_r_m^T (:312) + the copy into `out` (:328)
//   k_dr_step    : per-step  d_r += DGIFO(t+1) * W_gifo_r (:391) and in_diff(t+1) = DGIFO(t+1) * W_gifo_x
//                  (:457), split-K partial slabs
//   k_dm_step    : per-step  d_m = d_r * W_r_m (:408) + the 15 elementwise launches (:411-440)
//   k_grads      : all gradient accumulations (:468-487) in one grouped launch
//   k_update_repack : Update (:504-512) + refresh of the transposed weight copies
//   k_gemm       : batched x-projection (:246 + :259) when it is not fused into the step kernel
//
// The recurrence is latency-bound at small NumStream (a dependent kernel boundary costs ~1.6 us on
// this chip, a dependent HBM/L2 round trip 0.5-1 us), so every step kernel is written to make
// exactly ONE memory round trip: all weight / activation / epilogue operands are requested before
// the first MFMA issues.
//
// Skinny-GEMM layout of the step kernels ("weights on M, streams on N"):
//   D[16 weight rows][16 streams] += A[row][k] * B[k][stream],  A lane l: row l&15, k-group l>>4;
//   a lane loads 8 consecutive k (two dwordx4) of its weight row / stream row per 32-wide K chunk
//   and feeds them to 8 MFMAs; the 8 waves of a workgroup split K, partial tiles are summed in a
//   fixed order through LDS (deterministic), and the fused LSTM cell math runs on the summed tile
//   in registers.  D lane l holds stream l&15, rows 4*(l>>4)+{0..3}.
#include "klstm_kernels.h"
#include <type_traits>
#include "klstm_math.h"
#include "klstm_persist_dev.h"

#include <hip/hip_ext.h>
#include <stdint.h>

namespace klstm {

#pragma clang fp contract(off)

// ---- operand fetch helpers -----------------------------------------------------------------------
// VEC = true : rows are 16-byte aligned and every contraction length is a multiple of 8.  The load is
//              branch-free: a lane that is out of range reads the (always valid) start of its row and
//              the result is zeroed with selects, so every load of a kernel can be in flight at once.
// VEC = false: generic per-element guarded loads (odd test shapes); correct but latency-serialised.
template <bool VEC>
__device__ __forceinline__ void load8(const float *__restrict__ row, int k, int K, bool ok, float (&v)[8]) {
  if (VEC) {
    const bool in = ok && (k + 8 <= K);
    const float4 *p = reinterpret_cast<const float4 *>(row + (in ? k : 0));
    const float4 a = p[0], b = p[1];
    v[0] = in ? a.x : 0.f; v[1] = in ? a.y : 0.f; v[2] = in ? a.z : 0.f; v[3] = in ? a.w : 0.f;
    v[4] = in ? b.x : 0.f; v[5] = in ? b.y : 0.f; v[6] = in ? b.z : 0.f; v[7] = in ? b.w : 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (ok && k + j < K) ? row[k + j] : 0.f;
  }
}
template <bool VEC>
__device__ __forceinline__ void store8(float *__restrict__ row, int k, int K, const float (&v)[8]) {
  if (VEC) {
    if (k + 8 <= K) {
      *reinterpret_cast<float4 *>(row + k) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4 *>(row + k + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) if (k + j < K) row[k + j] = v[j];
  }
}
// 4 consecutive columns c..c+3 of a row (VEC: C % 4 == 0, 16-byte aligned rows)
template <bool VEC>
__device__ __forceinline__ void load4(const float *__restrict__ p, int c, int C, bool ok, float (&v)[4]) {
  if (VEC) {
    const bool in = ok && (c + 4 <= C);
    const float4 a = *reinterpret_cast<const float4 *>(p + (in ? c : 0));
    v[0] = in ? a.x : 0.f; v[1] = in ? a.y : 0.f; v[2] = in ? a.z : 0.f; v[3] = in ? a.w : 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (ok && c + j < C) ? p[c + j] : 0.f;
  }
}
template <bool VEC>
__device__ __forceinline__ void store4(float *__restrict__ p, int c, int C, const float (&v)[4]) {
  if (VEC) { if (c + 4 <= C) *reinterpret_cast<float4 *>(p + c) = make_float4(v[0], v[1], v[2], v[3]); }
  else {
#pragma unroll
    for (int j = 0; j < 4; j++) if (c + j < C) p[c + j] = v[j];
  }
}

// ---- tile geometry ---------------------------------------------------------------------------------
// SMALL = false: v_mfma_f32_16x16x4_f32, D[16 rows][16 streams]; lane l feeds A row l&15 / B stream l&15
//                of k-group l>>4 and receives rows 4*(l>>4)+{0..3} of stream l&15.
// SMALL = true : NumStream <= 4.  v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4x1 blocks, same FLOP rate):
//                block b = lane>>2 is used as (row group b&3, k-group b>>2), so a lane feeds the SAME A
//                element as above (row l&15, k-group l>>4) but B stream l&3, and receives rows
//                4*((l>>2)&3)+{0..3} of stream l&3 for ITS k-group; the 4 k-groups are added in the
//                fixed-order LDS combine.  No lanes are wasted on padding streams (4x fewer MFMA cycles).
template <bool SMALL> struct Geo {
  static constexpr int STREAMS = SMALL ? 4 : 16;                       // streams per MFMA tile
  __device__ static __forceinline__ int bstream(int lane) { return SMALL ? (lane & 3) : (lane & 15); }
  __device__ static __forceinline__ int q(int lane) { return SMALL ? ((lane >> 2) & 3) : (lane >> 4); }
  __device__ static __forceinline__ bool owner(int lane) { return SMALL ? lane < 16 : true; }
  __device__ static __forceinline__ f32x4 mma(float a, float b, f32x4 c) {
    if (SMALL) return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// Sum the per-wave partial tiles of s-tile `nt` in a fixed order.
template <int NT, bool SMALL>
__device__ __forceinline__ f32x4 reduce_tile(const f32x4 (*red)[NT][64], int nt, int lane) {
  f32x4 v = red[0][nt][lane];          // SMALL: the k-groups were already combined in-wave (VEC_COMBINE)
#pragma unroll
  for (int w = 1; w < NW; w++) v += red[w][nt][lane];
  return v;
}

// K loop shared by the step kernels.  nch chunks of 32 are dealt round-robin to the NW waves; CPW
// chunks per wave are fetched (A and every B tile) before any MFMA so that all loads of a
// super-iteration are in flight together.  afetch(ch, on, av) / bfetch(ch, nt, on, bv) fill 8 floats;
// bpost(ch, nt, on, bv) runs after the MFMAs (side stores of the fetched B operand).
template <int NT, int CPW, bool SMALL, class AF, class BF, class BP>
__device__ __forceinline__ void mma_k_loop(int nch, int wave, f32x4 (&acc)[NT][2], const AF &afetch,
                                           const BF &bfetch, const BP &bpost) {
  for (int base = 0; base < nch; base += CPW * NW) {
    float av[CPW][8];
    float bv[CPW][NT][8];
#pragma unroll
    for (int c = 0; c < CPW; c++) {
      const int ch = base + wave + c * NW;
      const bool on = ch < nch;
      afetch(ch, on, av[c]);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) bfetch(ch, nt, on, bv[c][nt]);
    }
#pragma unroll
    for (int c = 0; c < CPW; c++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[nt][j & 1] = Geo<SMALL>::mma(av[c][j], bv[c][nt][j], acc[nt][j & 1]);
#pragma unroll
    for (int c = 0; c < CPW; c++) {
      const int ch = base + wave + c * NW;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) bpost(ch, nt, ch < nch, bv[c][nt]);
    }
  }
}
struct NoPost { template <class... A> __device__ __forceinline__ void operator()(A &&...) const {} };

// GENERIC kernels (any shape; used when R, I or C is not a multiple of 8): per-element guarded loads straight
// from the natural layouts, 16x16x4 geometry, one K chunk per wave per iteration.
#define GENERIC_GEOMETRY() constexpr int CPW = 1; constexpr bool VEC = false, SMALL = false

#define STEP_PROLOGUE()                                                                          \
  const int lane = threadIdx.x & 63;                                                             \
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                             \
  const int i16 = lane & 15, kg = lane >> 4;                                                     \
  const int bs = Geo<SMALL>::bstream(lane), q = Geo<SMALL>::q(lane);                             \
  constexpr int TS_ = Geo<SMALL>::STREAMS;                                                       \
  (void)i16; (void)kg; (void)bs; (void)q

// ---------------------------------------------------------------------------------------------
// forward step 1/2: gates + cell.  One workgroup = 4 cells x 4 gates (16 weight rows) x NT stream tiles.
// tile row i -> (cell c0 + i/4, gate i%4) so that after the MFMA a lane owns g,i,f,o of ONE
// (cell, stream) pair in its four accumulator registers and the cell math is lane-local.
// ---------------------------------------------------------------------------------------------
struct GatesArgs {
  int C, R, S, I, t;
  const float *wr, *wx, *bias, *pi, *pf, *po;
  float *gifo, *cc, *hh, *mm;
  const float *cprev, *rprev;     // c(t-1) [S x C], r(t-1) [S x R]
  const float *x; int x_stride;   // frame-t input rows [S x I] (FUSEX)
  float *c_mirror, *r_mirror;     // time block 0 of cc / rr at t == 1 (BPTT reads them), else null
  float *c_save;                  // prev_c at t == T, else null
};

template <int NT, bool FUSEX>
__global__ __launch_bounds__(NW * 64) void k_gates_step(GatesArgs a) {
  GENERIC_GEOMETRY();
  STEP_PROLOGUE();
  const int C = a.C, R = a.R, S = a.S, I = a.I, t = a.t;
  const int c0 = blockIdx.x * 4;
  const int sbase = blockIdx.y * TS_ * NT;

  // ---- epilogue operands first: wave nt owns s-tile nt; lane = (cell c0+q, stream bs) ----
  const int e_cell = c0 + q;
  const int e_s = sbase + wave * TS_ + bs;
  const bool e_on = wave < NT && Geo<SMALL>::owner(lane) && e_cell < C && e_s < S;
  const int l_cell = e_on ? e_cell : 0, l_s = e_on ? e_s : 0;          // clamped: loads are unconditional
  const size_t e_row = (size_t)t * S + l_s;
  float pre[4];
#pragma unroll
  for (int g = 0; g < 4; g++) pre[g] = FUSEX ? a.bias[g * C + l_cell] : a.gifo[e_row * 4 * C + g * C + l_cell];
  const float cp = a.cprev[(size_t)l_s * C + l_cell];
  const float wpi = a.pi[l_cell], wpf = a.pf[l_cell], wpo = a.po[l_cell];

  // ---- contraction over [r(t-1) | x(t)] ----
  const int cell_a = c0 + (i16 >> 2), gate_a = i16 & 3;
  const bool row_ok = cell_a < C;
  const size_t wrow_idx = (size_t)gate_a * C + (row_ok ? cell_a : 0);
  const float *wr_row = a.wr + wrow_idx * R;
  const float *wx_row = FUSEX ? a.wx + wrow_idx * I : nullptr;
  const int nchR = (R + KCH - 1) / KCH;
  const int nch = nchR + (FUSEX ? (I + KCH - 1) / KCH : 0);
  const bool mirror_r = a.r_mirror != nullptr && blockIdx.x == 0 && (SMALL ? q == 0 : true);

  f32x4 acc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) { acc[nt][0] = (f32x4){0, 0, 0, 0}; acc[nt][1] = (f32x4){0, 0, 0, 0}; }

  auto afetch = [&](int ch, bool on, float (&v)[8]) {
    if (!FUSEX || ch < nchR) load8<VEC>(wr_row, ch * KCH + kg * 8, R, on && row_ok, v);
    else load8<VEC>(wx_row, (ch - nchR) * KCH + kg * 8, I, on && row_ok, v);
  };
  auto bfetch = [&](int ch, int nt, bool on, float (&v)[8]) {
    const int s = sbase + nt * TS_ + bs;
    const bool ok = on && s < S;
    if (!FUSEX || ch < nchR) load8<VEC>(a.rprev + (size_t)(ok ? s : 0) * R, ch * KCH + kg * 8, R, ok, v);
    else load8<VEC>(a.x + (size_t)(ok ? s : 0) * a.x_stride, (ch - nchR) * KCH + kg * 8, I, ok, v);
  };
  auto bpost = [&](int ch, int nt, bool on, const float (&v)[8]) {      // :231 (r columns of block 0)
    const int s = sbase + nt * TS_ + bs;
    if (mirror_r && on && s < S && ch < nchR) store8<VEC>(a.r_mirror + (size_t)s * R, ch * KCH + kg * 8, R, v);
  };
  mma_k_loop<NT, CPW, SMALL>(nch, wave, acc, afetch, bfetch, bpost);

  __shared__ f32x4 red[NW][NT][64];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) red[wave][nt][lane] = acc[nt][0] + acc[nt][1];
  __syncthreads();

  if (e_on) {
    const f32x4 v = reduce_tile<NT, SMALL>(red, wave, lane);
    float ag = v.x + pre[0];
    float ai = v.y + pre[1];
    float af = v.z + pre[2];
    float ao = v.w + pre[3];
    ai += wpi * cp;                                // :278
    af += wpf * cp;                                // :281
    const float gi = k_sigmoid(ai), gf = k_sigmoid(af), gg = k_tanh(ag);   // :284-288
    float c = gg * gi;                             // :291
    c = c + cp * gf;                               // :294
    c = c < -50.f ? -50.f : c;                     // :296
    c = c > 50.f ? 50.f : c;                       // :297
    const float h = k_tanh(c);                     // :300
    ao += wpo * c;                                 // :303
    const float go = k_sigmoid(ao);                // :306
    const float m = h * go;                        // :309
    float *gp = a.gifo + e_row * 4 * C + e_cell;
    gp[0] = gg; gp[C] = gi; gp[2 * C] = gf; gp[3 * C] = go;
    a.cc[e_row * C + e_cell] = c;
    a.hh[e_row * C + e_cell] = h;
    a.mm[e_row * C + e_cell] = m;
    if (a.c_mirror) a.c_mirror[(size_t)e_s * C + e_cell] = cp;   // :231 (c columns)
    if (a.c_save) a.c_save[(size_t)e_s * C + e_cell] = c;         // :331 (c columns)
  }
}

// ---------------------------------------------------------------------------------------------
// forward step 2/2: recurrent projection r(t) = m(t) * W_r_m^T (:312), also written to `out` (:328)
// and, at t == T, to the carried state (:331).  One workgroup = 16 projection rows x NT stream tiles;
// lane owns 4 consecutive r columns.
// ---------------------------------------------------------------------------------------------
struct ProjArgs {
  int C, R, S, t;
  const float *wm, *mm;
  float *rr, *out, *r_save;
  int out_stride;
  int vecOut;
};

template <int NT>
__global__ __launch_bounds__(NW * 64) void k_proj_step(ProjArgs a) {
  GENERIC_GEOMETRY();
  STEP_PROLOGUE();
  const int C = a.C, R = a.R, S = a.S, t = a.t;
  const int n0 = blockIdx.x * 16;
  const int sbase = blockIdx.y * TS_ * NT;
  const bool row_ok = n0 + i16 < R;
  const float *wrow = a.wm + (size_t)(row_ok ? n0 + i16 : 0) * C;
  const float *mrow = a.mm + (size_t)t * S * C;

  f32x4 acc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) { acc[nt][0] = (f32x4){0, 0, 0, 0}; acc[nt][1] = (f32x4){0, 0, 0, 0}; }
  auto afetch = [&](int ch, bool on, float (&v)[8]) { load8<VEC>(wrow, ch * KCH + kg * 8, C, on && row_ok, v); };
  auto bfetch = [&](int ch, int nt, bool on, float (&v)[8]) {
    const int s = sbase + nt * TS_ + bs;
    const bool ok = on && s < S;
    load8<VEC>(mrow + (size_t)(ok ? s : 0) * C, ch * KCH + kg * 8, C, ok, v);
  };
  mma_k_loop<NT, CPW, SMALL>((C + KCH - 1) / KCH, wave, acc, afetch, bfetch, NoPost());

  __shared__ f32x4 red[NW][NT][64];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) red[wave][nt][lane] = acc[nt][0] + acc[nt][1];
  __syncthreads();

  if (wave < NT && Geo<SMALL>::owner(lane)) {
    const f32x4 v = reduce_tile<NT, SMALL>(red, wave, lane);
    const int s = sbase + wave * TS_ + bs;
    const int n = n0 + 4 * q;
    if (s < S && n < R) {
      const float e[4] = {v.x, v.y, v.z, v.w};
      store4<VEC>(a.rr + ((size_t)t * S + s) * R, n, R, e);
      float *op = a.out + (size_t)((t - 1) * S + s) * a.out_stride;
      if (a.vecOut) store4<true>(op, n, R, e); else store4<false>(op, n, R, e);
      if (a.r_save) store4<VEC>(a.r_save + (size_t)s * R, n, R, e);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward step 1/2: partial products of DGIFO(t+1) [S x 4C] with W_gifo_r (-> d_r(t), :391) and,
// optionally, W_gifo_x (-> in_diff of frame t+1, :457) over one K slice of the 4C gate rows.
// grid (R tiles + I tiles, stream groups, KS); slab ks holds the partial sum of its slice, the
// consumer (k_dm_step) adds the slabs in fixed order.
// ---------------------------------------------------------------------------------------------
struct DrArgs {
  int C, R, I, S, t;
  const float *wrT, *wxT;   // [R x 4C], [I x 4C]
  const float *dgifo;
  float *part;              // [KS][S][R]
  float *xpart;             // [KS][S][x_ld]  (or in_diff rows of frame 1 when t == 0, KS == 1)
  int x_ld;
  int ntr;                  // number of 16-row tiles over R in grid.x (0 at t == 0)
  int klen;                 // K slice length (multiple of KCH)
  int vecX;
};

template <int NT>
__global__ __launch_bounds__(NW * 64) void k_dr_step(DrArgs a) {
  GENERIC_GEOMETRY();
  STEP_PROLOGUE();
  const int R = a.R, I = a.I, S = a.S, K = 4 * a.C;
  const bool is_x = (int)blockIdx.x >= a.ntr;
  const int n0 = (is_x ? (int)blockIdx.x - a.ntr : (int)blockIdx.x) * 16;
  const int N = is_x ? I : R;
  const int sbase = blockIdx.y * TS_ * NT;
  const int ks = blockIdx.z;
  const int kbeg = ks * a.klen;
  const int kend = min(K, kbeg + a.klen);
  const bool row_ok = n0 + i16 < N;
  const float *wrow = (is_x ? a.wxT : a.wrT) + (size_t)(row_ok ? n0 + i16 : 0) * K;
  const float *drow = a.dgifo + (size_t)(a.t + 1) * S * K;

  f32x4 acc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) { acc[nt][0] = (f32x4){0, 0, 0, 0}; acc[nt][1] = (f32x4){0, 0, 0, 0}; }
  auto afetch = [&](int ch, bool on, float (&v)[8]) { load8<VEC>(wrow, kbeg + ch * KCH + kg * 8, kend, on && row_ok, v); };
  auto bfetch = [&](int ch, int nt, bool on, float (&v)[8]) {
    const int s = sbase + nt * TS_ + bs;
    const bool ok = on && s < S;
    load8<VEC>(drow + (size_t)(ok ? s : 0) * K, kbeg + ch * KCH + kg * 8, kend, ok, v);
  };
  mma_k_loop<NT, CPW, SMALL>(kend > kbeg ? (kend - kbeg + KCH - 1) / KCH : 0, wave, acc, afetch, bfetch, NoPost());

  __shared__ f32x4 red[NW][NT][64];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) red[wave][nt][lane] = acc[nt][0] + acc[nt][1];
  __syncthreads();

  if (wave < NT && Geo<SMALL>::owner(lane)) {
    const f32x4 v = reduce_tile<NT, SMALL>(red, wave, lane);
    const int s = sbase + wave * TS_ + bs;
    const int n = n0 + 4 * q;
    if (s < S && n < N) {
      const float e[4] = {v.x, v.y, v.z, v.w};
      if (is_x) {
        float *xp = a.xpart + ((size_t)ks * S + s) * a.x_ld;
        if (a.vecX) store4<true>(xp, n, I, e); else store4<false>(xp, n, I, e);
      } else {
        store4<VEC>(a.part + ((size_t)ks * S + s) * R, n, R, e);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward step 2/2: d_r(t) = out_diff(t) + sum of slabs; d_m = d_r * W_r_m (:408) and the whole
// elementwise BPTT cell math (:411-440).  One workgroup = 16 cells x NT stream tiles; lane owns 4
// consecutive cells of one stream.  Workgroup x==0 also materialises d_r(t) (needed by the
// W_r_m gradient, :486); the last workgroup also reduces the in_diff slabs of frame t+1.
// ---------------------------------------------------------------------------------------------
struct DmArgs {
  int C, R, I, S, T, t;
  const float *wmT;       // [C x R]
  const float *pi, *pf, *po;
  const float *gifo, *cc, *hh;
  float *dgifo, *dc, *dr;
  const float *part;      // [KS][S][R]
  int nslab;              // 0 at t == T (the t+1 block is all zero, :351)
  const float *out_diff;
  int od_stride;
  const float *xpart;     // [KS][S][I] in_diff slabs of frame t+1 (null: nothing to reduce)
  float *in_diff;         // rows of frame t+1
  int id_stride;
};

// Side job of the d_m kernels: in_diff(t+1) = sum of its split-K slabs (:457), the (streams x I) elements of a
// stream tile spread evenly over the gx row-tile workgroups of that tile (one workgroup alone takes 16 dependent
// rounds at I = 512, the stacked-layer shape).
__device__ __forceinline__ void reduce_x_slabs(const DmArgs &a, int s_lo, int s_hi, int bx, int gx) {
  if (!a.xpart) return;
  const int I = a.I, S = a.S;
  const int per = ((s_hi - s_lo) * I + gx - 1) / gx;
  const int lo = s_lo * I + bx * per, hi = min(s_hi * I, lo + per);
  for (int idx = lo + (int)threadIdx.x; idx < hi; idx += NW * 64) {
    const int s = idx / I, n = idx - s * I;
    float p[KSMAX];
#pragma unroll
    for (int ks = 0; ks < KSMAX; ks++) p[ks] = a.xpart[((size_t)(ks < a.nslab ? ks : 0) * S + s) * I + n];
    float sum = p[0];
#pragma unroll
    for (int ks = 1; ks < KSMAX; ks++) sum += ks < a.nslab ? p[ks] : 0.f;
    a.in_diff[(size_t)s * a.id_stride + n] = sum;
  }
}

template <int NT>
__global__ __launch_bounds__(NW * 64) void k_dm_step(DmArgs a) {
  GENERIC_GEOMETRY();
  STEP_PROLOGUE();
  const int C = a.C, R = a.R, S = a.S, t = a.t;
  const int c0 = blockIdx.x * 16;
  const int sbase = blockIdx.y * TS_ * NT;
  const bool last = (t == a.T);

  reduce_x_slabs(a, sbase, min(S, sbase + TS_ * NT), blockIdx.x, gridDim.x);

  // ---- epilogue operands first: wave nt owns s-tile nt; lane = (stream, cells cb..cb+3) ----
  const int e_s = sbase + wave * TS_ + bs;
  const int cb = c0 + 4 * q;
  const bool e_on = wave < NT && Geo<SMALL>::owner(lane) && e_s < S && cb < C;
  const size_t row = (size_t)t * S + (e_on ? e_s : 0), rown = row + S, rowp = row - S;
  float yg[4], yi[4], yf[4], yo[4], yh[4], cpv[4], dcn[4], fn[4], din[4], dfn[4], wpi[4], wpf[4], wpo[4];
  {
    const float *yp = a.gifo + row * 4 * C;
    load4<VEC>(yp, cb, C, e_on, yg);
    load4<VEC>(yp + C, cb, C, e_on, yi);
    load4<VEC>(yp + 2 * C, cb, C, e_on, yf);
    load4<VEC>(yp + 3 * C, cb, C, e_on, yo);
    load4<VEC>(a.hh + row * C, cb, C, e_on, yh);
    load4<VEC>(a.cc + rowp * C, cb, C, e_on, cpv);
    const bool n_on = e_on && !last;
    const size_t rn = last ? row : rown;                 // clamped: block T+1 is never dereferenced
    load4<VEC>(a.dc + rn * C, cb, C, n_on, dcn);
    load4<VEC>(a.gifo + rn * 4 * C + 2 * C, cb, C, n_on, fn);
    load4<VEC>(a.dgifo + rn * 4 * C + C, cb, C, n_on, din);
    load4<VEC>(a.dgifo + rn * 4 * C + 2 * C, cb, C, n_on, dfn);
    load4<VEC>(a.pi, cb, C, e_on, wpi);
    load4<VEC>(a.pf, cb, C, e_on, wpf);
    load4<VEC>(a.po, cb, C, e_on, wpo);
  }

  const bool row_ok = c0 + i16 < C;
  const float *wrow = a.wmT + (size_t)(row_ok ? c0 + i16 : 0) * R;
  const bool write_dr = blockIdx.x == 0 && (SMALL ? q == 0 : true);

  f32x4 acc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) { acc[nt][0] = (f32x4){0, 0, 0, 0}; acc[nt][1] = (f32x4){0, 0, 0, 0}; }
  auto afetch = [&](int ch, bool on, float (&v)[8]) { load8<VEC>(wrow, ch * KCH + kg * 8, R, on && row_ok, v); };
  auto bfetch = [&](int ch, int nt, bool on, float (&v)[8]) {
    const int s = sbase + nt * TS_ + bs;
    const bool ok = on && s < S;
    const int k = ch * KCH + kg * 8;
    const float *odp = a.out_diff + (size_t)((t - 1) * S + (ok ? s : 0)) * a.od_stride;          // :367
    load8<VEC>(odp, k, R, ok, v);
    float p[KSMAX][8];
#pragma unroll
    for (int ks = 0; ks < KSMAX; ks++)
      load8<VEC>(a.part + ((size_t)(ks < a.nslab ? ks : 0) * S + (ok ? s : 0)) * R, k, R, ok && ks < a.nslab, p[ks]);
#pragma unroll
    for (int ks = 0; ks < KSMAX; ks++)
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] += p[ks][j];                                               // :391
  };
  auto bpost = [&](int ch, int nt, bool on, const float (&v)[8]) {
    const int s = sbase + nt * TS_ + bs;
    if (write_dr && on && s < S) store8<VEC>(a.dr + ((size_t)t * S + s) * R, ch * KCH + kg * 8, R, v);
  };
  mma_k_loop<NT, CPW, SMALL>((R + KCH - 1) / KCH, wave, acc, afetch, bfetch, bpost);

  __shared__ f32x4 red[NW][NT][64];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) red[wave][nt][lane] = acc[nt][0] + acc[nt][1];
  __syncthreads();

  if (e_on) {
    const f32x4 v = reduce_tile<NT, SMALL>(red, wave, lane);
    const float dm[4] = {v.x, v.y, v.z, v.w};
    float og[4], oi[4], of[4], oo[4], oc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float d_h = k_diff_tanh(dm[j] * yo[j], yh[j]);       // :411-412
      const float d_o = k_diff_sigmoid(dm[j] * yh[j], yo[j]);    // :415-416
      float d_c = d_h;                                           // :424
      d_c = d_c + dcn[j] * fn[j];                                // :425
      d_c = d_c + wpi[j] * din[j];                               // :426
      d_c = d_c + wpf[j] * dfn[j];                               // :427
      d_c = d_c + wpo[j] * d_o;                                  // :428
      of[j] = k_diff_sigmoid(d_c * cpv[j], yf[j]);               // :431-432
      oi[j] = k_diff_sigmoid(d_c * yg[j], yi[j]);                // :435-436
      og[j] = k_diff_tanh(d_c * yi[j], yg[j]);                   // :439-440
      oo[j] = d_o;
      oc[j] = d_c;
    }
    float *dp = a.dgifo + row * 4 * C;
    store4<VEC>(dp, cb, C, og);
    store4<VEC>(dp + C, cb, C, oi);
    store4<VEC>(dp + 2 * C, cb, C, of);
    store4<VEC>(dp + 3 * C, cb, C, oo);
    store4<VEC>(a.dc + row * C, cb, C, oc);
  }
}

// =============================================================================================
// VECTOR PATH (aligned shapes: R, I, C multiples of 8).  Two rules measured on MI355X shape it
// (tools/act_anatomy.hip, tools/step_anatomy.hip): a wave-load whose lanes hit 16+ scattered rows
// costs ~64 L1 lookups (+0.7 us per kernel for an 8 KB operand), while the same bytes fetched
// lane-contiguous cost +0.1 us.  Therefore
//   * weights are read from PACKED copies laid out in MFMA A-operand order
//       pk[tile][chunk][h][lane][4]:  row = rowmap(tile, lane&15), k = chunk*32 + (lane>>4)*8 + h*4 + e
//     (zero padded), written once per Update by k_pack -> every weight load is a contiguous 1 KB;
//   * activations are fetched lane-contiguous from their natural [stream][k] layout, staged in LDS
//     (padded rows, conflict-free ds_read_b128) and read from there in B-operand order.
// =============================================================================================
template <int CPW, bool SMALL> struct VGeo {
  static constexpr int SUPER = CPW * NW;                       // chunks per super-iteration
  static constexpr int LDB = SUPER * KCH + (SMALL ? 16 : 4);   // LDS row stride (floats): b128 reads conflict-free
  static constexpr int LDBH = SUPER * KCH + (SMALL ? 32 : 8);  // same padding in bytes for the bf16 slab (halves)
};

// One contraction: acc[nt] += A(tile rows, chunks [0,nch)) * B(streams, same chunks).
//   apk   : packed weights of this tile, chunk c at apk + c*128 (float4 units)
//   bload(s, k, on) -> float4 of B[s][k..k+3] in natural layout (zeros outside / when !on), k relative to chunk 0;
//                      must be branch-free (clamped addresses + selects) so that all loads of a slab overlap
//   bside(s, k, v): optional side store of the staged natural-layout value (mirrors)
//   BF    : bf16 operands -- chunk c of the packed weights at apk + c*64 (one 16-byte vector of 8 bf16 per lane),
//           B staged as bf16; one 16x16x32 (or two 4x4x4_16b) MFMAs per chunk instead of eight f32 ones
struct NoEpi { __device__ __forceinline__ void operator()() const {} };
//   epi() : issues the caller's epilogue-operand loads; called once, right AFTER the slab and weight loads of the first
//           super-iteration (loads return in order: what is needed last is requested last)
template <int NT, int CPW, bool SMALL, bool BF, class BL, class BS, class EP = NoEpi>
__device__ __forceinline__ void vec_contract(const float4 *__restrict__ apk, int nch, int rows, float *ldsB,
                                             int lane, int wave, f32x4 (&acc)[NT][2], const BL &bload, const BS &bside,
                                             const EP &epi = EP()) {
  (void)rows;
  constexpr int SUPER = VGeo<CPW, SMALL>::SUPER, LDB = VGeo<CPW, SMALL>::LDB, LDBH = VGeo<CPW, SMALL>::LDBH;
  constexpr int TS_ = Geo<SMALL>::STREAMS;
  constexpr int AU = BF ? 64 : 128;                            // float4 units per packed chunk
  unsigned short *ldsH = reinterpret_cast<unsigned short *>(ldsB);
  const int bs = Geo<SMALL>::bstream(lane), kg = lane >> 4;
  // The first super-iteration is PEELED out of the loop (it is the only one whenever K <= SUPER*32, i.e. in every
  // latency-critical shape): at a loop header the compiler drains all outstanding loads (s_waitcnt vmcnt(0)), which made
  // the epilogue operands the callers request up front a full memory round trip in FRONT of the weight fetch.
  auto iter = [&](int base) {
    const int nc = min(SUPER, nch - base);
    // (1) this wave's weight chunks: an even contiguous share [c0, c0 + per) of the super-iteration
    const int per = (nc + NW - 1) / NW;
    const int c0 = wave * per;
    // (2a) the B slab is requested FIRST: loads return in order, and the slab (LDS store, barrier) is needed before the
    //      weights (first MFMA) -- behind 8 weight loads its wait would drain them too
    constexpr int ROWS = NT * TS_, F4ROW = SUPER * 8;
    constexpr int U = (ROWS * F4ROW + NW * 64 - 1) / (NW * 64);
    float4 sv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int idx = threadIdx.x + u * NW * 64;
      const int sl = idx / F4ROW, k = (idx % F4ROW) * 4;
      sv[u] = bload(sl < ROWS ? sl : 0, base * KCH + k, sl < ROWS && k < nc * KCH);
    }
    float4 a0[CPW], a1[CPW];
#pragma unroll
    for (int c = 0; c < CPW; c++) {
      const int cl = min(c0 + c, nc - 1);                      // clamped: always a valid chunk, unused if off
      const float4 *ap = apk + (size_t)(base + cl) * AU + lane;
      a0[c] = ap[0]; a1[c] = BF ? a0[c] : ap[64];
    }
    if (base == 0) epi();
    // (2b) stage B[rows][nc*32] lane-contiguous into LDS: every load of the slab is issued before the first store
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int idx = threadIdx.x + u * NW * 64;
      const int sl = idx / F4ROW, k = (idx % F4ROW) * 4;
      if (sl < ROWS && k < nc * KCH) {
        if constexpr (BF) *reinterpret_cast<uint2 *>(ldsH + sl * LDBH + k) = pack_bf16x4(sv[u]);
        else *reinterpret_cast<float4 *>(ldsB + sl * LDB + k) = sv[u];
        bside(sl, base * KCH + k, sv[u]);
      }
    }
    __syncthreads();
    // (3) MFMAs
#pragma unroll
    for (int c = 0; c < CPW; c++) {
      if (c < per && c0 + c < nc) {
        if constexpr (BF) {
#pragma unroll
          for (int nt = 0; nt < NT; nt++) {
            const float4 braw = *reinterpret_cast<const float4 *>(ldsH + (nt * TS_ + bs) * LDBH + (c0 + c) * KCH + kg * 8);
            if constexpr (SMALL) {
              const float2 alo = make_float2(a0[c].x, a0[c].y), ahi = make_float2(a0[c].z, a0[c].w);
              const float2 blo = make_float2(braw.x, braw.y), bhi = make_float2(braw.z, braw.w);
              acc[nt][0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, alo), __builtin_bit_cast(s16x4, blo), acc[nt][0], 0, 0, 0);
              acc[nt][1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, ahi), __builtin_bit_cast(s16x4, bhi), acc[nt][1], 0, 0, 0);
            } else {
              acc[nt][c & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0[c]), __builtin_bit_cast(bf16x8, braw), acc[nt][c & 1], 0, 0, 0);
            }
          }
          continue;
        }
        const float av[8] = {a0[c].x, a0[c].y, a0[c].z, a0[c].w, a1[c].x, a1[c].y, a1[c].z, a1[c].w};
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const float *bp = ldsB + (nt * TS_ + bs) * LDB + (c0 + c) * KCH + kg * 8;
          const float4 b0 = *reinterpret_cast<const float4 *>(bp), b1 = *reinterpret_cast<const float4 *>(bp + 4);
          const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int j = 0; j < 8; j++) acc[nt][j & 1] = Geo<SMALL>::mma(av[j], bv[j], acc[nt][j & 1]);
        }
      }
    }
    if (base + SUPER < nch) __syncthreads();
  };
  if (nch > 0) iter(0); else epi();
  for (int base = SUPER; base < nch; base += SUPER) iter(base);
}
// Zero a fetched vector where the lane's element is padding.  Written as a bitwise AND with an OPAQUE lane mask: as a
// select the compiler sinks the load into a conditional block and drains every outstanding load (s_waitcnt vmcnt(0)) at
// its join -- the activation slab of the gates kernel was two serialized memory round trips in front of the weight fetch.
__device__ __forceinline__ float4 keep_if(const float4 &v, bool c) {
  unsigned m = c ? 0xffffffffu : 0u;
  asm volatile("" : "+v"(m));
  return make_float4(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.x) & m),
                     __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.y) & m),
                     __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.z) & m),
                     __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.w) & m));
}
struct NoSide { __device__ __forceinline__ void operator()(int, int, const float4 &) const {} };
__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

#define VEC_PROLOGUE()                                                                           \
  const int lane = threadIdx.x & 63;                                                             \
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                             \
  const int bs = Geo<SMALL>::bstream(lane), q = Geo<SMALL>::q(lane);                             \
  constexpr int TS_ = Geo<SMALL>::STREAMS;                                                       \
  __shared__ __attribute__((aligned(16))) float ldsB[NT * TS_ * VGeo<CPW, SMALL>::LDB];          \
  __shared__ f32x4 red[NW][NT][64];                                                              \
  f32x4 acc[NT][2];                                                                              \
  _Pragma("unroll") for (int nt = 0; nt < NT; nt++) { acc[nt][0] = (f32x4){0, 0, 0, 0}; acc[nt][1] = (f32x4){0, 0, 0, 0}; } \
  (void)bs; (void)q

// 4x4x1_16b geometry: the four k-groups of a (row, stream) pair sit 16 lanes apart -> xor butterfly first (a+b == b+a
// bitwise, so every lane ends with the same sum), then 8 per-wave partials instead of 32 go through LDS
#define VEC_COMBINE()                                                                            \
  _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {                                            \
    f32x4 _v = acc[nt][0] + acc[nt][1];                                                          \
    if (SMALL) {                                                                                 \
      _Pragma("unroll") for (int _m = 16; _m < 64; _m <<= 1) {                                   \
        _v.x += __shfl_xor(_v.x, _m); _v.y += __shfl_xor(_v.y, _m); _v.z += __shfl_xor(_v.z, _m); _v.w += __shfl_xor(_v.w, _m); \
      }                                                                                          \
    }                                                                                            \
    red[wave][nt][lane] = _v;                                                                    \
  }                                                                                              \
  __syncthreads()

struct GatesVArgs {
  int gx;                // real number of row-tile groups (many-stream kernels pad grid.x to a multiple of 8)
  GatesArgs g;
  const float4 *wpk;     // packed [W_gifo_r | W_gifo_x] : [C/4 tiles][nchR + nchX chunks][2][64]
  int nch_total;         // chunks per tile in wpk (nchR + nchX)
};

template <int NT, int CPW, bool SMALL, bool FUSEX, bool BF>
__global__ __launch_bounds__(NW * 64) void k_gates_v(GatesVArgs va) {
  const GatesArgs &a = va.g;
  VEC_PROLOGUE();
  const int C = a.C, R = a.R, S = a.S, I = a.I, t = a.t;
  const int c0 = blockIdx.x * 4;
  const int sbase = blockIdx.y * TS_ * NT;

  // ---- epilogue operands first: wave nt owns s-tile nt; lane = (cell c0+q, stream bs) ----
  const int e_cell = c0 + q;
  const int e_s = sbase + wave * TS_ + bs;
  const bool e_on = wave < NT && Geo<SMALL>::owner(lane) && e_cell < C && e_s < S;
  const int l_cell = e_on ? e_cell : 0, l_s = e_on ? e_s : 0;
  const size_t e_row = (size_t)t * S + l_s;
  // (8 scalar loads: measured 0.1 us faster requested here, under the scalar prologue, than after the weight loads)
  float pre[4];
#pragma unroll
  for (int g = 0; g < 4; g++) pre[g] = FUSEX ? a.bias[g * C + l_cell] : a.gifo[e_row * 4 * C + g * C + l_cell];
  const float cp = a.cprev[(size_t)l_s * C + l_cell];
  const float wpi = a.pi[l_cell], wpf = a.pf[l_cell], wpo = a.po[l_cell];

  const int nchR = R / KCH + (R % KCH != 0);
  const int Rp = nchR * KCH;
  const int nch = nchR + (FUSEX ? (I + KCH - 1) / KCH : 0);
  const bool mirror_r = a.r_mirror != nullptr && blockIdx.x == 0;
  auto bload = [&](int sl, int k, bool on) -> float4 {    // B = [ r(t-1) | pad | x(t) | pad ], branch-free
    const int s = min(sbase + sl, S - 1);
    const bool inr = k < R, inx = FUSEX && k >= Rp && k - Rp < I;
    const float *p = inr ? a.rprev + (size_t)s * R + k : inx ? a.x + (size_t)s * a.x_stride + (k - Rp) : a.rprev;
    const float4 v = ldg4(p);
    return keep_if(v, on && sbase + sl < S && (inr || inx));
  };
  auto bside = [&](int sl, int k, const float4 &v) {      // :231 (r columns of time block 0)
    const int s = sbase + sl;
    if (mirror_r && s < S && k < R) *reinterpret_cast<float4 *>(a.r_mirror + (size_t)s * R + k) = v;
  };
  vec_contract<NT, CPW, SMALL, BF>(va.wpk + (size_t)blockIdx.x * va.nch_total * (BF ? 64 : 128), nch, NT * TS_, ldsB, lane, wave, acc,
                               bload, bside);
  VEC_COMBINE();

  if (e_on) {
    const f32x4 v = reduce_tile<NT, SMALL>(red, wave, lane);
    float ag = v.x + pre[0];
    float ai = v.y + pre[1];
    float af = v.z + pre[2];
    float ao = v.w + pre[3];
    ai += wpi * cp;                                // :278
    af += wpf * cp;                                // :281
    const float gi = k_sigmoid(ai), gf = k_sigmoid(af), gg = k_tanh(ag);   // :284-288
    float c = gg * gi;                             // :291
    c = c + cp * gf;                               // :294
    c = c < -50.f ? -50.f : c;                     // :296
    c = c > 50.f ? 50.f : c;                       // :297
    const float h = k_tanh(c);                     // :300
    ao += wpo * c;                                 // :303
    const float go = k_sigmoid(ao);                // :306
    const float m = h * go;                        // :309
    float *gp = a.gifo + e_row * 4 * C + e_cell;
    gp[0] = gg; gp[C] = gi; gp[2 * C] = gf; gp[3 * C] = go;
    a.cc[e_row * C + e_cell] = c;
    a.hh[e_row * C + e_cell] = h;
    a.mm[e_row * C + e_cell] = m;
    if (a.c_mirror) a.c_mirror[(size_t)e_s * C + e_cell] = cp;   // :231 (c columns)
    if (a.c_save) a.c_save[(size_t)e_s * C + e_cell] = c;         // :331 (c columns)
  }
}

struct ProjVArgs { int gx; ProjArgs g; const float4 *wpk; };   // packed W_r_m: [R/16 tiles][C/32 chunks][2][64]

template <int NT, int CPW, bool SMALL, bool BF>
__global__ __launch_bounds__(NW * 64) void k_proj_v(ProjVArgs va) {
  const ProjArgs &a = va.g;
  VEC_PROLOGUE();
  const int C = a.C, R = a.R, S = a.S, t = a.t;
  const int n0 = blockIdx.x * 16;
  const int sbase = blockIdx.y * TS_ * NT;
  const int nch = (C + KCH - 1) / KCH;
  const float *mrow = a.mm + (size_t)t * S * C;
  auto bload = [&](int sl, int k, bool on) -> float4 {
    const float4 v = ldg4(mrow + (size_t)min(sbase + sl, S - 1) * C + min(k, C - 4));
    return keep_if(v, on && sbase + sl < S && k < C);
  };
  vec_contract<NT, CPW, SMALL, BF>(va.wpk + (size_t)blockIdx.x * nch * (BF ? 64 : 128), nch, NT * TS_, ldsB, lane, wave, acc, bload, NoSide());
  VEC_COMBINE();
  if (wave < NT && Geo<SMALL>::owner(lane)) {
    const f32x4 v = reduce_tile<NT, SMALL>(red, wave, lane);
    const int s = sbase + wave * TS_ + bs;
    const int n = n0 + 4 * q;
    if (s < S && n < R) {
      const float e[4] = {v.x, v.y, v.z, v.w};
      store4<true>(a.rr + ((size_t)t * S + s) * R, n, R, e);
      float *op = a.out + (size_t)((t - 1) * S + s) * a.out_stride;
      if (a.vecOut) store4<true>(op, n, R, e); else store4<false>(op, n, R, e);
      if (a.r_save) store4<true>(a.r_save + (size_t)s * R, n, R, e);
    }
  }
}

struct DrVArgs { int gx; DrArgs g; const float4 *wpk; int nch_total; };   // packed [W_gifo_r^T ; W_gifo_x^T]: [(R+I)/16 tiles][4C/32][2][64]

template <int NT, int CPW, bool SMALL, bool BF>
__global__ __launch_bounds__(NW * 64) void k_dr_v(DrVArgs va) {
  const DrArgs &a = va.g;
  VEC_PROLOGUE();
  const int R = a.R, I = a.I, S = a.S, K = 4 * a.C;
  const bool is_x = (int)blockIdx.x >= a.ntr;
  const int ntR = (R + 15) / 16;                               // x tiles follow the R tiles in the packed array
  const int tile = is_x ? ntR + ((int)blockIdx.x - a.ntr) : (int)blockIdx.x;
  const int n0 = (is_x ? (int)blockIdx.x - a.ntr : (int)blockIdx.x) * 16;
  const int N = is_x ? I : R;
  const int sbase = blockIdx.y * TS_ * NT;
  const int ks = blockIdx.z;
  const int kbeg = ks * a.klen;
  const int kend = min(K, kbeg + a.klen);
  const float *drow = a.dgifo + (size_t)(a.t + 1) * S * K;
  auto bload = [&](int sl, int k, bool on) -> float4 {
    const float4 v = ldg4(drow + (size_t)min(sbase + sl, S - 1) * K + min(kbeg + k, K - 4));
    return keep_if(v, on && sbase + sl < S && kbeg + k < kend);
  };
  vec_contract<NT, CPW, SMALL, BF>(va.wpk + ((size_t)tile * va.nch_total + kbeg / KCH) * (BF ? 64 : 128),
                               kend > kbeg ? (kend - kbeg + KCH - 1) / KCH : 0, NT * TS_, ldsB, lane, wave, acc, bload, NoSide());
  VEC_COMBINE();
  if (wave < NT && Geo<SMALL>::owner(lane)) {
    const f32x4 v = reduce_tile<NT, SMALL>(red, wave, lane);
    const int s = sbase + wave * TS_ + bs;
    const int n = n0 + 4 * q;
    if (s < S && n < N) {
      const float e[4] = {v.x, v.y, v.z, v.w};
      if (is_x) {
        float *xp = a.xpart + ((size_t)ks * S + s) * a.x_ld;
        if (a.vecX) store4<true>(xp, n, I, e); else store4<false>(xp, n, I, e);
      } else {
        store4<true>(a.part + ((size_t)ks * S + s) * R, n, R, e);
      }
    }
  }
}

struct DmVArgs { int gx; DmArgs g; const float4 *wpk; };   // packed W_r_m^T: [C/16 tiles][R/32 chunks][2][64]

template <int NT, int CPW, bool SMALL, bool BF>
__global__ __launch_bounds__(NW * 64) void k_dm_v(DmVArgs va) {
  const DmArgs &a = va.g;
  VEC_PROLOGUE();
  const int C = a.C, R = a.R, S = a.S, t = a.t;
  const int c0 = blockIdx.x * 16;
  const int sbase = blockIdx.y * TS_ * NT;
  const bool last = (t == a.T);

  reduce_x_slabs(a, sbase, min(S, sbase + TS_ * NT), blockIdx.x, gridDim.x);

  // ---- epilogue operands (requested from inside the contraction, after its own loads): wave nt owns s-tile nt;
  //      lane = (stream, cells cb..cb+3) ----
  const int e_s = sbase + wave * TS_ + bs;
  const int cb = c0 + 4 * q;
  const bool e_on = wave < NT && Geo<SMALL>::owner(lane) && e_s < S && cb < C;
  const size_t row = (size_t)t * S + (e_on ? e_s : 0), rown = row + S, rowp = row - S;
  float yg[4], yi[4], yf[4], yo[4], yh[4], cpv[4], dcn[4], fn[4], din[4], dfn[4], wpi[4], wpf[4], wpo[4];
  // 4x4x1 geometry: ONE cell per lane on all 64 lanes of wave nt (cell c0 + lane/4, stream lane%4) -- four cells on each
  // of 16 lanes is four dependent chains of cell math in a row (see k_dmf_v)
  const int s_cell = c0 + (lane >> 2), s_s = sbase + wave * TS_ + (lane & 3);
  const bool s_on = SMALL && wave < NT && s_s < S && s_cell < C;
  const size_t srow = (size_t)t * S + (s_on ? s_s : 0);
  const int slc = s_on ? s_cell : 0;
  float sy[13];
  auto epi = [&]() {
    if constexpr (SMALL) {
      const bool n_on = s_on && !last;
      const size_t rn = last ? srow : srow + S;            // clamped: block T+1 is never dereferenced
      const float *yp = a.gifo + srow * 4 * C + slc;
      sy[0] = yp[0]; sy[1] = yp[C]; sy[2] = yp[2 * C]; sy[3] = yp[3 * C];
      sy[4] = a.hh[srow * C + slc]; sy[5] = a.cc[(srow - S) * C + slc];
      const float d0 = a.dc[rn * C + slc], d1 = a.gifo[rn * 4 * C + 2 * C + slc];
      const float d2 = a.dgifo[rn * 4 * C + C + slc], d3 = a.dgifo[rn * 4 * C + 2 * C + slc];
      sy[6] = n_on ? d0 : 0.f; sy[7] = n_on ? d1 : 0.f; sy[8] = n_on ? d2 : 0.f; sy[9] = n_on ? d3 : 0.f;
      sy[10] = a.pi[slc]; sy[11] = a.pf[slc]; sy[12] = a.po[slc];
      return;
    }
    const float *yp = a.gifo + row * 4 * C;
    load4<true>(yp, cb, C, e_on, yg);
    load4<true>(yp + C, cb, C, e_on, yi);
    load4<true>(yp + 2 * C, cb, C, e_on, yf);
    load4<true>(yp + 3 * C, cb, C, e_on, yo);
    load4<true>(a.hh + row * C, cb, C, e_on, yh);
    load4<true>(a.cc + rowp * C, cb, C, e_on, cpv);
    const bool n_on = e_on && !last;
    const size_t rn = last ? row : rown;                 // clamped: block T+1 is never dereferenced
    load4<true>(a.dc + rn * C, cb, C, n_on, dcn);
    load4<true>(a.gifo + rn * 4 * C + 2 * C, cb, C, n_on, fn);
    load4<true>(a.dgifo + rn * 4 * C + C, cb, C, n_on, din);
    load4<true>(a.dgifo + rn * 4 * C + 2 * C, cb, C, n_on, dfn);
    load4<true>(a.pi, cb, C, e_on, wpi);
    load4<true>(a.pf, cb, C, e_on, wpf);
    load4<true>(a.po, cb, C, e_on, wpo);
  };

  const int nch = (R + KCH - 1) / KCH;
  const bool write_dr = blockIdx.x == 0;
  auto bload = [&](int sl, int k, bool on) -> float4 {    // d_r(t) = out_diff(t) + slabs   (:367, :391), branch-free
    const int s = min(sbase + sl, S - 1), kk = min(k, R - 4);
    float4 v = ldg4(a.out_diff + (size_t)((t - 1) * S + s) * a.od_stride + kk);
    float4 p[KSMAX];
#pragma unroll
    for (int ks = 0; ks < KSMAX; ks++) p[ks] = ldg4(a.part + ((size_t)(ks < a.nslab ? ks : 0) * S + s) * R + kk);
#pragma unroll
    for (int ks = 0; ks < KSMAX; ks++) {
      const bool use = ks < a.nslab;                     // fixed summation order; unused slabs add exactly +0 (select, not
      v.x += use ? p[ks].x : 0.f; v.y += use ? p[ks].y : 0.f;   // multiply: a never-written slab may hold NaN bit patterns)
      v.z += use ? p[ks].z : 0.f; v.w += use ? p[ks].w : 0.f;
    }
    return keep_if(v, on && sbase + sl < S && k < R);
  };
  auto bside = [&](int sl, int k, const float4 &v) {
    const int s = sbase + sl;
    if (write_dr && s < S && k < R) *reinterpret_cast<float4 *>(a.dr + ((size_t)t * S + s) * R + k) = v;
  };
  vec_contract<NT, CPW, SMALL, BF>(va.wpk + (size_t)blockIdx.x * nch * (BF ? 64 : 128), nch, NT * TS_, ldsB, lane, wave, acc, bload, bside, epi);
  VEC_COMBINE();

  if constexpr (SMALL) {
    if (s_on) {
      // rows 4q..4q+3 of stream j live in red[w][nt][4q + j] (k-groups already combined in-wave): this lane's cell is
      // component (lane/4)%4 of q = lane/16
      const float *rp = reinterpret_cast<const float *>(&red[0][wave][((lane >> 4) << 2) | (lane & 3)]) + ((lane >> 2) & 3);
      float dmv = rp[0];
#pragma unroll
      for (int w = 1; w < NW; w++) dmv += rp[(size_t)w * NT * 64 * 4];
      const float d_h = k_diff_tanh(dmv * sy[3], sy[4]);         // :411-412
      const float d_o = k_diff_sigmoid(dmv * sy[4], sy[3]);      // :415-416
      float d_c = d_h;                                           // :424
      d_c = d_c + sy[6] * sy[7];                                 // :425
      d_c = d_c + sy[10] * sy[8];                                // :426
      d_c = d_c + sy[11] * sy[9];                                // :427
      d_c = d_c + sy[12] * d_o;                                  // :428
      float *dp = a.dgifo + srow * 4 * C + s_cell;
      dp[0] = k_diff_tanh(d_c * sy[1], sy[0]);                   // :439-440
      dp[C] = k_diff_sigmoid(d_c * sy[0], sy[1]);                // :435-436
      dp[2 * C] = k_diff_sigmoid(d_c * sy[5], sy[2]);            // :431-432
      dp[3 * C] = d_o;
      a.dc[srow * C + s_cell] = d_c;
    }
    return;
  }
  if (e_on) {
    const f32x4 v = reduce_tile<NT, SMALL>(red, wave, lane);
    const float dm[4] = {v.x, v.y, v.z, v.w};
    float og[4], oi[4], of[4], oo[4], oc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float d_h = k_diff_tanh(dm[j] * yo[j], yh[j]);       // :411-412
      const float d_o = k_diff_sigmoid(dm[j] * yh[j], yo[j]);    // :415-416
      float d_c = d_h;                                           // :424
      d_c = d_c + dcn[j] * fn[j];                                // :425
      d_c = d_c + wpi[j] * din[j];                               // :426
      d_c = d_c + wpf[j] * dfn[j];                               // :427
      d_c = d_c + wpo[j] * d_o;                                  // :428
      of[j] = k_diff_sigmoid(d_c * cpv[j], yf[j]);               // :431-432
      oi[j] = k_diff_sigmoid(d_c * yg[j], yi[j]);                // :435-436
      og[j] = k_diff_tanh(d_c * yi[j], yg[j]);                   // :439-440
      oo[j] = d_o;
      oc[j] = d_c;
    }
    float *dp = a.dgifo + row * 4 * C;
    store4<true>(dp, cb, C, og);
    store4<true>(dp + C, cb, C, oi);
    store4<true>(dp + 2 * C, cb, C, of);
    store4<true>(dp + 3 * C, cb, C, oo);
    store4<true>(a.dc + row * C, cb, C, oc);
  }
}

// =============================================================================================
// MANY-STREAM VECTOR PATH (NumStream > 16 per GPU).  The tiles above make every workgroup re-stage the
// activations of ALL its streams and finish with stream-major (scattered) epilogues -- fine while a step is
// latency-bound, 10x off the MFMA roofline at 64+ streams.  Here the unit of work is one
// (16-row weight tile, 16-stream tile) pair per wave so that every SIMD gets an MFMA chain of 50-64
// instructions: a workgroup = MTW row tiles x KSW K-splits on ONE 16-stream tile (MTW*KSW = 8 waves),
// grid = (row tiles / MTW) x (stream tiles).  The B slab (16 streams x up to 512 k) is staged once and
// shared by the MTW row tiles, the K splits are combined through LDS in fixed order, and the result tile
// [16 streams][MTW*16 rows] is consumed by a coalesced elementwise pass (thread = one stream x one 16-byte
// row segment) whose operands were requested at kernel start.
// =============================================================================================
constexpr int FST = 16;                    // streams per workgroup
// FS = K chunks per slab (16 or 32: the whole contraction in one slab whenever K <= 1024);
// LDS row stride of the B slab FS*32 + 4 floats: conflict-free b128 reads

template <int MTW, int KSW, int FS, bool BF, class BL, class BS>
__device__ __forceinline__ void fat_contract(const float4 *__restrict__ apk, int nch, float *ldsB, int lane, int ksp,
                                             f32x4 (&acc)[2], const BL &bload, const BS &bside) {
  constexpr int FLDB = FS * KCH + 4, FLDBH = FS * KCH + 8;
  constexpr int PER = FS / KSW;              // max chunks of a slab per K split
  constexpr int AU = BF ? 64 : 128;
  unsigned short *ldsH = reinterpret_cast<unsigned short *>(ldsB);
  const int bs = lane & 15, kg = lane >> 4;
  // first slab peeled out of the loop and the activation slab requested before the weights: see vec_contract
  auto iter = [&](int base) {
    const int nc = min(FS, nch - base);
    const int per = (nc + KSW - 1) / KSW;
    const int c0 = ksp * per;
    constexpr int F4ROW = FS * 8, U = (FST * F4ROW) / (NW * 64);
    float4 sv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int idx = threadIdx.x + u * NW * 64;
      const int sl = idx / F4ROW, k = (idx % F4ROW) * 4;
      sv[u] = bload(sl, base * KCH + k, k < nc * KCH);
    }
    float4 a0[PER], a1[PER];
#pragma unroll
    for (int c = 0; c < PER; c++) {
      const int cl = min(c0 + c, nc - 1);
      const float4 *ap = apk + (size_t)(base + cl) * AU + lane;
      a0[c] = ap[0]; a1[c] = BF ? a0[c] : ap[64];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int idx = threadIdx.x + u * NW * 64;
      const int sl = idx / F4ROW, k = (idx % F4ROW) * 4;
      if (k < nc * KCH) {
        if constexpr (BF) *reinterpret_cast<uint2 *>(ldsH + sl * FLDBH + k) = pack_bf16x4(sv[u]);
        else *reinterpret_cast<float4 *>(ldsB + sl * FLDB + k) = sv[u];
        bside(sl, base * KCH + k, sv[u]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < PER; c++) {
      if (c < per && c0 + c < nc) {
        if constexpr (BF) {
          const float4 braw = *reinterpret_cast<const float4 *>(ldsH + bs * FLDBH + (c0 + c) * KCH + kg * 8);
          acc[c & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0[c]), __builtin_bit_cast(bf16x8, braw), acc[c & 1], 0, 0, 0);
          continue;
        }
        const float av[8] = {a0[c].x, a0[c].y, a0[c].z, a0[c].w, a1[c].x, a1[c].y, a1[c].z, a1[c].w};
        const float *bp = ldsB + bs * FLDB + (c0 + c) * KCH + kg * 8;
        const float4 b0 = *reinterpret_cast<const float4 *>(bp), b1 = *reinterpret_cast<const float4 *>(bp + 4);
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j & 1] = MFMA16(av[j], bv[j], acc[j & 1]);
      }
    }
    if (base + FS < nch) __syncthreads();
  };
  if (nch > 0) iter(0);
  for (int base = FS; base < nch; base += FS) iter(base);
}

// K-split combine (fixed order) + scatter of the MFMA tiles into the result tile rt[stream][row], row stride RTS.
// GATES: tile row 4q+reg is (cell mt*4+q, gate reg) -> rt column gate*(4*MTW) + cell;  else row mt*16 + 4q + reg.
template <int MTW, int KSW, bool GATES>
__device__ __forceinline__ void fat_combine(f32x4 (&acc)[2], f32x4 (*red)[MTW][64], float *rt, int lane, int mt, int ksp) {
  constexpr int RTS = MTW * 16 + 4;
  if (ksp > 0) red[ksp - 1][mt][lane] = acc[0] + acc[1];
  __syncthreads();
  if (ksp == 0) {
    f32x4 v = acc[0] + acc[1];
#pragma unroll
    for (int p = 0; p < KSW - 1; p++) v += red[p][mt][lane];
    const int q = lane >> 4;
    float *rp = rt + (lane & 15) * RTS;
    if (GATES) { const int cl = mt * 4 + q; rp[cl] = v.x; rp[4 * MTW + cl] = v.y; rp[8 * MTW + cl] = v.z; rp[12 * MTW + cl] = v.w; }
    else *reinterpret_cast<float4 *>(rp + mt * 16 + 4 * q) = make_float4(v.x, v.y, v.z, v.w);
  }
  __syncthreads();
}

#define FAT_PROLOGUE()                                                                           \
  if ((int)blockIdx.x >= va.gx) return;      /* grid.x is padded to a multiple of 8 (XCD affinity of row tiles) */ \
  const int lane = threadIdx.x & 63;                                                             \
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                             \
  const int mt = wave % MTW, ksp = wave / MTW;                                                   \
  constexpr int RTS = MTW * 16 + 4;                                                              \
  __shared__ __attribute__((aligned(16))) float ldsB[FST * (FS * KCH + 4)];                      \
  __shared__ f32x4 red[KSW - 1][MTW][64];                                                        \
  __shared__ __attribute__((aligned(16))) float rt[FST * RTS];                                   \
  f32x4 acc[2] = {(f32x4){0, 0, 0, 0}, (f32x4){0, 0, 0, 0}}

template <int MTW, int KSW, int FS, bool FUSEX, bool BF>
__global__ __launch_bounds__(NW * 64) void k_gates_f(GatesVArgs va) {
  const GatesArgs &a = va.g;
  FAT_PROLOGUE();
  constexpr int CELLS = 4 * MTW;
  const int C = a.C, R = a.R, S = a.S, I = a.I, t = a.t;
  const int c0 = blockIdx.x * CELLS;
  const int sbase = blockIdx.y * FST;
  // elementwise operands: thread = (stream sbase + tid/CELLS, cell c0 + tid%CELLS)
  const int sl_e = threadIdx.x / CELLS, j_e = threadIdx.x % CELLS;
  const int e_s = sbase + sl_e, e_cell = c0 + j_e;
  const bool e_on = sl_e < FST && e_s < S && e_cell < C;
  const int l_cell = e_on ? e_cell : 0, l_s = e_on ? e_s : 0;
  const size_t e_row = (size_t)t * S + l_s;
  float pre[4];
#pragma unroll
  for (int g = 0; g < 4; g++) pre[g] = FUSEX ? a.bias[g * C + l_cell] : a.gifo[e_row * 4 * C + g * C + l_cell];
  const float cp = a.cprev[(size_t)l_s * C + l_cell];
  const float wpi = a.pi[l_cell], wpf = a.pf[l_cell], wpo = a.po[l_cell];

  const int nchR = (R + KCH - 1) / KCH, Rp = nchR * KCH;
  const int nch = nchR + (FUSEX ? (I + KCH - 1) / KCH : 0);
  const bool mirror_r = a.r_mirror != nullptr && blockIdx.x == 0;
  auto bload = [&](int sl, int k, bool on) -> float4 {    // B = [ r(t-1) | pad | x(t) | pad ], branch-free
    const int s = min(sbase + sl, S - 1);
    const bool inr = k < R, inx = FUSEX && k >= Rp && k - Rp < I;
    const float *p = inr ? a.rprev + (size_t)s * R + k : inx ? a.x + (size_t)s * a.x_stride + (k - Rp) : a.rprev;
    const float4 v = ldg4(p);
    return keep_if(v, on && sbase + sl < S && (inr || inx));
  };
  auto bside = [&](int sl, int k, const float4 &v) {
    const int s = sbase + sl;
    if (mirror_r && s < S && k < R) *reinterpret_cast<float4 *>(a.r_mirror + (size_t)s * R + k) = v;
  };
  fat_contract<MTW, KSW, FS, BF>(va.wpk + (size_t)(blockIdx.x * MTW + mt) * va.nch_total * (BF ? 64 : 128), nch, ldsB, lane, ksp, acc, bload, bside);
  fat_combine<MTW, KSW, true>(acc, red, rt, lane, mt, ksp);

  if (e_on) {
    const float *rp = rt + sl_e * RTS + j_e;
    float ag = rp[0] + pre[0];
    float ai = rp[CELLS] + pre[1];
    float af = rp[2 * CELLS] + pre[2];
    float ao = rp[3 * CELLS] + pre[3];
    ai += wpi * cp;                                // :278
    af += wpf * cp;                                // :281
    const float gi = k_sigmoid(ai), gf = k_sigmoid(af), gg = k_tanh(ag);   // :284-288
    float c = gg * gi;                             // :291
    c = c + cp * gf;                               // :294
    c = c < -50.f ? -50.f : c;                     // :296
    c = c > 50.f ? 50.f : c;                       // :297
    const float h = k_tanh(c);                     // :300
    ao += wpo * c;                                 // :303
    const float go = k_sigmoid(ao);                // :306
    const float m = h * go;                        // :309
    float *gp = a.gifo + e_row * 4 * C + e_cell;
    gp[0] = gg; gp[C] = gi; gp[2 * C] = gf; gp[3 * C] = go;
    a.cc[e_row * C + e_cell] = c;
    a.hh[e_row * C + e_cell] = h;
    a.mm[e_row * C + e_cell] = m;
    if (a.c_mirror) a.c_mirror[(size_t)e_s * C + e_cell] = cp;   // :231 (c columns)
    if (a.c_save) a.c_save[(size_t)e_s * C + e_cell] = c;         // :331 (c columns)
  }
}

template <int MTW, int KSW, int FS, bool BF>
__global__ __launch_bounds__(NW * 64) void k_proj_f(ProjVArgs va) {
  const ProjArgs &a = va.g;
  FAT_PROLOGUE();
  constexpr int Q = MTW * 4;                         // 16-byte row segments per stream
  const int C = a.C, R = a.R, S = a.S, t = a.t;
  const int n0 = blockIdx.x * MTW * 16;
  const int sbase = blockIdx.y * FST;
  const int nch = (C + KCH - 1) / KCH;
  const int ntile = min((int)blockIdx.x * MTW + mt, (R + 15) / 16 - 1);      // clamped: rows past R are dropped below
  const float *mrow = a.mm + (size_t)t * S * C;
  auto bload = [&](int sl, int k, bool on) -> float4 {
    const float4 v = ldg4(mrow + (size_t)min(sbase + sl, S - 1) * C + min(k, C - 4));
    return keep_if(v, on && sbase + sl < S && k < C);
  };
  fat_contract<MTW, KSW, FS, BF>(va.wpk + (size_t)ntile * nch * (BF ? 64 : 128), nch, ldsB, lane, ksp, acc, bload, NoSide());
  fat_combine<MTW, KSW, false>(acc, red, rt, lane, mt, ksp);
  const int sl_e = threadIdx.x / Q, j_e = threadIdx.x % Q;
  const int s = sbase + sl_e, n = n0 + 4 * j_e;
  if (sl_e < FST && s < S && n < R) {
    const float4 v = *reinterpret_cast<const float4 *>(rt + sl_e * RTS + 4 * j_e);
    const float e[4] = {v.x, v.y, v.z, v.w};
    store4<true>(a.rr + ((size_t)t * S + s) * R, n, R, e);
    float *op = a.out + (size_t)((t - 1) * S + s) * a.out_stride;
    if (a.vecOut) store4<true>(op, n, R, e); else store4<false>(op, n, R, e);
    if (a.r_save) store4<true>(a.r_save + (size_t)s * R, n, R, e);
  }
}

template <int MTW, int KSW, int FS, bool BF>
__global__ __launch_bounds__(NW * 64) void k_dr_f(DrVArgs va) {
  const DrArgs &a = va.g;
  FAT_PROLOGUE();
  constexpr int Q = MTW * 4;
  const int R = a.R, I = a.I, S = a.S, K = 4 * a.C;
  const int ntR16 = (R + 15) / 16, ntX16 = (I + 15) / 16;
  const bool is_x = (int)blockIdx.x >= a.ntr;                      // a.ntr counts MTW-tile groups over R here
  const int g = is_x ? (int)blockIdx.x - a.ntr : (int)blockIdx.x;
  const int tile = is_x ? ntR16 + min(g * MTW + mt, ntX16 - 1) : min(g * MTW + mt, ntR16 - 1);
  const int n0 = g * MTW * 16;
  const int N = is_x ? I : R;
  const int sbase = blockIdx.y * FST;
  const int ks = blockIdx.z;
  const int kbeg = ks * a.klen;
  const int kend = min(K, kbeg + a.klen);
  const float *drow = a.dgifo + (size_t)(a.t + 1) * S * K;
  auto bload = [&](int sl, int k, bool on) -> float4 {
    const float4 v = ldg4(drow + (size_t)min(sbase + sl, S - 1) * K + min(kbeg + k, K - 4));
    return keep_if(v, on && sbase + sl < S && kbeg + k < kend);
  };
  fat_contract<MTW, KSW, FS, BF>(va.wpk + ((size_t)tile * va.nch_total + kbeg / KCH) * (BF ? 64 : 128),
                         kend > kbeg ? (kend - kbeg + KCH - 1) / KCH : 0, ldsB, lane, ksp, acc, bload, NoSide());
  fat_combine<MTW, KSW, false>(acc, red, rt, lane, mt, ksp);
  const int sl_e = threadIdx.x / Q, j_e = threadIdx.x % Q;
  const int s = sbase + sl_e, n = n0 + 4 * j_e;
  if (sl_e < FST && s < S && n < N) {
    const float4 v = *reinterpret_cast<const float4 *>(rt + sl_e * RTS + 4 * j_e);
    const float e[4] = {v.x, v.y, v.z, v.w};
    if (is_x) {
      float *xp = a.xpart + ((size_t)ks * S + s) * a.x_ld;
      if (a.vecX) store4<true>(xp, n, I, e); else store4<false>(xp, n, I, e);
    } else {
      store4<true>(a.part + ((size_t)ks * S + s) * R, n, R, e);
    }
  }
}

template <int MTW, int KSW, int FS, bool BF>
__global__ __launch_bounds__(NW * 64) void k_dm_f(DmVArgs va) {
  const DmArgs &a = va.g;
  FAT_PROLOGUE();
  const int C = a.C, R = a.R, S = a.S, t = a.t;
  const int c0 = blockIdx.x * MTW * 16;
  const int sbase = blockIdx.y * FST;
  const bool last = (t == a.T);

  reduce_x_slabs(a, sbase, min(S, sbase + FST), blockIdx.x, va.gx);

  // elementwise operands, requested before the contraction: thread = (stream, ONE cell) -- 16 streams x 16*MTW cells
  // = 256*MTW of the 512 threads (a thread that walks four cells runs four dependent chains of cell math in a row)
  constexpr int NC = MTW * 16;
  const int sl_e = threadIdx.x / NC, j_e = threadIdx.x % NC;
  const int e_s = sbase + sl_e, e_c = c0 + j_e;
  const bool e_on = sl_e < FST && e_s < S && e_c < C;
  const int lc = e_on ? e_c : 0;
  const size_t row = (size_t)t * S + (e_on ? e_s : 0);
  const bool n_on = e_on && !last;
  const size_t rn = last ? row : row + S;                // clamped: block T+1 is never dereferenced
  const float *yp = a.gifo + row * 4 * C + lc;
  const float yg = yp[0], yi = yp[C], yf = yp[2 * C], yo = yp[3 * C];
  const float yh = a.hh[row * C + lc], cpv = a.cc[(row - S) * C + lc];
  const float dcn_r = a.dc[rn * C + lc], fn_r = a.gifo[rn * 4 * C + 2 * C + lc];
  const float din_r = a.dgifo[rn * 4 * C + C + lc], dfn_r = a.dgifo[rn * 4 * C + 2 * C + lc];
  const float wpi = a.pi[lc], wpf = a.pf[lc], wpo = a.po[lc];

  const int nch = (R + KCH - 1) / KCH;
  const int ctile = min((int)blockIdx.x * MTW + mt, (C + 15) / 16 - 1);
  const bool write_dr = blockIdx.x == 0;
  auto bload = [&](int sl, int k, bool on) -> float4 {    // d_r(t) = out_diff(t) + slabs   (:367, :391), branch-free
    const int s = min(sbase + sl, S - 1), kk = min(k, R - 4);
    float4 v = ldg4(a.out_diff + (size_t)((t - 1) * S + s) * a.od_stride + kk);
    float4 p[KSMAX];
#pragma unroll
    for (int ks = 0; ks < KSMAX; ks++) p[ks] = ldg4(a.part + ((size_t)(ks < a.nslab ? ks : 0) * S + s) * R + kk);
#pragma unroll
    for (int ks = 0; ks < KSMAX; ks++) {
      const bool use = ks < a.nslab;                     // fixed summation order; unused slabs add exactly +0 (select, not
      v.x += use ? p[ks].x : 0.f; v.y += use ? p[ks].y : 0.f;   // multiply: a never-written slab may hold NaN bit patterns)
      v.z += use ? p[ks].z : 0.f; v.w += use ? p[ks].w : 0.f;
    }
    return keep_if(v, on && sbase + sl < S && k < R);
  };
  auto bside = [&](int sl, int k, const float4 &v) {
    const int s = sbase + sl;
    if (write_dr && s < S && k < R) *reinterpret_cast<float4 *>(a.dr + ((size_t)t * S + s) * R + k) = v;
  };
  fat_contract<MTW, KSW, FS, BF>(va.wpk + (size_t)ctile * nch * (BF ? 64 : 128), nch, ldsB, lane, ksp, acc, bload, bside);
  fat_combine<MTW, KSW, false>(acc, red, rt, lane, mt, ksp);

  if (e_on) {
    const float dm = rt[sl_e * RTS + j_e];
    const float dcn = n_on ? dcn_r : 0.f, fn = n_on ? fn_r : 0.f, din = n_on ? din_r : 0.f, dfn = n_on ? dfn_r : 0.f;
    const float d_h = k_diff_tanh(dm * yo, yh);                  // :411-412
    const float d_o = k_diff_sigmoid(dm * yh, yo);               // :415-416
    float d_c = d_h;                                             // :424
    d_c = d_c + dcn * fn;                                        // :425
    d_c = d_c + wpi * din;                                       // :426
    d_c = d_c + wpf * dfn;                                       // :427
    d_c = d_c + wpo * d_o;                                       // :428
    float *dp = a.dgifo + row * 4 * C + e_c;
    dp[0] = k_diff_tanh(d_c * yi, yg);                           // :439-440
    dp[C] = k_diff_sigmoid(d_c * yg, yi);                        // :435-436
    dp[2 * C] = k_diff_sigmoid(d_c * cpv, yf);                   // :431-432
    dp[3 * C] = d_o;
    a.dc[row * C + e_c] = d_c;
  }
}

// =============================================================================================
// FOLDED RECURRENCE (NumStream <= small_max, engine option "fold").  At 4-12 streams a step is pure latency:
// two dependent kernels per direction (gates needs all of r, the projection all of m), ~4 us each, MFMA idle.
// With  W_rm = W_gifo_r * W_r_m  [4C x C]  (recomputed after every Update) the recurrence closes over m alone:
//     forward   a(t)   = W_x x(t) + b + W_rm m(t-1)                          (t >= 2; t = 1 uses the carried r, :275)
//     backward  d_m(t) = out_diff(t) W_r_m + dgifo(t+1) W_rm                 (:391 substituted into :408)
// ONE kernel per step and direction; r(1..T) (:312), d_r(1..T) (:391) and in_diff (:457) become batched GEMMs
// outside the chain.  Same algebra as the reference up to fp32 summation order (the parity tests run both paths).
// The forward step is k_gates_v itself with B = m(t-1) and the packed [W_rm | W_x]; the backward step needs a
// contraction over K = 4C for only C rows, i.e. 4-row tiles to fill the chip:
//   4-row geometry of the 4x4x1_16b MFMA: block b = k-group (16 of them), A lane 4b+i = row i, B lane 4b+j = stream j,
//   chunk = 16 k-groups x 8 = 128 k;  packed weights pk[tile4][chunk][h][lane][4]: row = 4*tile + (lane&3),
//   k = 128*chunk + 64h + 4*(lane>>2) + e.
// =============================================================================================
constexpr int KCH4 = 128;

// acc += A(4 rows, chunks [0,nch)) * B(4 streams, same chunks).  Chunk c is dealt to wave c % NW.  No LDS staging: the B
// operand of consumer lane 4b+j (k-group b, stream j) is float4 b of each 64-float half of the chunk in stream j's row;
// it is FETCHED by loader lane 16j+b, so that a load instruction covers 4 x 256 contiguous bytes (lanes of a quad on
// one cache line -- a quad spread over four rows costs four tag lookups), and moved with ds_bpermute.
// brow: row pointer of the LOADER lane's stream (clamped), bok: the CONSUMER lane's stream exists.
__device__ __forceinline__ float perm(int byte_idx, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_idx, __builtin_bit_cast(int, v)));
}
// NG stream groups of 4 share one fetch of the weight chunk.  brow: row pointer of the LOADER lane's stream within group 0
// (clamped per group by the caller through gstride = 4*K or 0), bok[g]: the CONSUMER lane's stream of group g exists.
template <int CPW, int NG>
__device__ __forceinline__ void vec_contract_r4(const float4 *__restrict__ apk, int nch, int K, const float *const (&brow)[NG],
                                                const bool (&bok)[NG], int lane, int wave, int rot, f32x4 (&acc)[NG][2]) {
  const int kg = lane >> 2;
  for (int base = 0; base < nch; base += CPW * NW) {
    float4 a0[CPW], a1[CPW], b0[NG][CPW], b1[NG][CPW];
#pragma unroll
    for (int c = 0; c < CPW; c++) {
      const int ch = base + c * NW + wave;
      int cl = min(ch, nch - 1) + rot;                             // clamped: always a valid chunk, unused if off
      cl = cl >= nch ? cl - nch : cl;                              // rotated: see k_dmf_v
      const float4 *ap = apk + (size_t)cl * 128 + lane;
      a0[c] = ap[0]; a1[c] = ap[64];
      // loader role: lane = 16*stream + q fetches float4 q of each 64-float half of the chunk (4 x 256 contiguous bytes
      // per instruction); the consumer lane 4b+j takes its operand from loader lane 16j+b below
      const int k = cl * KCH4 + (lane & 15) * 4;
#pragma unroll
      for (int g = 0; g < NG; g++) { b0[g][c] = ldg4(brow[g] + min(k, K - 4)); b1[g][c] = ldg4(brow[g] + min(k + 64, K - 4)); }
    }
#pragma unroll
    for (int c = 0; c < CPW; c++) {
      const int ch = base + c * NW + wave;
      int cl = min(ch, nch - 1) + rot;
      cl = cl >= nch ? cl - nch : cl;
      const bool k0 = ch < nch && cl * KCH4 + kg * 4 < K, k1 = ch < nch && cl * KCH4 + 64 + kg * 4 < K;
      const float av[8] = {a0[c].x, a0[c].y, a0[c].z, a0[c].w, a1[c].x, a1[c].y, a1[c].z, a1[c].w};
      const int src = (((lane & 3) << 4) | kg) << 2;               // byte index of loader lane 16j+b
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const bool on0 = k0 && bok[g], on1 = k1 && bok[g];
        const float p[8] = {perm(src, b0[g][c].x), perm(src, b0[g][c].y), perm(src, b0[g][c].z), perm(src, b0[g][c].w),
                            perm(src, b1[g][c].x), perm(src, b1[g][c].y), perm(src, b1[g][c].z), perm(src, b1[g][c].w)};
        const float bv[8] = {on0 ? p[0] : 0.f, on0 ? p[1] : 0.f, on0 ? p[2] : 0.f, on0 ? p[3] : 0.f,
                             on1 ? p[4] : 0.f, on1 ? p[5] : 0.f, on1 ? p[6] : 0.f, on1 ? p[7] : 0.f};
#pragma unroll
        for (int j = 0; j < 8; j++) acc[g][j & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], bv[j], acc[g][j & 1], 0, 0, 0);
      }
    }
  }
}

struct DmfArgs {
  int C, S, T, t;
  const float *pi, *pf, *po;
  const float *gifo, *cc, *hh;
  float *dgifo, *dc;
  const float *P;          // out_diff * W_r_m for all frames [T*S x C]
  const float4 *wpk;       // packed W_rm^T, 4-row geometry: [C/4 tiles][nch_total chunks][2][64]
  int nch_total;           // chunks of 128 over 4C
  int nch;                 // chunks to contract: nch_total, or 0 at t == T (dgifo(T+1) = 0, :351)
};

template <int CPW, int NG>
__global__ __launch_bounds__(NW * 64) void k_dmf_v(DmfArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ f32x4 red[NW][NG][4];
  const int C = a.C, S = a.S, t = a.t, K = 4 * a.C;
  const int c0 = blockIdx.x * 4;
  const int sbase = blockIdx.y * 4 * NG;
  const bool last = (t == a.T);

  // ---- epilogue operands first: lanes 0..15 of wave g = (cell c0 + lane/4, stream sbase + 4g + lane%4): one cell per
  //      lane, so the 15-operation cell math is one short dependent chain instead of four in a row ----
  const int e_i = (lane >> 2) & 3, e_j = lane & 3;
  const int e_s = sbase + 4 * wave + e_j, e_c = c0 + e_i;
  const bool e_on = wave < NG && lane < 16 && e_s < S && e_c < C;
  const size_t row = (size_t)t * S + (e_on ? e_s : 0), rown = row + S, rowp = row - S;
  const int lc = e_on ? e_c : 0;
  const bool n_on = e_on && !last;
  const size_t rn = last ? row : rown;                   // clamped: block T+1 is never dereferenced
  const float *yp = a.gifo + row * 4 * C + lc;
  const float yg = yp[0], yi = yp[C], yf = yp[2 * C], yo = yp[3 * C];
  const float yh = a.hh[row * C + lc], cpv = a.cc[rowp * C + lc];
  const float dcn_r = a.dc[rn * C + lc], fn_r = a.gifo[rn * 4 * C + 2 * C + lc];
  const float din_r = a.dgifo[rn * 4 * C + C + lc], dfn_r = a.dgifo[rn * 4 * C + 2 * C + lc];
  const float wpi = a.pi[lc], wpf = a.pf[lc], wpo = a.po[lc];
  const float pv = a.P[(row - S) * C + lc];              // frame t is row block t-1 of P

  f32x4 acc[NG][2];
  const float *brow[NG];
  bool bok[NG];
#pragma unroll
  for (int g = 0; g < NG; g++) {
    acc[g][0] = (f32x4){0, 0, 0, 0}; acc[g][1] = (f32x4){0, 0, 0, 0};
    bok[g] = sbase + 4 * g + (lane & 3) < S;                                                          // consumer role
    brow[g] = a.dgifo + ((size_t)(last ? t : t + 1) * S + min(sbase + 4 * g + (lane >> 4), S - 1)) * K;   // loader role
  }
  // every workgroup reads the SAME activation rows: started at the same chunk they would all hit one L2 channel at a
  // time, so workgroup w walks the chunks rotated by 7w (the weights are private, their order does not matter)
  const int rot = a.nch > 0 ? (int)((blockIdx.x * 7u) % (unsigned)a.nch) : 0;
  vec_contract_r4<CPW, NG>(a.wpk + (size_t)blockIdx.x * a.nch_total * 128, a.nch, K, brow, bok, lane, wave, rot, acc);

  // the 16 k-groups of a (row, stream) pair sit in the lanes with equal lane&3: xor butterfly (a+b == b+a bitwise, so
  // every lane ends with the same sum), then the 8 waves in fixed order
#pragma unroll
  for (int g = 0; g < NG; g++) {
    f32x4 v = acc[g][0] + acc[g][1];
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) {
      v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
    }
    if (lane < 4) red[wave][g][lane] = v;
  }
  __syncthreads();

  if (e_on) {
    // row (cell) e_i of stream e_j: component e_i of red[w][wave][e_j]
    const float *rp = reinterpret_cast<const float *>(&red[0][0][0]) + (wave * 4 + e_j) * 4 + e_i;
    float sum = rp[0];
#pragma unroll
    for (int w = 1; w < NW; w++) sum += rp[w * NG * 16];
    const float dm = sum + pv;                                   // :408 with :391 substituted
    const float dcn = n_on ? dcn_r : 0.f, fn = n_on ? fn_r : 0.f, din = n_on ? din_r : 0.f, dfn = n_on ? dfn_r : 0.f;
    const float d_h = k_diff_tanh(dm * yo, yh);                  // :411-412
    const float d_o = k_diff_sigmoid(dm * yh, yo);               // :415-416
    float d_c = d_h;                                             // :424
    d_c = d_c + dcn * fn;                                        // :425
    d_c = d_c + wpi * din;                                       // :426
    d_c = d_c + wpf * dfn;                                       // :427
    d_c = d_c + wpo * d_o;                                       // :428
    const float o_f = k_diff_sigmoid(d_c * cpv, yf);             // :431-432
    const float o_i = k_diff_sigmoid(d_c * yg, yi);              // :435-436
    const float o_g = k_diff_tanh(d_c * yi, yg);                 // :439-440
    float *dp = a.dgifo + row * 4 * C + e_c;
    dp[0] = o_g; dp[C] = o_i; dp[2 * C] = o_f; dp[3 * C] = d_o;
    a.dc[row * C + e_c] = d_c;
    if (last) {     // the batched d_r product reads dgifo(T+1) as rows of its operand: keep that block zero (:351) even
      float *zp = a.dgifo + rown * 4 * C + e_c;                  // after a longer minibatch has used it
      zp[0] = 0.f; zp[C] = 0.f; zp[2 * C] = 0.f; zp[3 * C] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_pack: (re)build the four packed weight copies from the natural blob + its transposes.
// One thread per packed float4 (destination-contiguous, so the writes are coalesced 1 KB/wave).
//   array 0 gates: rows (cell-major: tile row i -> gate i&3, cell 4*tile + i>>2), k over [R | pad | I | pad]
//   array 1 proj : rows n of W_r_m   [R x C],   k over C
//   array 2 dr   : rows n of [W_gifo_r^T ; W_gifo_x^T]  ([R x 4C] tiles then [I x 4C] tiles), k over 4C
//   array 3 dm   : rows c of W_r_m^T [C x R],   k over R
// ---------------------------------------------------------------------------------------------
struct PackArgs {
  int C, R, I;
  const float *wx, *wr, *wm, *wrT, *wmT, *wxT;
  float4 *pk[4];
  long n4[4];          // 16-byte vector count of each array (0: array not selected in this launch)
  int nch[4];          // chunks per tile
  int bf16;            // 1: entries are 8 bf16 (RNE of the fp32 master) covering k..k+7 -> half as many vectors
  float4 *foldx;       // fp32 only: the W_x chunks of array 0 are also the W_x chunks of the folded gates array
  int nch_fold, nchm_fold;
};

__global__ __launch_bounds__(256) void k_pack(PackArgs a) {
  const int C = a.C, R = a.R, I = a.I;
  const long total = a.n4[0] + a.n4[1] + a.n4[2] + a.n4[3];
  for (long gid = blockIdx.x * 256L + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    int arr = 0; long id = gid;
    while (id >= a.n4[arr]) { id -= a.n4[arr]; arr++; }
    const int lane = (int)(id & 63), h = a.bf16 ? 0 : (int)((id >> 6) & 1);
    const long tc = a.bf16 ? id >> 6 : id >> 7;
    const int nch = a.nch[arr];
    const int tile = (int)(tc / nch), ch = (int)(tc - (long)tile * nch);
    const int i = lane & 15;
    const int k = ch * KCH + (lane >> 4) * 8 + h * 4;
    const float *src = nullptr;
    int klim = 0, koff = k;
    bool row_ok = false;
    if (arr == 0) {
      const int cell = tile * 4 + (i >> 2), gate = i & 3;
      row_ok = cell < C;
      const int nchR = (R + KCH - 1) / KCH;
      if (ch < nchR) { src = a.wr + ((size_t)gate * C + (row_ok ? cell : 0)) * R; klim = R; }
      else { src = a.wx + ((size_t)gate * C + (row_ok ? cell : 0)) * I; klim = I; koff = k - nchR * KCH; }
    } else if (arr == 1) {
      const int n = tile * 16 + i; row_ok = n < R; src = a.wm + (size_t)(row_ok ? n : 0) * C; klim = C;
    } else if (arr == 2) {
      const int ntR = (R + 15) / 16;
      if (tile < ntR) { const int n = tile * 16 + i; row_ok = n < R; src = a.wrT + (size_t)(row_ok ? n : 0) * 4 * C; }
      else { const int n = (tile - ntR) * 16 + i; row_ok = n < I; src = a.wxT + (size_t)(row_ok ? n : 0) * 4 * C; }
      klim = 4 * C;
    } else {
      const int c = tile * 16 + i; row_ok = c < C; src = a.wmT + (size_t)(row_ok ? c : 0) * R; klim = R;
    }
    float4 v = f4zero();
    if (row_ok && koff + 4 <= klim) v = ldg4(src + koff);     // all extents are multiples of 8 on this path
    if (a.bf16) {
      float4 w = f4zero();
      if (row_ok && koff + 8 <= klim) w = ldg4(src + koff + 4);
      const uint2 lo = pack_bf16x4(v), hi = pack_bf16x4(w);
      v = __builtin_bit_cast(float4, make_uint4(lo.x, lo.y, hi.x, hi.y));
    }
    a.pk[arr][id] = v;
    if (arr == 0 && a.foldx && !a.bf16) {
      const int nchR = (R + KCH - 1) / KCH;
      if (ch >= nchR) a.foldx[(((size_t)tile * a.nch_fold + a.nchm_fold + (ch - nchR)) * 2 + h) * 64 + lane] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-tiled MFMA GEMM tile (64x64 output, BK = 32, 256 threads = 2x2 waves of 32x32).  The next
// K tile is fetched into registers while the current one is multiplied out of LDS.
// ---------------------------------------------------------------------------------------------
constexpr int GT = 64, GK = 64, GLD = 80;   // LDS row stride 80 floats: k-groups land on disjoint banks

struct GemmJob {
  int M, N, K;
  const float *A; int lda;
  const float *B; int ldb;
  float beta;
  float *Cm; int ldc;
  const float *bias;
  int vecA, vecB;
  // optional extra destinations of the same result (all null by default):
  float *Ct; int ldct;    // transposed copy  Ct[n][m]  (M % 4 == 0)
  float *C2; int ldc2;    // second copy      C2[m][n]
  float *C3; int tail0;   // rows m >= tail0 also to C3[m - tail0][n], dense (ld = N)
  int kslice = 0;         // k_gemm_bf16_nt only: split K, slice length (0: off)
  // fold product only (launch_fold): rows of A are read in gates-packed order (gperm = C) and the result goes straight
  // into the two packed operand arrays of the folded step kernels instead of Cm
  int gperm;
  float4 *pk1; int nch1;  // [W_rm | W_x] gates array: [C/4 tiles][nch1 chunks of 32][2][64]
  float4 *pk2; int nch2;  // W_rm^T 4-row array:       [C/4 tiles][nch2 chunks of 128][2][64]
  // gradient product with the Update folded in (launch_grads with a GradsUpdate): Cm is the momentum buffer,
  // Cm = beta*Cm + A*B (:468-487), clipped if clip > 0, then P -= lr*Cm (:504-512); Ct then receives the UPDATED P
  float *P; float lr, clip;
  int s3mode;                                // plane format (klstm_math.h split_store4)
  unsigned short *s3; long s3pl; int s3t;   // the bf16 / fp16 planes of the UPDATED P (s3t = 0: P's layout, ld = ldc; 1: Ct's layout, ld = ldct)
  unsigned short *cth = nullptr;            // (or null) bf16 copy (RNE) of Ct, same layout: the B operand of klstm_gemm16.hip's LDS-DMA form
  int coal;               // 1: Cm = beta*Cm + A*B through the same coalesced 16-byte epilogue without P (N, ldc % 4 == 0, aligned, no bias)
  const unsigned *guard = nullptr;   // k_gemm only: the engine's control words; a persistent launch in front gave up ([2] | [6]) -> write nothing
};

// One operand tile = GT x GK elements = 2 x (8 floats per thread).  Operand stored [X x K] (TA=false: 8 consecutive k
// of one x) or [K x X] (TA=true: 8 consecutive x of one k).  `vec` (block-uniform): rows 16-byte aligned and the
// contiguous extent a multiple of 8 -> branch-free loads.
template <bool TA>
__device__ __forceinline__ void fetch_tile(const float *__restrict__ P, int ld, bool vec, int X, int K, int x0, int k0,
                                           int tid, float (&r)[2][8], int gperm = 0) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    if (!TA) {
      const int x = x0 + (tid >> 2), k = k0 + h * 32 + (tid & 3) * 8;
      // gperm = C: logical row x = 4*cell + gate of the gates-packed order reads stored row gate*C + cell
      const int xs = x < X ? (gperm ? (x & 3) * gperm + (x >> 2) : x) : 0;
      const float *row = P + (size_t)xs * ld;
      if (vec) load8<true>(row, k, K, x < X, r[h]); else load8<false>(row, k, K, x < X, r[h]);
    } else {
      const int k = k0 + h * 32 + (tid >> 3), x = x0 + (tid & 7) * 8;
      const float *row = P + (size_t)(k < K ? k : 0) * ld;
      if (vec) load8<true>(row, x, X, k < K, r[h]); else load8<false>(row, x, X, k < K, r[h]);
    }
  }
}
// The same tile for operands whose rows are 16-byte aligned with a contiguous extent that is a multiple of 8 (every
// product of the engine itself): straight-line code -- clamped addresses, unconditional 16-byte loads, padding zeroed
// with keep_if.  (fetch_tile's runtime `vec` switch and its selects compile to conditional blocks whose joins drain
// every outstanding load: the four fetches of a K tile became four serialized memory round trips.)
struct RawTile { float4 v[2][2]; bool ok[2]; };
template <bool TA>
__device__ __forceinline__ void fetch_tile_vec(const float *__restrict__ P, int ld, int X, int K, int x0, int k0, int tid,
                                               RawTile &t, int gperm = 0) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const float *p;
    if (!TA) {
      const int x = x0 + (tid >> 2), k = k0 + h * 32 + (tid & 3) * 8;
      t.ok[h] = x < X && k < K;                            // K % 8 == 0: an 8-group never straddles the end
      const int xc = x < X ? x : 0;
      const int xs = gperm ? (xc & 3) * gperm + (xc >> 2) : xc;
      p = P + (size_t)xs * ld + min(k, K - 8);
    } else {
      const int k = k0 + h * 32 + (tid >> 3), x = x0 + (tid & 7) * 8;
      t.ok[h] = k < K && x < X;                            // X % 8 == 0
      p = P + (size_t)min(k, K - 1) * ld + min(x, X - 8);
    }
    t.v[h][0] = ldg4(p); t.v[h][1] = ldg4(p + 4);
  }
}
// The [K x X] tile of an operand that OTHER workgroups of the same launch have just written with write-through stores (k_grads_tm: the
// d_r rows from the reduce workgroups): 16-byte sc1 loads -- served from beyond this XCD's L2, which may hold a line of it that another
// XCD completed later.  Same addresses, same padding rule.
__device__ __forceinline__ void fetch_tile_vec_coh(const float *__restrict__ P, int ld, int X, int K, int x0, int k0, int tid, RawTile &t) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P), 0, K * ld * 4, 0x00020000);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int k = k0 + h * 32 + (tid >> 3), x = x0 + (tid & 7) * 8;
    t.ok[h] = k < K && x < X;
    const int off = (min(k, K - 1) * ld + min(x, X - 8)) * 4;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16), q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16);   // aux 16 = sc1
    t.v[h][0] = __builtin_bit_cast(float4, q0); t.v[h][1] = __builtin_bit_cast(float4, q1);
  }
}
// the padding mask is applied HERE, one K tile later: touching the fetched registers any earlier makes the wave wait for
// the loads before it multiplies the current tile
__device__ __forceinline__ void unpack_tile(const RawTile &t, float (&r)[2][8]) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const float4 a = keep_if(t.v[h][0], t.ok[h]), b = keep_if(t.v[h][1], t.ok[h]);
    r[h][0] = a.x; r[h][1] = a.y; r[h][2] = a.z; r[h][3] = a.w;
    r[h][4] = b.x; r[h][5] = b.y; r[h][6] = b.z; r[h][7] = b.w;
  }
}
// LDS layout of an operand tile follows its storage so that the stash is two 16-byte stores either way:
//   TA (stored [K x X]):  Ls[k*GLD + x]   (GLD = 80: the four k-groups of an MFMA read land on disjoint banks)
//   !TA (stored [X x K]): Ls[x*GLX + k]   (GLX = 68: row x shifts the bank by 4, again disjoint for 16 rows x 4 k-groups)
constexpr int GLX = 68;
constexpr int GLDS = GK * GLD;               // floats per operand buffer (>= GT * GLX)
template <bool TA>
__device__ __forceinline__ void stash_tile(float *Ls, int tid, const float (&r)[2][8]) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    float *dst = TA ? Ls + (h * 32 + (tid >> 3)) * GLD + (tid & 7) * 8 : Ls + (tid >> 2) * GLX + h * 32 + (tid & 3) * 8;
    *reinterpret_cast<float4 *>(dst) = make_float4(r[h][0], r[h][1], r[h][2], r[h][3]);
    *reinterpret_cast<float4 *>(dst + 4) = make_float4(r[h][4], r[h][5], r[h][6], r[h][7]);
  }
}
template <bool TA>
__device__ __forceinline__ float lds_operand(const float *Ls, int k, int x) { return TA ? Ls[k * GLD + x] : Ls[x * GLX + k]; }

// 64x64 output tile, 4 waves (2x2) of 32x32, K tile 64.  The next K tile is fetched into registers while the
// current one is multiplied out of LDS (global latency hides under 16 k-steps x 4 MFMAs per wave).
template <bool TA, bool TB, bool VEC, bool COHA = false>   // COHA: A through fetch_tile_vec_coh (TA, VEC)
__device__ __forceinline__ void gemm_tile_impl(const GemmJob &g, int m0, int n0, float *As, float *Bs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0, 0, 0, 0};

  float ra[2][8], rb[2][8];
  RawTile ta, tb;
  auto fetch = [&](int k0) {
    if constexpr (VEC) {
      if constexpr (COHA) fetch_tile_vec_coh(g.A, g.lda, g.M, g.K, m0, k0, tid, ta);
      else fetch_tile_vec<TA>(g.A, g.lda, g.M, g.K, m0, k0, tid, ta, g.gperm);
      fetch_tile_vec<!TB>(g.B, g.ldb, g.N, g.K, n0, k0, tid, tb);   // B [N x K] when TB, else [K x N]
    } else {
      fetch_tile<TA>(g.A, g.lda, g.vecA, g.M, g.K, m0, k0, tid, ra, g.gperm);
      fetch_tile<!TB>(g.B, g.ldb, g.vecB, g.N, g.K, n0, k0, tid, rb);
    }
  };
  fetch(0);
  // beta != 0 (momentum folded into the gradient products, :468-487): the old C tile is requested now so that its
  // HBM latency hides under the K loop instead of sitting in front of the stores
  float cold[2][2][4];
  if (g.beta != 0.f && !g.P && !g.coal) {
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < 2; ni++) {
        const int n = n0 + wc * 32 + ni * 16 + i16;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int m = m0 + wr * 32 + mi * 16 + 4 * kg + r;
          const bool ok = n < g.N && m < g.M;
          const float v = g.Cm[ok ? (size_t)m * g.ldc + n : 0];
          cold[mi][ni][r] = ok ? v : 0.f;
        }
      }
  }
  for (int k0 = 0; k0 < g.K; k0 += GK) {
    if constexpr (VEC) { unpack_tile(ta, ra); unpack_tile(tb, rb); }
    stash_tile<TA>(As, tid, ra);
    stash_tile<!TB>(Bs, tid, rb);
    __syncthreads();
    if (k0 + GK < g.K) fetch(k0 + GK);
    // full K tiles run a fully unrolled 16-step body (a runtime trip count defeats the unroller and leaves a rolled
    // ds_read -> MFMA loop); only the last, short tile takes the rolled path and skips its all-zero tail steps
    auto kstep = [&](int kk) {
      const int k = kk * 4 + kg;
      const float a0 = lds_operand<TA>(As, k, wr * 32 + i16), a1 = lds_operand<TA>(As, k, wr * 32 + 16 + i16);
      const float b0 = lds_operand<!TB>(Bs, k, wc * 32 + i16), b1 = lds_operand<!TB>(Bs, k, wc * 32 + 16 + i16);
      acc[0][0] = MFMA16(a0, b0, acc[0][0]);
      acc[0][1] = MFMA16(a0, b1, acc[0][1]);
      acc[1][0] = MFMA16(a1, b0, acc[1][0]);
      acc[1][1] = MFMA16(a1, b1, acc[1][1]);
    };
    if (!TA && TB) {
      // both operands k-contiguous in LDS ([x][k]): one 16-byte read per operand block feeds FOUR k-steps.  Within a
      // group of 16 k the MFMA k-lane kg of step e contracts k = 16j + 4kg + e -- the same bijection for A and B, so the
      // sum is over every k exactly once (zero-filled past K)
#pragma unroll
      for (int j = 0; j < GK / 16; j++) {
        const int ko = 16 * j + 4 * kg;
        const float4 a0 = *reinterpret_cast<const float4 *>(As + (wr * 32 + i16) * GLX + ko);
        const float4 a1 = *reinterpret_cast<const float4 *>(As + (wr * 32 + 16 + i16) * GLX + ko);
        const float4 b0 = *reinterpret_cast<const float4 *>(Bs + (wc * 32 + i16) * GLX + ko);
        const float4 b1 = *reinterpret_cast<const float4 *>(Bs + (wc * 32 + 16 + i16) * GLX + ko);
        const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
        const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          acc[0][0] = MFMA16(av0[e], bv0[e], acc[0][0]);
          acc[0][1] = MFMA16(av0[e], bv1[e], acc[0][1]);
          acc[1][0] = MFMA16(av1[e], bv0[e], acc[1][0]);
          acc[1][1] = MFMA16(av1[e], bv1[e], acc[1][1]);
        }
      }
    } else if (k0 + GK <= g.K) {
#pragma unroll
      for (int kk = 0; kk < GK / 4; kk++) kstep(kk);
    } else {
      const int ksteps = (g.K - k0 + 3) / 4;
      for (int kk = 0; kk < ksteps; kk++) kstep(kk);
    }
    __syncthreads();
  }
  if (g.P || g.coal) {
    // Gradient product with momentum and (P != null) Update folded in (launch_grads + GradsUpdate; N, ldc, ldct multiples of 4, 16-byte
    // aligned blobs).  Everything that touches HBM moves as 16-byte pieces, 256 contiguous bytes per tile row: the tile goes
    // through LDS (the K loop ended with a barrier, As is free), then  corr = beta*corr + grad (:468-487), clip,
    // theta -= lr*corr (:504-512)  on float4 rows, then the transposed copy of the updated parameters.
    float *Cs = As;                                    // 64 x GLX floats <= GLDS
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < 2; ni++) {
        const float e[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
#pragma unroll
        for (int r = 0; r < 4; r++) Cs[(wr * 32 + mi * 16 + 4 * kg + r) * GLX + wc * 32 + ni * 16 + i16] = e[r];
      }
    // this thread's four pieces of the old corr and parameter tiles, all requested before the first store (a load behind a store
    // waits for it: the compiler cannot know that the rows do not overlap -- four memory round trips per tile instead of one)
    float4 oc4[4], op4[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int p = tid + 256 * u, m = m0 + (p >> 4), n = n0 + (p & 15) * 4;
      const size_t off = m < g.M && n + 4 <= g.N ? (size_t)m * g.ldc + n : 0;
      oc4[u] = g.beta != 0.f ? *reinterpret_cast<const float4 *>(g.Cm + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      op4[u] = g.P ? *reinterpret_cast<const float4 *>(g.P + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int p = tid + 256 * u, ml = p >> 4, nq = (p & 15) * 4;
      const int m = m0 + ml, n = n0 + nq;
      if (m < g.M && n + 4 <= g.N) {
        float *cs = Cs + ml * GLX + nq;
        const float4 a = *reinterpret_cast<const float4 *>(cs);
        float4 *cp = reinterpret_cast<float4 *>(g.Cm + (size_t)m * g.ldc + n);
        float4 *pp = reinterpret_cast<float4 *>((g.P ? g.P : g.Cm) + (size_t)m * g.ldc + n);
        float4 pv = op4[u];
        float c[4] = {a.x, a.y, a.z, a.w};
        if (g.beta != 0.f) {
          const float4 o = oc4[u];
          c[0] = g.beta * o.x + c[0]; c[1] = g.beta * o.y + c[1]; c[2] = g.beta * o.z + c[2]; c[3] = g.beta * o.w + c[3];
        }
        if (g.clip > 0.f) {
#pragma unroll
          for (int q = 0; q < 4; q++) { c[q] = c[q] < -g.clip ? -g.clip : c[q]; c[q] = c[q] > g.clip ? g.clip : c[q]; }
        }
        // (plain stores.  Measured and dropped, round 6: corr / parameters / planes as nontemporal stores -- the launch 14.4 -> 15.6 us and
        //  the fold behind it 13.9 -> 14.4 us at 40/800/512: it finds the planes this pass has just written closer than HBM.)
        *cp = make_float4(c[0], c[1], c[2], c[3]);
        if (g.P) {
          pv.x = pv.x + (-g.lr) * c[0]; pv.y = pv.y + (-g.lr) * c[1]; pv.z = pv.z + (-g.lr) * c[2]; pv.w = pv.w + (-g.lr) * c[3];
          *pp = pv;
          *reinterpret_cast<float4 *>(cs) = pv;
          if (g.s3 && !g.s3t) {
            const float v4[4] = {pv.x, pv.y, pv.z, pv.w};
            split_store4(g.s3mode, v4, g.s3 + (size_t)m * g.ldc + n, g.s3pl);
          }
        }
      }
    }
    if (!g.Ct || !g.P) return;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int p = tid + 256 * u, nl = p >> 4, mq = (p & 15) * 4;
      const int n = n0 + nl, m = m0 + mq;
      if (n < g.N && m + 4 <= g.M) {
        const float *cs = Cs + mq * GLX + nl;
        const float v4[4] = {cs[0], cs[GLX], cs[2 * GLX], cs[3 * GLX]};
        *reinterpret_cast<float4 *>(g.Ct + (size_t)n * g.ldct + m) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        if (g.s3 && g.s3t) split_store4(g.s3mode, v4, g.s3 + (size_t)n * g.ldct + m, g.s3pl);
        if (g.cth) split_store4(3, v4, g.cth + (size_t)n * g.ldct + m, 0);
      }
    }
    return;
  }
  if (g.pk1) {
    // tile rows are in gates-packed order: row r = 4*(cell - cell0) + gate, cell0 = m0/4 (a multiple of 16); columns are
    // the k axis of the folded gates operand and the row (cell) axis of the folded d_m operand.  Stage the tile in LDS,
    // then every thread writes 16-byte pieces in DESTINATION order (runs of 1 KB / 256 B).
    constexpr int CLD = 68;
    float *Cs = As;                                    // 64 x 68 floats <= GLDS; the K loop ended with a barrier
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < 2; ni++) {
        const float e[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
#pragma unroll
        for (int r = 0; r < 4; r++) Cs[(wr * 32 + mi * 16 + 4 * kg + r) * CLD + wc * 32 + ni * 16 + i16] = e[r];
      }
    __syncthreads();
    const int C = g.gperm, cell0 = m0 >> 2;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int p = tid + 256 * u;
      {   // gates array: piece = (packed tile pt, chunk half chl, h, lane16 = kgrp*16 + i) -> row pt*16 + i, cols chl*32 + kgrp*8 + h*4
        const int pt = p >> 8, chl = (p >> 7) & 1, h = (p >> 6) & 1, l16 = p & 63, i = l16 & 15;
        const int nl = chl * 32 + (l16 >> 4) * 8 + h * 4, n = n0 + nl;
        const int cell = cell0 + pt * 4 + (i >> 2);
        if (cell < C && n < g.N) {
          const float4 v = *reinterpret_cast<const float4 *>(Cs + (pt * 16 + i) * CLD + nl);
          g.pk1[(((size_t)(cell >> 2) * g.nch1 + (n >> 5)) * 2 + h) * 64 + l16] = v;
        }
      }
      {   // d_m array: piece = (column quad ct, gate, cell quad kq, cq = column % 4) -> 4 cells of one gate at one column
        const int cq = p & 3, kq = (p >> 2) & 3, gate = (p >> 4) & 3, ct = p >> 6;
        const int c = n0 + ct * 4 + cq, cell = cell0 + kq * 4;
        if (g.pk2 && c < g.N && cell < C) {                  // (pk2 null: only the gates-order operand is wanted)
          const float *cp = Cs + (kq * 16 + gate) * CLD + ct * 4 + cq;
          const float4 v = make_float4(cp[0], cp[4 * CLD], cp[8 * CLD], cp[12 * CLD]);
          const int k = gate * C + cell;
          g.pk2[(((size_t)(c >> 2) * g.nch2 + (k >> 7)) * 2 + ((k >> 6) & 1)) * 64 + ((k & 63) >> 2) * 4 + cq] = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
      const int n = n0 + wc * 32 + ni * 16 + i16;
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
      const float e[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = m0 + wr * 32 + mi * 16 + 4 * kg + r;
        if (m >= g.M) continue;
        float *cp = g.Cm + (size_t)m * g.ldc + n;
        float val = e[r] + bv;
        if (g.beta != 0.f) val = g.beta * cold[mi][ni][r] + val;
        *cp = val;
        if (g.C2) g.C2[(size_t)m * g.ldc2 + n] = val;
        if (g.C3 && m >= g.tail0) g.C3[(size_t)(m - g.tail0) * g.N + n] = val;
      }
      if (g.Ct) {               // transposed copy of the raw product (beta == 0, no bias)
        const int m = m0 + wr * 32 + mi * 16 + 4 * kg;
        if (m + 4 <= g.M) *reinterpret_cast<float4 *>(g.Ct + (size_t)n * g.ldct + m) = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
}

template <bool TA, bool TB>
__device__ __forceinline__ void gemm_tile(const GemmJob &g, int m0, int n0, float *As, float *Bs) {
  if (g.vecA && g.vecB) gemm_tile_impl<TA, TB, true>(g, m0, n0, As, Bs);     // block-uniform: the whole K loop is one of two versions
  else gemm_tile_impl<TA, TB, false>(g, m0, n0, As, Bs);
}

// 1-D grid padded to a multiple of 8.  XCD-aware order: workgroup w lands on XCD w % 8 (observed dispatch rule, speed
// only); XCD x gets the contiguous m-major tile range [x*cpx, (x+1)*cpx), so its private L2 keeps a few A row panels and
// the B column panels instead of every XCD streaming every panel from the fabric.
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm(GemmJob g) {
  __shared__ __attribute__((aligned(16))) float As[GLDS];
  __shared__ __attribute__((aligned(16))) float Bs[GLDS];
  const int ntn = (g.N + GT - 1) / GT, nbt = ((g.M + GT - 1) / GT) * ntn;
  const int cpx = (nbt + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * cpx + (int)(blockIdx.x >> 3);
  if (b >= nbt) return;
  if (g.guard && (g.guard[2] | g.guard[6])) return;          // (grid-uniform, before any barrier)
  gemm_tile<TA, TB>(g, (b / ntn) * GT, (b % ntn) * GT, As, Bs);
}

// Split-K variant for short-and-wide products whose 64x64 output tiles cannot fill the chip (AffineTransform
// in_diff: 80 x 512 over K = 16624 is 16 tiles): blockIdx.z owns K slice z and writes its partial tile to
// ws[z][M][N]; k_splitk_reduce sums the slices in fixed order (deterministic, no atomics).
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm_splitk(GemmJob g, int klen, float *ws) {
  __shared__ __attribute__((aligned(16))) float As[GLDS];
  __shared__ __attribute__((aligned(16))) float Bs[GLDS];
  const int k0 = blockIdx.z * klen;
  GemmJob s = g;
  s.K = min(g.K - k0, klen);
  s.A = TA ? g.A + (size_t)k0 * g.lda : g.A + k0;
  s.B = TB ? g.B + k0 : g.B + (size_t)k0 * g.ldb;
  s.vecA = g.vecA && (TA || s.K % 8 == 0);
  s.vecB = g.vecB && (!TB || s.K % 8 == 0);
  s.beta = 0.f; s.bias = nullptr;
  s.Cm = ws + (size_t)blockIdx.z * g.M * g.N; s.ldc = g.N;
  gemm_tile<TA, TB>(s, blockIdx.y * GT, blockIdx.x * GT, As, Bs);
}
struct ReduceArgs {
  const float *ws; int ks, M, N;
  float beta; float *Cm; int ldc;
  const float *bias;
  const float *add; int add_ld;       // C = beta*C + add + bias + sum of slices
  float *C2; int ldc2;                // mirrors, as in GemmJob
  float *C3; int tail0;
  const unsigned *guard = nullptr;    // as in GemmJob
};
__global__ __launch_bounds__(256) void k_splitk_reduce(ReduceArgs a) {
  if (a.guard && (a.guard[2] | a.guard[6])) return;
  const long total = (long)a.M * a.N;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / a.N), n = (int)(i - (long)m * a.N);
    float v = 0.f;
    for (int z0 = 0; z0 < a.ks; z0 += 8) {           // 8 independent loads in flight, summed in slice order
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; j++) t[j] = a.ws[(size_t)min(z0 + j, a.ks - 1) * total + i];
#pragma unroll
      for (int j = 0; j < 8; j++) v += (z0 + j < a.ks) ? t[j] : 0.f;
    }
    if (a.bias) v += a.bias[n];
    if (a.add) v = a.add[(size_t)m * a.add_ld + n] + v;
    float *cp = a.Cm + (size_t)m * a.ldc + n;
    if (a.beta != 0.f) v = a.beta * *cp + v;
    *cp = v;
    if (a.C2) a.C2[(size_t)m * a.ldc2 + n] = v;
    if (a.C3 && m >= a.tail0) a.C3[(size_t)(m - a.tail0) * a.N + n] = v;
  }
}

// Two NN split-K products that share K and the slice length in ONE launch pair (the folded BPTT tail: d_r and in_diff
// both contract dgifo rows over 4C): tiles [0, nb1) belong to job 1, the rest to job 2.
__global__ __launch_bounds__(256) void k_gemm_splitk2(GemmJob g1, GemmJob g2, int nb1, int klen, float *ws1, float *ws2) {
  __shared__ __attribute__((aligned(16))) float As[GLDS];
  __shared__ __attribute__((aligned(16))) float Bs[GLDS];
  const bool second = (int)blockIdx.x >= nb1;
  const GemmJob &g = second ? g2 : g1;
  const int lb = second ? (int)blockIdx.x - nb1 : (int)blockIdx.x;
  const int ntn = (g.N + GT - 1) / GT;
  const int k0 = blockIdx.z * klen;
  GemmJob s = g;
  s.K = min(g.K - k0, klen);
  s.A = g.A + k0;
  s.B = g.B + (size_t)k0 * g.ldb;
  s.vecA = g.vecA && s.K % 8 == 0;
  s.beta = 0.f; s.bias = nullptr;
  s.Cm = (second ? ws2 : ws1) + (size_t)blockIdx.z * g.M * g.N; s.ldc = g.N;
  gemm_tile<false, false>(s, (lb / ntn) * GT, (lb % ntn) * GT, As, Bs);
}
__global__ __launch_bounds__(256) void k_splitk_reduce2(ReduceArgs r1, ReduceArgs r2, int nbr1) {
  const bool second = (int)blockIdx.x >= nbr1;
  const ReduceArgs &a = second ? r2 : r1;
  const int nb = second ? (int)gridDim.x - nbr1 : nbr1;
  const int bid = second ? (int)blockIdx.x - nbr1 : (int)blockIdx.x;
  const long total = (long)a.M * a.N;
  for (long i = bid * 256L + threadIdx.x; i < total; i += (long)nb * 256) {
    const int m = (int)(i / a.N), n = (int)(i - (long)m * a.N);
    float v = 0.f;
    for (int z0 = 0; z0 < a.ks; z0 += 8) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; j++) t[j] = a.ws[(size_t)min(z0 + j, a.ks - 1) * total + i];
#pragma unroll
      for (int j = 0; j < 8; j++) v += (z0 + j < a.ks) ? t[j] : 0.f;
    }
    if (a.add) v = a.add[(size_t)m * a.add_ld + n] + v;
    a.Cm[(size_t)m * a.ldc + n] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// all gradient accumulations of one minibatch in ONE launch (...streams.h:468-487):
//   blocks [0, nb0)        W_gifo_x_corr = beta*corr + DGIFO^T * in
//   blocks [nb0, nb1)      W_gifo_r_corr = beta*corr + DGIFO^T * YR[0..T-1]
//   blocks [nb1, nb2)      W_r_m_corr    = beta*corr + DR^T * YM
//   blocks [nb2, nb3)      bias / peephole column sums (AddRowSumMat, AddDiagMatMat x3)
// ---------------------------------------------------------------------------------------------
struct GradsArgs {
  const unsigned *guard;   // the engine's control words: status of the persistent chain ([2], [6]) or of the one-shot all-reduce ([9]) non-zero -> the minibatch is invalid, touch nothing
  float *mark;             // (or null) data-parallel runs: the validity word that rides at the end of the gradient blob through the all-reduce --
                           // 0 when this rank's gradient is real, 1 when the guard stopped it; the SUM every rank receives gates every rank's Update
  GemmJob wx, wr, wm;
  int nb0, nb1, nb2, nvec;   // tile-id ranges of the three products, then nvec column-sum blocks
  int bf16_narrow;           // k_grads_bf16: 128 x 64 tiles instead of 128 x 128
  int C, S, T;
  const float *dgifo, *cc;
  float beta;
  float *g_bias, *g_pi, *g_pf, *g_po;
  float *p_bias, *p_pi, *p_pf, *p_po;   // parameters to update in the same pass (null: gradient only)
  float lr, clip;
  // k_grads_tm ("tail_merge"): the first nred workgroups add the tail workgroups' partial d_r / in_diff rows (TailReduceJob), the
  // W_r_m tiles wait until tr.ctr[0] has reached tr_target
  TailReduceJob tr; int nred; unsigned tr_target;
};

// bias / peephole column sums of k_grads: block vb covers 64 columns of the 4C gate axis with 4 row groups
__device__ __forceinline__ void grads_column_sums(const GradsArgs &a, int vb, float *lds0, float *lds1) {
  const int C = a.C, S = a.S, rows = a.T * a.S;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = vb * 64 + tx;
  float sb = 0.f, sp = 0.f;
  if (col < 4 * C) {
    const int gate = col / C, cell = col - gate * C;
    // DI/DF[1..T] pair with YC[0..T-1]; DO[1..T] pairs with YC[1..T]
    const float *cbase = a.cc + (gate == 3 ? (size_t)S * C : 0) + cell;
    const float *dbase = a.dgifo + (size_t)S * 4 * C + col;
    // 8 row pairs in flight per thread (a serial loop pays one memory latency per row), summed in row order
    for (int r0 = ty; r0 < rows; r0 += 32) {
      float dv[8], cv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int r = min(r0 + 4 * u, rows - 1);
        dv[u] = dbase[(size_t)r * 4 * C];
        cv[u] = cbase[(size_t)r * C];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const bool on = r0 + 4 * u < rows;
        sb += on ? dv[u] : 0.f;
        if (gate != 0) sp += on ? dv[u] * cv[u] : 0.f;
      }
    }
  }
  float(*rb)[64] = reinterpret_cast<float(*)[64]>(lds0);
  float(*rp)[64] = reinterpret_cast<float(*)[64]>(lds1);
  rb[ty][tx] = sb; rp[ty][tx] = sp;
  __syncthreads();
  if (ty == 0 && col < 4 * C) {
    for (int w = 1; w < 4; w++) { sb += rb[w][tx]; sp += rp[w][tx]; }
    const int gate = col / C, cell = col - gate * C;
    auto fold_update = [&](float c, float *pp) {       // :504-512 for one vector element
      if (a.clip > 0.f) { c = c < -a.clip ? -a.clip : c; c = c > a.clip ? a.clip : c; }
      float pv = *pp;
      pv = pv + (-a.lr) * c;
      *pp = pv;
      return c;
    };
    float cb = (a.beta != 0.f ? a.beta * a.g_bias[col] : 0.f) + sb;
    if (a.p_bias) cb = fold_update(cb, a.p_bias + col);
    a.g_bias[col] = cb;
    float *gp = gate == 1 ? a.g_pi : gate == 2 ? a.g_pf : gate == 3 ? a.g_po : nullptr;
    float *pq = gate == 1 ? a.p_pi : gate == 2 ? a.p_pf : gate == 3 ? a.p_po : nullptr;
    if (gp) {
      float cp = (a.beta != 0.f ? a.beta * gp[cell] : 0.f) + sp;
      if (pq) cp = fold_update(cp, pq + cell);
      gp[cell] = cp;
    }
  }
}

// TM (k_grads_tm, "tail_merge"): the launch also runs the reduction of the tail workgroups' partial rows that used to be k_tail_reduce behind the BPTT
// launch.  Workgroups [0, nred) (dispatched first) add the partial rows -- d_r as write-through stores --, wait for their stores'
// acknowledgements and arrive at tr.ctr[0]; the W_r_m tiles (the only readers of d_r) sit at the END of every XCD's range, wait for
// ctr[0] to reach tr_target (a launch ordinal times nred: nothing to reset, a launch that does nothing still arrives) and read d_r with sc1
// loads; the W_gifo_x / W_gifo_r tiles and the column sums start at once.  The wait is bounded (200 ms; an expiry is counted in ctr[1] and
// the tile goes on -- the reduce workgroups have the lowest indices, so they are resident or done before any tile that waits for them).
template <bool TM>
__device__ __forceinline__ void grads_body(const GradsArgs &a) {
  const bool invalid = a.guard && (a.guard[2] | a.guard[6] | a.guard[9]);
  if (a.mark && blockIdx.x == 0 && threadIdx.x == 0) *a.mark = invalid ? 1.f : 0.f;
  int bi = (int)blockIdx.x;
  if constexpr (TM) {
    if (bi < a.nred) {
      if (!invalid) {
        const int gidx = bi * 256 + (int)threadIdx.x;
        tail_reduce_outputs<true>(a.tr.tws, a.tr.nslots, a.tr.T, a.tr.S, a.tr.R, a.tr.ncols, a.tr.od, a.tr.od_stride, a.tr.dr, a.tr.in_diff,
                                  a.tr.id_stride, gidx >> 3, a.nred * 32, gidx & 7, nullptr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(a.tr.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    bi -= a.nred;
  }
  if (invalid) return;   // a persistent launch of this minibatch gave up: leave momentum and parameters alone
  __shared__ __attribute__((aligned(16))) float As[GLDS];
  __shared__ __attribute__((aligned(16))) float Bs[GLDS];
  // XCD-aware order: workgroup w lands on XCD w % 8 (observed dispatch rule, speed only); XCD x gets the contiguous
  // m-major tile range [x*cpx, (x+1)*cpx), so its private L2 holds a few A row panels and the B column panels
  // instead of streaming every A panel once per XCD.  (TM: two such ranges per XCD -- its share of the tiles that wait for nobody,
  // then its share of the W_r_m tiles and column sums.)
  const int nbt = a.nb2 + a.nvec;
  int b;
  if constexpr (TM) {
    const int cpx1 = (a.nb1 + 7) >> 3, cpx2 = (nbt - a.nb1 + 7) >> 3, x = bi & 7, pos = bi >> 3;
    if (pos < cpx1) { b = x * cpx1 + pos; if (b >= a.nb1) return; }
    else { b = a.nb1 + x * cpx2 + (pos - cpx1); if (b >= nbt) return; }
  } else {
    const int cpx = (nbt + 7) >> 3;
    b = (bi & 7) * cpx + (bi >> 3);
    if (b >= nbt) return;
  }
  if (b < a.nb2) {
    const GemmJob &g = b < a.nb0 ? a.wx : b < a.nb1 ? a.wr : a.wm;
    const int lb = b < a.nb0 ? b : b < a.nb1 ? b - a.nb0 : b - a.nb1;
    const int ntn = (g.N + GT - 1) / GT;
    if constexpr (TM) {
      if (b >= a.nb1) {
        if (threadIdx.x == 0) {
          const long long t0 = wall_clock64();
          while ((int)(__hip_atomic_load(a.tr.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.tr_target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 20000000LL) { __hip_atomic_fetch_add(a.tr.ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          }
        }
        __syncthreads();
        gemm_tile_impl<true, false, true, true>(g, (lb / ntn) * GT, (lb % ntn) * GT, As, Bs);   // (the merged form is only launched with 16-byte rows)
        return;
      }
    }
    gemm_tile<true, false>(g, (lb / ntn) * GT, (lb % ntn) * GT, As, Bs);
    return;
  }
  grads_column_sums(a, b - a.nb2, As, Bs);
}
// Four waves per SIMD (128 registers; two values spill): a 512-input layer has 954 tiles and column-sum blocks -- at three per SIMD 768 are
// resident and the rest wait a whole tile time (every tile is latency-bound: operands, old corr / parameters, stores).  tools/ab_step.py,
// bench.py --config c4 (A-B of builds, twice each): 40-input layer 14.4 -> 13.7 us, 512-input layer 21.3 -> 18.3 us, configs[3] 0.4110 -> 0.4077 ms.
__global__ __launch_bounds__(256, 4) void k_grads(GradsArgs a) { grads_body<false>(a); }
__global__ __launch_bounds__(256) void k_grads_tm(GradsArgs a) { grads_body<true>(a); }   // "tail_merge"

// ---------------------------------------------------------------------------------------------
// bf16 operand mode: the three gradient products on v_mfma_f32_16x16x32_bf16.  128x128 output tiles (the 64x64 fp32
// tile is L2-bandwidth-bound long before the 16x faster bf16 pipe is busy), K tile 64 (the K loop exposes one memory
// latency per tile: 64 KB in flight per workgroup).  Both operands are stored
// [K x X] (fp32 planes); a thread fetches two consecutive k rows x four x columns, rounds to bf16 (RNE) and writes the
// (k, k+1) pairs as 32-bit words into a k-contiguous LDS layout  word[(x&3)*PLANE + (x>>2)*LDQ + k/2]: four planes by
// x mod 4 (the four columns a thread holds go to four planes), LDQ = 36 and PLANE = 16 mod 64 make the 16-byte operand
// reads of 16 rows x 4 k-groups conflict-free and leave the 32-bit stash writes 2-way (a plain [x][k/2] layout is 8-way:
// its writes alone cost more than the MFMAs).  An MFMA operand (8 consecutive k of one row) is one ds_read_b128.
// fp32 accumulate, beta and the column-sum blocks exactly as in k_grads.
// ---------------------------------------------------------------------------------------------
constexpr int GRADS_BF16_MIN_ROWS = 256;
constexpr int BT = 128, BK = 64, LDQ = 36, PLANE = 32 * LDQ + 16;   // see the layout note above

template <int XQ = 32>                                // x quads of the tile (32: 128 columns, 16: 64)
__device__ __forceinline__ void fetch_pair(const float *__restrict__ P, int ld, int X, int K, int x0, int k0, int u,
                                           float4 (&r)[2], int &ok) {
  const int kp = u / XQ, xq = u % XQ;
  const int k = k0 + 2 * kp, x = x0 + 4 * xq;
  const bool xin = x + 4 <= X;                       // X % 4 == 0 on this path
  const float *p0 = P + (size_t)min(k, K - 1) * ld + (xin ? x : 0);
  const float *p1 = P + (size_t)min(k + 1, K - 1) * ld + (xin ? x : 0);
  r[0] = ldg4(p0); r[1] = ldg4(p1);
  ok = (xin && k < K ? 1 : 0) | (xin && k + 1 < K ? 2 : 0);     // applied by stash_pair, one K tile later (see unpack_tile)
}
__device__ __forceinline__ unsigned pack_pair(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 h = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, h);
}
template <int XQ = 32>
__device__ __forceinline__ void stash_pair(unsigned *Ls, int u, const float4 (&rr)[2], int ok) {
  const float4 r[2] = {keep_if(rr[0], (ok & 1) != 0), keep_if(rr[1], (ok & 2) != 0)};
  const int kp = u / XQ, xq = u % XQ;
  unsigned *d = Ls + xq * LDQ + kp;
  d[0] = pack_pair(r[0].x, r[1].x);
  d[PLANE] = pack_pair(r[0].y, r[1].y);
  d[2 * PLANE] = pack_pair(r[0].z, r[1].z);
  d[3 * PLANE] = pack_pair(r[0].w, r[1].w);
}

// NJ = 16-column blocks per wave: 4 = 128 x 128 tiles, 2 = 128 x 64 (twice the workgroups: the K loop exposes a memory latency per
// tile, and 288 tiles of 128 x 128 leave the 256 CUs with one workgroup each)
template <int NJ, bool NO32 = false>                 // NO32: the kernel's fast epilogues are the ones WITHOUT the fp32 transposed copy (4, 5 instead of 1, 2)
__device__ __forceinline__ void gemm_tile_bf16_tn(const GemmJob &g, int m0, int n0, unsigned *As, unsigned *Bs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
  constexpr int NU = (BK / 2) * 32 / 256;            // (k pair, x quad) units per thread and operand
  constexpr int NUB = (BK / 2) * (8 * NJ) / 256;     // ... of B: 32 NJ columns = 8 NJ quads
  float4 ra[NU][2], rb[NUB][2];
  int oa[NU], ob[NUB];
#pragma unroll
  for (int h = 0; h < NU; h++) fetch_pair(g.A, g.lda, g.M, g.K, m0, 0, tid + 256 * h, ra[h], oa[h]);
#pragma unroll
  for (int h = 0; h < NUB; h++) fetch_pair<8 * NJ>(g.B, g.ldb, g.N, g.K, n0, 0, tid + 256 * h, rb[h], ob[h]);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int h = 0; h < NU; h++) stash_pair(As, tid + 256 * h, ra[h], oa[h]);
#pragma unroll
    for (int h = 0; h < NUB; h++) stash_pair<8 * NJ>(Bs, tid + 256 * h, rb[h], ob[h]);
    __syncthreads();
    if (k0 + BK < g.K) {
#pragma unroll
      for (int h = 0; h < NU; h++) fetch_pair(g.A, g.lda, g.M, g.K, m0, k0 + BK, tid + 256 * h, ra[h], oa[h]);
#pragma unroll
      for (int h = 0; h < NUB; h++) fetch_pair<8 * NJ>(g.B, g.ldb, g.N, g.K, n0, k0 + BK, tid + 256 * h, rb[h], ob[h]);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ks++) {
      bf16x8 af[4], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int xa = wr * 64 + i * 16 + i16;
        af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(As + (xa & 3) * PLANE + (xa >> 2) * LDQ + ks * 16 + kg * 4));
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int xb = wc * 16 * NJ + j * 16 + i16;
        bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(Bs + (xb & 3) * PLANE + (xb >> 2) * LDQ + ks * 16 + kg * 4));
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  if (g.P) {
    // Gradient product with momentum and Update folded in (launch_grads + GradsUpdate on the bf16 tiles; N, ldc, ldct multiples of 4,
    // 16-byte aligned blobs) -- the epilogue of gemm_tile_impl on this tile shape: the 128-row tile leaves in FOUR chunks of 32
    // CONSECUTIVE rows (chunk c = blocks i = 2 (c & 1), + 1 of the two waves with wr = c >> 1: what fits the A staging area; the
    // transposed copy then leaves in whole 128-byte lines per column, its bf16 copy in 64-byte runs), each chunk through LDS as 16-byte
    // row pieces: corr = beta*corr + grad (:468-487), clip, theta -= lr*corr (:504-512), the bf16 / fp16 planes of the updated tile,
    // then its transposed copy.  The K loop ended with a barrier: As is free.
    constexpr int BTN = 32 * NJ, CLD = BTN + 4, Q = BTN / 4, NPU = 32 * Q / 256, NPT = BTN * 8 / 256;
    static_assert(32 * CLD <= 4 * PLANE, "the chunk does not fit the A staging area");
    float *Cs = reinterpret_cast<float *>(As);
    // V = 0: any tile.  V = 1, 2, 3: the three jobs of the many-stream bf16 mode on a FULL tile with momentum (W_gifo_r: bf16 plane in
    // its own layout + transposed copy + that copy's bf16 copy; W_gifo_x: transposed copy + its bf16 copy; W_r_m: transposed copy + bf16
    // plane in the copy's layout; V = 4, 5: as 1, 2 WITHOUT the fp32 transposed copy -- only its bf16 copy has a reader while the
    // per-XCD chains run: 17 MB of writes less per layer) with every load and store UNCONDITIONAL -- the compiler then counts what is outstanding and waits for
    // exactly the loads it needs; behind a conditional store it waits for everything, i.e. for the previous chunk's stores to drain
    // (twice per chunk).
    auto epilogue = [&](auto vtag) {
      constexpr int V = decltype(vtag)::value;
      constexpr bool F = V != 0;
      // this thread's pieces of the old corr and parameter rows of a chunk: requested ONE CHUNK AHEAD (chunk 0's before the loop,
      // chunk i + 1's in front of chunk i's arithmetic and stores), so that only the first memory round trip is exposed -- a tile is
      // usually the only one its compute unit has (256-288 tiles on 256 units), nothing else would hide the other three
      float4 oc4[2][NPU], op4[2][NPU];
      auto fetch_old = [&](int i, float4 (&oc)[NPU], float4 (&op)[NPU]) {
#pragma unroll
        for (int u = 0; u < NPU; u++) {
          const int p = tid + 256 * u, lr = p / Q, m = m0 + 32 * i + lr, n = n0 + (p % Q) * 4;
          const size_t off = F || (m < g.M && n + 4 <= g.N) ? (size_t)m * g.ldc + n : 0;
          if (F) oc[u] = *reinterpret_cast<const float4 *>(g.Cm + off);
          else oc[u] = g.beta != 0.f ? *reinterpret_cast<const float4 *>(g.Cm + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          op[u] = *reinterpret_cast<const float4 *>(g.P + off);
        }
      };
      fetch_old(0, oc4[0], op4[0]);
#pragma unroll
      for (int i = 0; i < 4; i++) {                    // (chunk index: rows m0 + 32 i .. + 31)
        if (wr == (i >> 1)) {
#pragma unroll
          for (int b = 0; b < 2; b++)
#pragma unroll
            for (int j = 0; j < NJ; j++) {
              const f32x4 &av = acc[2 * (i & 1) + b][j];
              const float e[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
              for (int r = 0; r < 4; r++) Cs[(b * 16 + 4 * kg + r) * CLD + wc * 16 * NJ + j * 16 + i16] = e[r];
            }
        }
        if (i < 3) fetch_old(i + 1, oc4[(i + 1) & 1], op4[(i + 1) & 1]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NPU; u++) {
          const int p = tid + 256 * u, lr = p / Q, nq = (p % Q) * 4;
          const int m = m0 + 32 * i + lr, n = n0 + nq;
          if (F || (m < g.M && n + 4 <= g.N)) {
            float *cs = Cs + lr * CLD + nq;
            const float4 a4 = *reinterpret_cast<const float4 *>(cs);
            float c[4] = {a4.x, a4.y, a4.z, a4.w};
            if (F || g.beta != 0.f) {
              const float4 o = oc4[i & 1][u];
              c[0] = g.beta * o.x + c[0]; c[1] = g.beta * o.y + c[1]; c[2] = g.beta * o.z + c[2]; c[3] = g.beta * o.w + c[3];
            }
            if (g.clip > 0.f) {
#pragma unroll
              for (int q = 0; q < 4; q++) { c[q] = c[q] < -g.clip ? -g.clip : c[q]; c[q] = c[q] > g.clip ? g.clip : c[q]; }
            }
            *reinterpret_cast<float4 *>(g.Cm + (size_t)m * g.ldc + n) = make_float4(c[0], c[1], c[2], c[3]);
            float4 pv = op4[i & 1][u];
            pv.x = pv.x + (-g.lr) * c[0]; pv.y = pv.y + (-g.lr) * c[1]; pv.z = pv.z + (-g.lr) * c[2]; pv.w = pv.w + (-g.lr) * c[3];
            *reinterpret_cast<float4 *>(g.P + (size_t)m * g.ldc + n) = pv;
            *reinterpret_cast<float4 *>(cs) = pv;
            if (V == 1 || V == 4 || (V == 0 && g.s3 && !g.s3t)) {
              const float v4[4] = {pv.x, pv.y, pv.z, pv.w};
              split_store4(F ? 3 : g.s3mode, v4, g.s3 + (size_t)m * g.ldc + n, g.s3pl);
            }
          }
        }
        __syncthreads();
        if (F || g.Ct) {
#pragma unroll
          for (int u = 0; u < NPT; u++) {
            const int p = tid + 256 * u, nl = p >> 3, lr = (p & 7) * 4;
            const int n = n0 + nl, m = m0 + 32 * i + lr;
            if (F || (n < g.N && m + 4 <= g.M)) {
              const float *cs = Cs + lr * CLD + nl;
              const float v4[4] = {cs[0], cs[CLD], cs[2 * CLD], cs[3 * CLD]};
              if (V <= 3) *reinterpret_cast<float4 *>(g.Ct + (size_t)n * g.ldct + m) = make_float4(v4[0], v4[1], v4[2], v4[3]);
              if (V == 3 || (V == 0 && g.s3 && g.s3t)) split_store4(F ? 3 : g.s3mode, v4, g.s3 + (size_t)n * g.ldct + m, g.s3pl);
              if (V == 1 || V == 2 || V == 4 || V == 5 || (V == 0 && g.cth)) split_store4(3, v4, g.cth + (size_t)n * g.ldct + m, 0);
            }
          }
          __syncthreads();
        }
      }
    };
    // (NO32 is the KERNEL's: full tiles of W_gifo_r / W_gifo_x leave the fp32 transposed copy out, partial tiles -- the generic epilogue --
    //  still write it, which is harmless: the engine treats the whole copy as stale.  A kernel with six epilogues, or a generic one with
    //  a pointer test around that store, sent the register allocator into scratch: 872 bytes, 45 -> 73 us per launch.)
    const bool fast = m0 + BT <= g.M && n0 + BTN <= g.N && g.beta != 0.f && g.Ct && (!g.s3 || g.s3mode == 3);
    if (fast && g.s3 && !g.s3t && g.cth) epilogue(std::integral_constant<int, NO32 ? 4 : 1>());
    else if (fast && !g.s3 && g.cth) epilogue(std::integral_constant<int, NO32 ? 5 : 2>());
    else if (fast && g.s3 && g.s3t && !g.cth) epilogue(std::integral_constant<int, 3>());
    else epilogue(std::integral_constant<int, 0>());
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int n = n0 + wc * 16 * NJ + j * 16 + i16;
      if (n >= g.N) continue;
      const float e[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = m0 + wr * 64 + i * 16 + 4 * kg + r;
        if (m >= g.M) continue;
        float *cp = g.Cm + (size_t)m * g.ldc + n;
        *cp = g.beta != 0.f ? g.beta * *cp + e[r] : e[r];
      }
    }
}

// NT form on the same tiles: C = A B^T (+ bias), A [M x K] and B [N x K] both k-contiguous (the batched x-projection :246
// in bf16 operand mode: x rows against W_gifo_x rows).  A thread fetches 8 consecutive k of one row (two 16-byte loads), rounds
// to bf16 and writes the four (k, k+1) words as ONE 16-byte store into the same LDS layout (LDQ % 4 == 0).
__device__ __forceinline__ void fetch_row8(const float *__restrict__ P, int ld, int X, int K, int x0, int k0, int u,
                                           float4 (&r)[2], int &ok) {
  const int x = x0 + (u >> 3), k = k0 + 8 * (u & 7);
  ok = x < X && k + 8 <= K;                          // K % 8 == 0 on this path
  const float *p = P + (size_t)(ok ? x : 0) * ld + (ok ? k : 0);
  r[0] = ldg4(p); r[1] = ldg4(p + 4);
}
__device__ __forceinline__ void stash_row8(unsigned *Ls, int u, const float4 (&rr)[2], int ok) {
  const float4 r[2] = {keep_if(rr[0], ok != 0), keep_if(rr[1], ok != 0)};
  const int xl = u >> 3;
  uint4 w;
  w.x = pack_pair(r[0].x, r[0].y); w.y = pack_pair(r[0].z, r[0].w);
  w.z = pack_pair(r[1].x, r[1].y); w.w = pack_pair(r[1].z, r[1].w);
  *reinterpret_cast<uint4 *>(Ls + (xl & 3) * PLANE + (xl >> 2) * LDQ + 4 * (u & 7)) = w;
}
// NJ: 16-column blocks per wave (4: 128-column tiles; 2: 64-column tiles -- twice the workgroups for results with few tiles)
// Split K (launch_gemm_bf16_nt_splitk): blockIdx.y = K slice of g.kslice columns (0: the whole K), its partial tile goes to slab
// blockIdx.y of g.Cm ([slices][M x N], dense); k_splitk_reduce adds the slabs in order.
template <int NJ>
__global__ __launch_bounds__(256) void k_gemm_bf16_nt(GemmJob g) {
  if (g.kslice > 0) {
    const int k0 = (int)blockIdx.y * g.kslice;
    g.A += k0; g.B += k0;
    g.K = g.K - k0 < g.kslice ? g.K - k0 : g.kslice;
    g.Cm += (size_t)blockIdx.y * g.M * g.N;
  }
  constexpr int BTN = 32 * NJ;
  __shared__ __attribute__((aligned(16))) unsigned As[4 * PLANE];
  __shared__ __attribute__((aligned(16))) unsigned Bs[4 * PLANE];
  const int ntm = (g.M + BT - 1) / BT, ntn = (g.N + BTN - 1) / BTN, nt = ntm * ntn;
  const int cpx = (nt + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * cpx + (int)(blockIdx.x >> 3);     // XCD x: a contiguous n-major range (A is small, B streams once)
  if (b >= nt) return;
  const int m0 = (b % ntm) * BT, n0 = (b / ntm) * BTN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
  constexpr int NU = BT * (BK / 8) / 256, NUB = BTN * (BK / 8) / 256;   // (row, 8-k group) units per thread: A, B
  float4 ra[NU][2], rb[NUB][2];
  int oa[NU], ob[NUB];
#pragma unroll
  for (int h = 0; h < NU; h++) fetch_row8(g.A, g.lda, g.M, g.K, m0, 0, tid + 256 * h, ra[h], oa[h]);
#pragma unroll
  for (int h = 0; h < NUB; h++) fetch_row8(g.B, g.ldb, g.N, g.K, n0, 0, tid + 256 * h, rb[h], ob[h]);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int h = 0; h < NU; h++) stash_row8(As, tid + 256 * h, ra[h], oa[h]);
#pragma unroll
    for (int h = 0; h < NUB; h++) stash_row8(Bs, tid + 256 * h, rb[h], ob[h]);
    __syncthreads();
    if (k0 + BK < g.K) {
#pragma unroll
      for (int h = 0; h < NU; h++) fetch_row8(g.A, g.lda, g.M, g.K, m0, k0 + BK, tid + 256 * h, ra[h], oa[h]);
#pragma unroll
      for (int h = 0; h < NUB; h++) fetch_row8(g.B, g.ldb, g.N, g.K, n0, k0 + BK, tid + 256 * h, rb[h], ob[h]);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ks++) {
      bf16x8 af[4], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int xa = wr * 64 + i * 16 + i16;
        af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(As + (xa & 3) * PLANE + (xa >> 2) * LDQ + ks * 16 + kg * 4));
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int xb = wc * 16 * NJ + j * 16 + i16;
        bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(Bs + (xb & 3) * PLANE + (xb >> 2) * LDQ + ks * 16 + kg * 4));
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // lane (i16, kg): rows 4kg + (0..3) of block i at column i16 of block j: 16 lanes = 64 contiguous bytes per row
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int n = n0 + wc * 16 * NJ + j * 16 + i16;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float e[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = m0 + wr * 64 + i * 16 + 4 * kg + r;
        if (m >= g.M) continue;
        g.Cm[(size_t)m * g.ldc + n] = e[r] + bv + (g.beta != 0.f ? g.beta * g.Cm[(size_t)m * g.ldc + n] : 0.f);
        if (g.C2) g.C2[(size_t)m * g.ldc2 + n] = e[r] + bv;            // (the batched projection: rr rows, output rows, carried r)
        if (g.C3 && m >= g.tail0) g.C3[(size_t)(m - g.tail0) * g.N + n] = e[r] + bv;
      }
    }
  }
}

// NARROW: 128 x 64 tiles as an instance of their own -- half the accumulators and B staging registers: three waves per SIMD instead of two,
// so that twice the tiles are still all resident (a tile is a chain of dependent memory round trips: residency is what counts)
template <bool NO32, bool NARROW = false>
__global__ __launch_bounds__(256, NARROW ? 3 : 2) void k_grads_bf16(GradsArgs a) {
  const bool invalid = a.guard && (a.guard[2] | a.guard[6] | a.guard[9]);
  if (a.mark && blockIdx.x == 0 && threadIdx.x == 0) *a.mark = invalid ? 1.f : 0.f;
  if (invalid) return;   // a persistent launch of this minibatch gave up: leave momentum and parameters alone
  __shared__ __attribute__((aligned(16))) unsigned As[4 * PLANE];
  __shared__ __attribute__((aligned(16))) unsigned Bs[4 * PLANE];
  const int nbt = a.nb2 + a.nvec;
  const int cpx = (nbt + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * cpx + (int)(blockIdx.x >> 3);
  if (b >= nbt) return;
  if (b < a.nb2) {
    const GemmJob &g = b < a.nb0 ? a.wx : b < a.nb1 ? a.wr : a.wm;
    const int lb = b < a.nb0 ? b : b < a.nb1 ? b - a.nb0 : b - a.nb1;
    if constexpr (NARROW) {
      const int ntn = (g.N + BT / 2 - 1) / (BT / 2);
      gemm_tile_bf16_tn<2, NO32>(g, (lb / ntn) * BT, (lb % ntn) * (BT / 2), As, Bs);
    } else {
      const int ntn = (g.N + BT - 1) / BT;
      gemm_tile_bf16_tn<4, NO32>(g, (lb / ntn) * BT, (lb % ntn) * BT, As, Bs);
    }
    return;
  }
  grads_column_sums(a, b - a.nb2, reinterpret_cast<float *>(As), reinterpret_cast<float *>(Bs));
}

// ---------------------------------------------------------------------------------------------
// Update (:504-512) fused with the refresh of the transposed copies the backward kernels read.
// 32x32 tiles of the three matrices (+ 1024-element chunks of the vector parameters).
// ---------------------------------------------------------------------------------------------
struct UpdArgs {
  const unsigned *guard;   // as in GradsArgs
  const float *mark;       // (or null) the validity word behind the all-reduced gradient blob: non-zero = some rank's gradient of this minibatch was not real
  unsigned *peer_skip;     // (or null) counts the Updates that were left out for that reason
  float *param, *corr;
  const float *grad;        // DP: corr = mmt*corr + grad first
  float mmt, lr, clip;
  int touch;                // 0: pure repack (no corr / param writes)
  // three matrices: offset into blob, rows, cols, transposed destination
  long off[3]; int rows[3], cols[3]; float *dstT[3];
  int tb[3];                // first block of matrix i;  vector blocks start at tb_vec
  int tb_vec;
  long voff, vlen;          // vector parameters (bias + 3 peepholes) are contiguous in the blob
  // optional (vector kernel): the three bf16 planes of the fold operands, written from the same tiles (klstm_fold3.hip):
  // a3 in W_gifo_r's own layout (matrix 1), b3 in W_r_m^T's (matrix 2, the transposed destination's)
  unsigned short *a3, *b3; long a_plane, b_plane; int split_mode;
  unsigned short *dstTh[3];  // (or null; vector kernel) bf16 copies (RNE) of the transposed destinations, same layout
};

__device__ __forceinline__ float upd_elem(const UpdArgs &a, long idx) {
  float p = a.param[idx];
  if (a.touch) {
    float c = a.corr[idx];
    if (a.grad) c = a.mmt * c + a.grad[idx];
    if (a.clip > 0.f) { c = c < -a.clip ? -a.clip : c; c = c > a.clip ? a.clip : c; }
    if (a.grad || a.clip > 0.f) a.corr[idx] = c;
    p = p + (-a.lr) * c;
    a.param[idx] = p;
  }
  return p;
}

__global__ __launch_bounds__(256) void k_update_repack(UpdArgs a) {
  // a persistent launch of this minibatch gave up: leave momentum and parameters alone -- on THIS rank (guard) or on ANY rank (the
  // reduced validity word: every rank leaves this Update out, the replicas stay identical; counted, the same number everywhere)
  const bool own = a.guard && (a.guard[2] | a.guard[6] | a.guard[9]);
  if (own || (a.mark && *a.mark != 0.f)) {
    if (a.mark && a.peer_skip && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.peer_skip, 1u);
    return;
  }
  __shared__ float tile[32][33];
  const int b = blockIdx.x, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (b >= a.tb_vec) {
    const long base = (long)(b - a.tb_vec) * 1024;
    for (int j = threadIdx.x; j < 1024; j += 256)
      if (base + j < a.vlen) (void)upd_elem(a, a.voff + base + j);
    return;
  }
  const int mi = b >= a.tb[2] ? 2 : b >= a.tb[1] ? 1 : 0;
  const int rows = a.rows[mi], cols = a.cols[mi];
  const int lb = b - a.tb[mi];
  const int ntc = (cols + 31) / 32;
  const int by = (lb / ntc) * 32, bx = (lb % ntc) * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    tile[j][tx] = (r < rows && c < cols) ? upd_elem(a, a.off[mi] + (long)r * cols + c) : 0.f;
  }
  __syncthreads();
  float *dst = a.dstT[mi];
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;     // dst[c][r]
    if (dst && c < cols && r < rows) dst[(size_t)c * rows + r] = tile[tx][j];
  }
}

// The same on 64x64 tiles with 16-byte accesses everywhere (cols and rows multiples of 4, 16-byte aligned blobs): every
// thread takes four float4 pieces of a tile row-wise (256 contiguous bytes per tile row), the updated tile goes through LDS
// and leaves transposed as float4 pieces along the row axis.  k_update_repack's 4-byte accesses in 128-byte segments
// reach ~4 TB/s of the 35 MB it moves; this form ~6.
// the three 16-byte loads of one piece, and the arithmetic + stores on them: a tile's four pieces are all requested before the first
// store (a load behind a store waits for it -- the compiler cannot know that the rows do not overlap)
struct UpdPiece { float4 p, c, g; };
__device__ __forceinline__ UpdPiece upd_vec_load(const UpdArgs &a, long idx) {
  UpdPiece q;
  q.p = *reinterpret_cast<const float4 *>(a.param + idx);
  q.c = a.touch ? *reinterpret_cast<const float4 *>(a.corr + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  q.g = a.touch && a.grad ? *reinterpret_cast<const float4 *>(a.grad + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  return q;
}
__device__ __forceinline__ float4 upd_vec_apply(const UpdArgs &a, long idx, const UpdPiece &q) {
  float4 p = q.p;
  if (a.touch) {
    float cv[4] = {q.c.x, q.c.y, q.c.z, q.c.w};
    if (a.grad) {
      const float4 g = q.g;
      cv[0] = a.mmt * cv[0] + g.x; cv[1] = a.mmt * cv[1] + g.y; cv[2] = a.mmt * cv[2] + g.z; cv[3] = a.mmt * cv[3] + g.w;
    }
    if (a.clip > 0.f) {
#pragma unroll
      for (int q_ = 0; q_ < 4; q_++) { cv[q_] = cv[q_] < -a.clip ? -a.clip : cv[q_]; cv[q_] = cv[q_] > a.clip ? a.clip : cv[q_]; }
    }
    if (a.grad || a.clip > 0.f) *reinterpret_cast<float4 *>(a.corr + idx) = make_float4(cv[0], cv[1], cv[2], cv[3]);
    p.x = p.x + (-a.lr) * cv[0]; p.y = p.y + (-a.lr) * cv[1]; p.z = p.z + (-a.lr) * cv[2]; p.w = p.w + (-a.lr) * cv[3];
    *reinterpret_cast<float4 *>(a.param + idx) = p;
  }
  return p;
}

__global__ __launch_bounds__(256) void k_update_repack_v(UpdArgs a) {
  // a persistent launch of this minibatch gave up: leave momentum and parameters alone -- on THIS rank (guard) or on ANY rank (the
  // reduced validity word: every rank leaves this Update out, the replicas stay identical; counted, the same number everywhere)
  const bool own = a.guard && (a.guard[2] | a.guard[6] | a.guard[9]);
  if (own || (a.mark && *a.mark != 0.f)) {
    if (a.mark && a.peer_skip && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.peer_skip, 1u);
    return;
  }
  __shared__ __attribute__((aligned(16))) float tile[64 * 68];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.tb_vec) {
    const long base = (long)(b - a.tb_vec) * 1024;
    for (int j = tid; j < 1024; j += 256)
      if (base + j < a.vlen) (void)upd_elem(a, a.voff + base + j);
    return;
  }
  const int mi = b >= a.tb[2] ? 2 : b >= a.tb[1] ? 1 : 0;
  const int rows = a.rows[mi], cols = a.cols[mi];
  const int lb = b - a.tb[mi];
  const int ntc = (cols + 63) / 64;
  const int by = (lb / ntc) * 64, bx = (lb % ntc) * 64;
  UpdPiece piece[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int p = tid + 256 * u, r = by + (p >> 4), c = bx + (p & 15) * 4;
    piece[u] = upd_vec_load(a, a.off[mi] + (r < rows && c + 4 <= cols ? (long)r * cols + c : 0));
  }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int p = tid + 256 * u, rl = p >> 4, cq = (p & 15) * 4;
    const int r = by + rl, c = bx + cq;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows && c + 4 <= cols) {
      v = upd_vec_apply(a, a.off[mi] + (long)r * cols + c, piece[u]);
      if (mi == 1 && a.a3) {
        const float v4[4] = {v.x, v.y, v.z, v.w};
        split_store4(a.split_mode, v4, a.a3 + (size_t)r * cols + c, a.a_plane);
      }
    }
    *reinterpret_cast<float4 *>(tile + rl * 68 + cq) = v;
  }
  __syncthreads();
  float *dst = a.dstT[mi];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int p = tid + 256 * u, cl = p >> 4, rq = (p & 15) * 4;
    const int c = bx + cl, r = by + rq;       // dst[c][r .. r+3]
    if (c < cols && r + 4 <= rows) {
      const float *tp = tile + rq * 68 + cl;
      const float v4[4] = {tp[0], tp[68], tp[2 * 68], tp[3 * 68]};
      if (dst) *reinterpret_cast<float4 *>(dst + (size_t)c * rows + r) = make_float4(v4[0], v4[1], v4[2], v4[3]);   // (null: only the bf16 copy has a reader)
      if (mi == 2 && a.b3) split_store4(a.split_mode, v4, a.b3 + (size_t)c * rows + r, a.b_plane);
      if (a.dstTh[mi]) split_store4(3, v4, a.dstTh[mi] + (size_t)c * rows + r, 0);
    }
  }
}

// TimeShift::PropagateFnc (standard/nnet/nnet-time-shift.h:42-51): out[dst] = in[clamp(dst + shift, 0, rows-1)]
// (shift == 0 is TransmitComponent's copy, nnet-transmit-component.h:26-33).  One row per blockIdx.y, lane-contiguous.
__global__ void k_time_shift(const float *__restrict__ in, int rows, int cols, int in_stride, float *__restrict__ out,
                             int out_stride, int shift, int vec) {
  const int dst = blockIdx.y;
  int src = dst + shift;
  src = src < 0 ? 0 : src;
  src = src > rows - 1 ? rows - 1 : src;
  const float *ip = in + (size_t)src * in_stride;
  float *op = out + (size_t)dst * out_stride;
  if (vec) {
    for (int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4; c < cols; c += gridDim.x * blockDim.x * 4)
      *reinterpret_cast<float4 *>(op + c) = *reinterpret_cast<const float4 *>(ip + c);
  } else {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) op[c] = ip[c];
  }
}

// Xent::EvalMasked for GENERAL posteriors (google/nnet/nnet-loss.cc:76-142): frame r carries the (pdf, weight) entries
// post_pdf/post_w[post_off[r] .. post_off[r+1]); the reference scatters them into a dense zero matrix (`tgt(t, pdf) += w`,
// so repeated pdfs add up, :86-96) and then works on dense rows.  Here the dense target never exists: one workgroup per
// row writes diff = y*mask, then the first occurrence of every pdf fixes its column to (y - t)*mask (:102-107) and
// contributes -mask*t*log(y) (:122-128) and -mask*t*log(t + 1e-20) (:130-136); the target's arg-max (:113) is found among
// the entries and the implicit zeros with FindRowMaxId's lowest-index tie break.
__global__ __launch_bounds__(256) void k_xent_post_rows(const float *__restrict__ y, int cols, int stride, const int *__restrict__ post_off,
                                                        const int *__restrict__ post_pdf, const float *__restrict__ post_w,
                                                        const float *__restrict__ mask, float *__restrict__ diff, int diff_stride,
                                                        float *__restrict__ row_xent, float *__restrict__ row_ent,
                                                        float *__restrict__ row_correct) {
  __shared__ float smv[4];
  __shared__ int smi[4];
  __shared__ float s_xe[256], s_en[256], s_acc[256];
  __shared__ int s_first[256];
  const int row = blockIdx.x;
  const float *yp = y + (size_t)row * stride;
  float *dp = diff + (size_t)row * diff_stride;
  const float m = mask[row];
  const int e0 = post_off[row], n = post_off[row + 1] - e0;
  float best = -3.4e38f; int bi = 0x7fffffff;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float v = yp[c];
    dp[c] = v * m;                                   // (y - 0) * mask; columns with a target are redone below
    if (v > best) { best = v; bi = c; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {                 // arg-max of the network output, lowest index on ties
    const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { smv[wave] = best; smi[wave] = bi; }
  __syncthreads();                                   // also orders the dense diff pass before the per-entry fix-up
  float xe = 0.f, en = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {       // (rows with more than 256 entries: the tail is handled by the loops below)
    const int pdf = post_pdf[e0 + e];
    float acc = 0.f; bool first = true;
    for (int k = 0; k < n; k++)
      if (post_pdf[e0 + k] == pdf) { acc += post_w[e0 + k]; if (k < e) first = false; }   // same order of additions as the scatter
    if (e < 256) { s_acc[e] = acc; s_first[e] = first ? pdf : -1; }
    if (first) {
      const float v = yp[pdf];
      dp[pdf] = (v - acc) * m;                       // :102-107
      xe -= (logf(v) * acc) * m;                     // :122-128
      en -= (logf(acc + 1e-20f) * acc) * m;          // :130-136
    }
  }
  s_xe[threadIdx.x] = xe; s_en[threadIdx.x] = en;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) if (smv[w] > best || (smv[w] == best && smi[w] < bi)) { best = smv[w]; bi = smi[w]; }
    float sx = 0.f, se = 0.f;
    for (int t = 0; t < 256; t++) { sx += s_xe[t]; se += s_en[t]; }
    // arg-max of the (virtual) dense target row
    const int ne = n < 256 ? n : 256;
    float tb = -3.4e38f; int ti = 0x7fffffff, distinct = 0;
    for (int e = 0; e < ne; e++)
      if (s_first[e] >= 0) { distinct++; if (s_acc[e] > tb || (s_acc[e] == tb && s_first[e] < ti)) { tb = s_acc[e]; ti = s_first[e]; } }
    if (distinct < cols && tb <= 0.f) {              // an implicit zero is (one of) the maxima: lowest index whose value is 0
      int z = 0;
      for (bool again = true; again;) {
        again = false;
        for (int e = 0; e < ne; e++) if (s_first[e] == z && s_acc[e] != 0.f) { z++; again = true; break; }
      }
      if (tb < 0.f || z < ti) ti = z;
    }
    row_xent[row] = sx; row_ent[row] = se;
    row_correct[row] = (m == 1.f && bi == ti) ? 1.f : 0.f;
  }
}


// ---------------------------------------------------------------------------------------------
// output tail of the nnet: Softmax rows, Xent::EvalMasked, and the small vector ops of
// AffineTransform::Update.  One 256-thread workgroup per frame row, lane-contiguous column sweeps.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, float *sm, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(v, o); v = is_max ? fmaxf(v, w) : v + w; }
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  float r = sm[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); w++) r = is_max ? fmaxf(r, sm[w]) : r + sm[w];
  return r;
}
// Softmax::PropagateFnc [UPSTREAM-unvendored nnet-activation.h -> CuMatrix::ApplySoftMaxPerRow]: y = exp(x - max) / sum
__global__ __launch_bounds__(256) void k_softmax_rows(const float *__restrict__ in, int cols, int in_stride,
                                                      float *__restrict__ out, int out_stride) {
  __shared__ float sm[4];
  const float *ip = in + (size_t)blockIdx.x * in_stride;
  float *op = out + (size_t)blockIdx.x * out_stride;
  float mx = -3.4e38f;
  for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, ip[c]);
  mx = block_reduce(mx, sm, true);
  float sum = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) { const float e = expf(ip[c] - mx); op[c] = e; sum += e; }
  sum = block_reduce(sum, sm, false);
  const float inv = 1.f / sum;
  for (int c = threadIdx.x; c < cols; c += 256) op[c] *= inv;
}
// Wide rows (the 16624-way output layer): 1024 threads per row, the whole row in registers (8 x float4 per thread, all
// loads issued up front), one pass over memory: 26 -> ~6 us at 80 x 16624.  Needs cols % 4 == 0, cols <= 32768 and
// 16-byte aligned rows; other shapes use k_softmax_rows.
__global__ __launch_bounds__(1024) void k_softmax_rows_v(const float *__restrict__ in, int cols, int in_stride,
                                                        float *__restrict__ out, int out_stride) {
  __shared__ float sm[16];
  const float4 *ip = reinterpret_cast<const float4 *>(in + (size_t)blockIdx.x * in_stride);
  float4 *op = reinterpret_cast<float4 *>(out + (size_t)blockIdx.x * out_stride);
  const int n4 = cols >> 2;
  float4 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int c = threadIdx.x + 1024 * u;
    const float4 t = ip[c < n4 ? c : 0];
    v[u] = c < n4 ? t : make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
  }
  float mx = -3.4e38f;
#pragma unroll
  for (int u = 0; u < 8; u++) mx = fmaxf(mx, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
  mx = block_reduce(mx, sm, true);
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const bool on = threadIdx.x + 1024 * u < n4;
    v[u].x = on ? expf(v[u].x - mx) : 0.f; v[u].y = on ? expf(v[u].y - mx) : 0.f;
    v[u].z = on ? expf(v[u].z - mx) : 0.f; v[u].w = on ? expf(v[u].w - mx) : 0.f;
    sum += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  sum = block_reduce(sum, sm, false);
  const float inv = 1.f / sum;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int c = threadIdx.x + 1024 * u;
    if (c < n4) op[c] = make_float4(v[u].x * inv, v[u].y * inv, v[u].z * inv, v[u].w * inv);
  }
}
// Xent::EvalMasked (google/nnet/nnet-loss.cc:76-142) for one-hot targets (alignments): per frame row
//   diff = (y - t) * mask (:102-107), row_xent = -mask * log(y[target]) (:122-128),
//   row_correct = mask == 1 && argmax(y) == target (:109-120; first maximum wins like FindRowMaxId).
__global__ __launch_bounds__(256) void k_xent_rows(const float *__restrict__ y, int cols, int stride,
                                                   const int *__restrict__ target, const float *__restrict__ mask,
                                                   float *__restrict__ diff, int diff_stride,
                                                   float *__restrict__ row_xent, float *__restrict__ row_correct) {
  __shared__ float smv[4];
  __shared__ int smi[4];
  const int row = blockIdx.x;
  const float *yp = y + (size_t)row * stride;
  float *dp = diff + (size_t)row * diff_stride;
  const int tgt = target[row];
  const float m = mask[row];
  float best = -3.4e38f; int bi = 0x7fffffff;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float v = yp[c];
    dp[c] = (v - (c == tgt ? 1.f : 0.f)) * m;
    if (v > best) { best = v; bi = c; }
  }
  // argmax with lowest-index tie break
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { smv[wave] = best; smi[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) if (smv[w] > best || (smv[w] == best && smi[w] < bi)) { best = smv[w]; bi = smi[w]; }
    row_xent[row] = (tgt >= 0 && tgt < cols) ? -m * logf(yp[tgt]) : 0.f;
    row_correct[row] = (m == 1.f && bi == tgt) ? 1.f : 0.f;
  }
}
__global__ __launch_bounds__(1024) void k_xent_rows_v(const float *__restrict__ y, int cols, int stride,
                                                     const int *__restrict__ target, const float *__restrict__ mask,
                                                     float *__restrict__ diff, int diff_stride,
                                                     float *__restrict__ row_xent, float *__restrict__ row_correct) {
  __shared__ float smv[16];
  __shared__ int smi[16];
  const int row = blockIdx.x;
  const float4 *yp = reinterpret_cast<const float4 *>(y + (size_t)row * stride);
  float4 *dp = reinterpret_cast<float4 *>(diff + (size_t)row * diff_stride);
  const int tgt = target[row];
  const float m = mask[row];
  const int n4 = cols >> 2;
  float4 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) { const int c = threadIdx.x + 1024 * u; v[u] = yp[c < n4 ? c : 0]; }
  float best = -3.4e38f; int bi = 0x7fffffff;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int c = threadIdx.x + 1024 * u;
    if (c < n4) {
      const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      float d[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int col = 4 * c + j;
        d[j] = (e[j] - (col == tgt ? 1.f : 0.f)) * m;
        if (e[j] > best) { best = e[j]; bi = col; }          // ascending columns per thread: first maximum wins
      }
      dp[c] = make_float4(d[0], d[1], d[2], d[3]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { smv[wave] = best; smi[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++) if (smv[w] > best || (smv[w] == best && smi[w] < bi)) { best = smv[w]; bi = smi[w]; }
    row_xent[row] = (tgt >= 0 && tgt < cols) ? -m * logf(y[(size_t)row * stride + tgt]) : 0.f;
    row_correct[row] = (m == 1.f && bi == tgt) ? 1.f : 0.f;
  }
}
// Softmax and Xent::EvalMasked of a wide row in ONE pass: k_softmax_rows_v's arithmetic, then k_xent_rows_v's on the registers
// (same operations in the same order: the same bits as the pair), the posterior row written only if the caller wants it.
// 80 x 16624: 9.2 + 6.3 us and 21 MB -> one launch, 10.6 MB.
__global__ __launch_bounds__(1024) void k_softmax_xent_rows_v(const float *__restrict__ in, int cols, int in_stride, float *__restrict__ post,
                                                             int post_stride, const int *__restrict__ target, const float *__restrict__ mask,
                                                             float *__restrict__ diff, int diff_stride, float *row_xent,
                                                             float *row_correct, double *totals, unsigned *ticket) {
  __shared__ float sm[16];
  __shared__ float smv[16];
  __shared__ int smi[16];
  __shared__ float s_yt;
  __shared__ int s_last;
  __shared__ double s_tot[3][16];
  const int row = blockIdx.x;
  const float4 *ip = reinterpret_cast<const float4 *>(in + (size_t)row * in_stride);
  float4 *dp = reinterpret_cast<float4 *>(diff + (size_t)row * diff_stride);
  const int tgt = target[row];
  const float m = mask[row];
  const int n4 = cols >> 2;
  float4 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int c = threadIdx.x + 1024 * u;
    const float4 t = ip[c < n4 ? c : 0];
    v[u] = c < n4 ? t : make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
  }
  float mx = -3.4e38f;
#pragma unroll
  for (int u = 0; u < 8; u++) mx = fmaxf(mx, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
  mx = block_reduce(mx, sm, true);
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const bool on = threadIdx.x + 1024 * u < n4;
    v[u].x = on ? expf(v[u].x - mx) : 0.f; v[u].y = on ? expf(v[u].y - mx) : 0.f;
    v[u].z = on ? expf(v[u].z - mx) : 0.f; v[u].w = on ? expf(v[u].w - mx) : 0.f;
    sum += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  sum = block_reduce(sum, sm, false);
  const float inv = 1.f / sum;
  float best = -3.4e38f; int bi = 0x7fffffff;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int c = threadIdx.x + 1024 * u;
    if (c < n4) {
      const float e[4] = {v[u].x * inv, v[u].y * inv, v[u].z * inv, v[u].w * inv};
      if (post) reinterpret_cast<float4 *>(post + (size_t)row * post_stride)[c] = make_float4(e[0], e[1], e[2], e[3]);
      float d[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int col = 4 * c + j;
        d[j] = (e[j] - (col == tgt ? 1.f : 0.f)) * m;
        if (col == tgt) s_yt = e[j];
        if (e[j] > best) { best = e[j]; bi = col; }          // ascending columns per thread: first maximum wins
      }
      dp[c] = make_float4(d[0], d[1], d[2], d[3]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { smv[wave] = best; smi[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++) if (smv[w] > best || (smv[w] == best && smi[w] < bi)) { best = smv[w]; bi = smi[w]; }
    // (write-through, agent scope: the workgroup that takes the last ticket reads them with agent-scope loads)
    __hip_atomic_store(row_xent + row, (tgt >= 0 && tgt < cols) ? -m * logf(s_yt) : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(row_correct + row, (m == 1.f && bi == tgt) ? 1.f : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!totals) return;
  // statistics onto the device totals without a launch of their own: the workgroup that takes the last ticket sees every row's two
  // numbers and adds them up in a fixed order.  Round 6: the two numbers leave as write-through (sc1) stores and the ticket is taken
  // once they are acknowledged (s_waitcnt vmcnt(0)) -- no __threadfence(): an agent-scope release writes back EVERY dirty line of the
  // L2 (here: the 5.3 MB of derivatives the launch has just stored; ~3.5 us and more per workgroup, MI355X_MICROARCH.md) for 8 bytes
  // that matter.  The same store-then-flag idiom as the granules of the persistent chains (klstm_persist_dev.h publish()).
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0;
  for (int r = threadIdx.x; r < (int)gridDim.x; r += 1024) {
    t0 += (double)__hip_atomic_load(row_xent + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t1 += (double)__hip_atomic_load(row_correct + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t2 += (double)mask[r];
  }
  for (int o = 32; o > 0; o >>= 1) { t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
  if (lane == 0) { s_tot[0][wave] = t0; s_tot[1][wave] = t1; s_tot[2][wave] = t2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int w = 0; w < 16; w++) t += s_tot[threadIdx.x][w];
    totals[threadIdx.x] += t;
  }
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (ready for the next launch on this stream)
}
// dst[j] = beta*dst[j] + sum_rows src[row][j]   (AddRowSumMat)
// dst[j] = beta*dst[j] + sum over rows of src[r][j]: 64 columns x 4 row groups per workgroup, 8 loads in flight per thread
// (a serial row loop costs one memory latency per row: 20 us for 80 rows), row groups combined through LDS in fixed order
__global__ __launch_bounds__(256) void k_col_sum(const float *__restrict__ src, int rows, int cols, int stride, float beta,
                                                 float *__restrict__ dst) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  const int jc = j < cols ? j : 0;
  float s = 0.f;
  for (int r0 = ty; r0 < rows; r0 += 32) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) t[u] = src[(size_t)min(r0 + 4 * u, rows - 1) * stride + jc];
#pragma unroll
    for (int u = 0; u < 8; u++) s += (r0 + 4 * u < rows) ? t[u] : 0.f;
  }
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < cols) {
    s = part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx];
    dst[j] = (beta != 0.f ? beta * dst[j] : 0.f) + s;
  }
}
__global__ void k_axpy(float *__restrict__ y, const float *__restrict__ x, float a, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = y[i] + a * x[i];
}

// corr = mmt*corr + grad ; param -= lr*corr  in one pass (the post-all-reduce step of a data-parallel Affine layer)
__global__ void k_sgd_momentum(float *__restrict__ param, float *__restrict__ corr, const float *__restrict__ grad, float mmt,
                               float lr, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float c = mmt * corr[i] + grad[i];
    corr[i] = c;
    param[i] = param[i] + (-lr) * c;
  }
}
__global__ void k_apply_momentum(float *__restrict__ corr, const float *__restrict__ grad, float mmt, long n, const unsigned *guard,
                                 const float *mark) {
  if (guard && (guard[2] | guard[6] | guard[9])) return;          // a persistent launch of this minibatch gave up: its gradient must not reach the momentum buffers
  if (mark && *mark != 0.f) return;                               // (or a peer's: UpdArgs::mark)
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    corr[i] = mmt * corr[i] + grad[i];
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

// ---- variant selection -------------------------------------------------------------------------
// generic kernels (any shape): 16 x 16*NT stream tiles, NT in {1,2,4,8}
static inline int pick_nt(int S) {
  const int tiles = cdiv(S, 16);
  return tiles >= 8 ? 8 : tiles >= 4 ? 4 : tiles >= 2 ? 2 : 1;
}
#define GEN_DISPATCH(KERN, nt, grid, st, pr, args, ...)                                           \
  do {                                                                                            \
    const dim3 _blk(NW * 64);                                                                     \
    if (nt == 1) KLAUNCH((KERN<1 __VA_ARGS__>), grid, _blk, st, pr, args);                        \
    if (nt == 2) KLAUNCH((KERN<2 __VA_ARGS__>), grid, _blk, st, pr, args);                        \
    if (nt == 4) KLAUNCH((KERN<4 __VA_ARGS__>), grid, _blk, st, pr, args);                        \
    KLAUNCH((KERN<8 __VA_ARGS__>), grid, _blk, st, pr, args);                                     \
  } while (0)
// vector kernels: SMALL (4x4x1_16b, S <= 4) with CPW in {1,2,4};  16x16x4 with (NT,CPW) in {(1,1),(1,2),(2,1),(4,1)}
struct VecCfg { bool small; int nt, cpw; };
int g_fat_fine = -1;            // many-stream kernels with half-size row tiles: -1 auto (<= 64 streams), 0 never, 1 always
// measured, fwd+bwd+update us with full-size / half-size row tiles: 512->1024/512 at 32 streams 744 / 653 (gates 6.7 -> 5.3,
// proj 5.7 -> 4.1 us: the full-size tiles leave half the chip without a workgroup); 40/800/512: 32 streams 650 / 561,
// 64: 755 / 710, 128: 1056 / 1102
static inline bool fat_fine(int S) { return g_fat_fine >= 1 || (g_fat_fine < 0 && S <= 64); }
int g_small_nt2 = -1;         // -1 auto, 0 never, 1 always
int g_small_max = 16;           // largest NumStream served by the 4x4x1_16b geometry (4 streams per workgroup, grid.y = S/4);
                                // measured at 40/800/512 (us per minibatch, 4x4x1 vs 16x16x4 tiles): S=8 325 vs 452, S=12 394 vs 460,
                                // S=16 442 vs 474
static inline VecCfg pick_vec(int S, int nch, bool gates = false) {
  VecCfg c;
  const int need = cdiv(nch, NW);
  c.small = S <= g_small_max;
  // two stream groups per workgroup share one fetch of the weight tile: measured at 40/800/512 (g_small_nt2 = 1 forces
  // it everywhere) it only pays for the gates kernel at 9+ streams (7.0 -> 5.7 us at 12; proj/dr/dm lose 0.4-1.0 us)
  if (c.small) { c.nt = (S > 4 && (g_small_nt2 == 1 || (g_small_nt2 < 0 && gates && S > 8))) ? 2 : 1; c.cpw = need <= 1 ? 1 : need == 2 ? 2 : 4; return c; }
  c.nt = S <= 16 ? 1 : S <= 32 ? 2 : 4;
  c.cpw = (c.nt == 1 && need >= 2) ? 2 : 1;
  return c;
}
#define VEC_DISPATCH(KERN, cfg, grid, st, pr, args, ...)                                          \
  do {                                                                                            \
    const dim3 _blk(NW * 64);                                                                     \
    if (cfg.small && cfg.nt == 2) {                                                               \
      if (cfg.cpw == 1) KLAUNCH((KERN<2, 1, true __VA_ARGS__>), grid, _blk, st, pr, args);        \
      if (cfg.cpw == 2) KLAUNCH((KERN<2, 2, true __VA_ARGS__>), grid, _blk, st, pr, args);        \
      KLAUNCH((KERN<2, 4, true __VA_ARGS__>), grid, _blk, st, pr, args);                          \
    }                                                                                             \
    if (cfg.small) {                                                                              \
      if (cfg.cpw == 1) KLAUNCH((KERN<1, 1, true __VA_ARGS__>), grid, _blk, st, pr, args);        \
      if (cfg.cpw == 2) KLAUNCH((KERN<1, 2, true __VA_ARGS__>), grid, _blk, st, pr, args);        \
      KLAUNCH((KERN<1, 4, true __VA_ARGS__>), grid, _blk, st, pr, args);                          \
    }                                                                                             \
    if (cfg.nt == 1 && cfg.cpw == 1) KLAUNCH((KERN<1, 1, false __VA_ARGS__>), grid, _blk, st, pr, args); \
    if (cfg.nt == 1) KLAUNCH((KERN<1, 2, false __VA_ARGS__>), grid, _blk, st, pr, args);          \
    if (cfg.nt == 2) KLAUNCH((KERN<2, 1, false __VA_ARGS__>), grid, _blk, st, pr, args);          \
    KLAUNCH((KERN<4, 1, false __VA_ARGS__>), grid, _blk, st, pr, args);                           \
  } while (0)
#define COMMA ,
static inline dim3 vec_grid(int ntiles, int S, const VecCfg &c, int z = 1) {
  return dim3(ntiles, cdiv(S, (c.small ? 4 : 16) * c.nt), z);
}

hipError_t launch_gates_step(const Dims &d0, const FwdPtrs &p, int t, bool fuse_x, const float *in,
                             int in_stride, hipStream_t st, LaunchProbe pr, bool fold) {
  // fold (t >= 2, S <= small_max): the recurrent operand is m(t-1) and the packed weights are [W_rm | W_x]; the kernel
  // is the same with R := C
  Dims d = d0;
  if (fold) d.R = d0.C;
  GatesArgs a;
  a.C = d.C; a.R = d.R; a.S = d.S; a.I = d.I; a.t = t;
  a.wr = p.wr; a.wx = p.wx; a.bias = p.bias; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh; a.mm = p.mm;
  a.cprev = t == 1 ? p.prev_c : p.cc + (size_t)(t - 1) * d.S * d.C;
  a.rprev = fold ? p.mm + (size_t)(t - 1) * d.S * d.C : t == 1 ? p.prev_r : p.rr + (size_t)(t - 1) * d.S * d.R;
  a.x = fuse_x ? in + (size_t)(t - 1) * d.S * in_stride : nullptr;
  a.x_stride = in_stride;
  a.c_mirror = t == 1 ? p.cc : nullptr;
  a.r_mirror = (t == 1 && !fold) ? p.rr : nullptr;
  a.c_save = t == d.T ? p.next_c : nullptr;          // (the carried state is double-buffered: read prev_*, write next_*)
  const float4 *wpk = fold ? p.pk_fold : p.pk_gates;
  const bool vec = wpk != nullptr && aligned16(p.rr) && aligned16(p.prev_r) && aligned16(p.mm) &&
                   (!fuse_x || (aligned16(in) && in_stride % 4 == 0));
  if (fold && !vec) return hipErrorInvalidValue;      // the engine only folds on the vector path
  if (vec) {
    GatesVArgs va; va.g = a; va.wpk = wpk;
    va.nch_total = cdiv(d.R, KCH) + cdiv(d.I, KCH);
    if (p.fat && d.S > 16) {                                 // 4 row tiles (16 cells) x 2 K splits per workgroup
      if (fat_fine(d.S)) {                       // few stream tiles: 8 cells x 4 K splits per workgroup -> twice the workgroups
        va.gx = cdiv(d.C, 8);
        const dim3 fg(cdiv(va.gx, 8) * 8, cdiv(d.S, FST));
        const bool bigf = (fuse_x ? cdiv(d.R, KCH) + cdiv(d.I, KCH) : cdiv(d.R, KCH)) > 16;
        if (p.bf16) {
          if (fuse_x && bigf) KLAUNCH((k_gates_f<2, 4, 32, true, true>), fg, dim3(NW * 64), st, pr, va);
          if (fuse_x) KLAUNCH((k_gates_f<2, 4, 16, true, true>), fg, dim3(NW * 64), st, pr, va);
          if (bigf) KLAUNCH((k_gates_f<2, 4, 32, false, true>), fg, dim3(NW * 64), st, pr, va);
          KLAUNCH((k_gates_f<2, 4, 16, false, true>), fg, dim3(NW * 64), st, pr, va);
        }
        if (fuse_x && bigf) KLAUNCH((k_gates_f<2, 4, 32, true, false>), fg, dim3(NW * 64), st, pr, va);
        if (fuse_x) KLAUNCH((k_gates_f<2, 4, 16, true, false>), fg, dim3(NW * 64), st, pr, va);
        if (bigf) KLAUNCH((k_gates_f<2, 4, 32, false, false>), fg, dim3(NW * 64), st, pr, va);
        KLAUNCH((k_gates_f<2, 4, 16, false, false>), fg, dim3(NW * 64), st, pr, va);
      }
      va.gx = cdiv(d.C, 16);
      const dim3 fgrid(cdiv(va.gx, 8) * 8, cdiv(d.S, FST));
      const bool big = (fuse_x ? cdiv(d.R, KCH) + cdiv(d.I, KCH) : cdiv(d.R, KCH)) > 16;
      if (p.bf16) {
        if (fuse_x && big) KLAUNCH((k_gates_f<4, 2, 32, true, true>), fgrid, dim3(NW * 64), st, pr, va);
        if (fuse_x) KLAUNCH((k_gates_f<4, 2, 16, true, true>), fgrid, dim3(NW * 64), st, pr, va);
        if (big) KLAUNCH((k_gates_f<4, 2, 32, false, true>), fgrid, dim3(NW * 64), st, pr, va);
        KLAUNCH((k_gates_f<4, 2, 16, false, true>), fgrid, dim3(NW * 64), st, pr, va);
      }
      if (fuse_x && big) KLAUNCH((k_gates_f<4, 2, 32, true, false>), fgrid, dim3(NW * 64), st, pr, va);
      if (fuse_x) KLAUNCH((k_gates_f<4, 2, 16, true, false>), fgrid, dim3(NW * 64), st, pr, va);
      if (big) KLAUNCH((k_gates_f<4, 2, 32, false, false>), fgrid, dim3(NW * 64), st, pr, va);
      KLAUNCH((k_gates_f<4, 2, 16, false, false>), fgrid, dim3(NW * 64), st, pr, va);
    }
    const VecCfg cfg = pick_vec(d.S, cdiv(d.R, KCH) + (fuse_x ? cdiv(d.I, KCH) : 0), true);
    const dim3 grid = vec_grid(cdiv(d.C, 4), d.S, cfg);
    if (p.bf16) {
      if (fuse_x) VEC_DISPATCH(k_gates_v, cfg, grid, st, pr, va, COMMA true COMMA true);
      VEC_DISPATCH(k_gates_v, cfg, grid, st, pr, va, COMMA false COMMA true);
    }
    if (fuse_x) VEC_DISPATCH(k_gates_v, cfg, grid, st, pr, va, COMMA true COMMA false);
    VEC_DISPATCH(k_gates_v, cfg, grid, st, pr, va, COMMA false COMMA false);
  }
  const int nt = pick_nt(d.S);
  const dim3 grid(cdiv(d.C, 4), cdiv(d.S, 16 * nt));
  if (fuse_x) GEN_DISPATCH(k_gates_step, nt, grid, st, pr, a, COMMA true);
  GEN_DISPATCH(k_gates_step, nt, grid, st, pr, a, COMMA false);
}

hipError_t launch_proj_step(const Dims &d, const FwdPtrs &p, int t, float *out, int out_stride,
                            hipStream_t st, LaunchProbe pr) {
  ProjArgs a;
  a.C = d.C; a.R = d.R; a.S = d.S; a.t = t;
  a.wm = p.wm; a.mm = p.mm; a.rr = p.rr; a.out = out; a.out_stride = out_stride;
  a.r_save = t == d.T ? p.next_r : nullptr;
  a.vecOut = aligned16(out) && out_stride % 4 == 0 && d.R % 4 == 0;
  const bool vec = p.pk_proj != nullptr && aligned16(p.mm) && aligned16(p.rr) && aligned16(p.prev_r);
  if (vec) {
    ProjVArgs va; va.g = a; va.wpk = p.pk_proj;
    if (p.fat && d.S > 16) {
      if (fat_fine(d.S)) {                       // one 16-row tile x 8 K splits per workgroup
        va.gx = cdiv(d.R, 16);
        const dim3 fg(cdiv(va.gx, 8) * 8, cdiv(d.S, FST));
        if (p.bf16) {
          if (cdiv(d.C, KCH) > 16) KLAUNCH((k_proj_f<1, 8, 32, true>), fg, dim3(NW * 64), st, pr, va);
          KLAUNCH((k_proj_f<1, 8, 16, true>), fg, dim3(NW * 64), st, pr, va);
        }
        if (cdiv(d.C, KCH) > 16) KLAUNCH((k_proj_f<1, 8, 32, false>), fg, dim3(NW * 64), st, pr, va);
        KLAUNCH((k_proj_f<1, 8, 16, false>), fg, dim3(NW * 64), st, pr, va);
      }
      va.gx = cdiv(d.R, 32);
      const dim3 fgrid(cdiv(va.gx, 8) * 8, cdiv(d.S, FST));
      if (p.bf16) {
        if (cdiv(d.C, KCH) > 16) KLAUNCH((k_proj_f<2, 4, 32, true>), fgrid, dim3(NW * 64), st, pr, va);
        KLAUNCH((k_proj_f<2, 4, 16, true>), fgrid, dim3(NW * 64), st, pr, va);
      }
      if (cdiv(d.C, KCH) > 16) KLAUNCH((k_proj_f<2, 4, 32, false>), fgrid, dim3(NW * 64), st, pr, va);
      KLAUNCH((k_proj_f<2, 4, 16, false>), fgrid, dim3(NW * 64), st, pr, va);
    }
    const VecCfg cfg = pick_vec(d.S, cdiv(d.C, KCH));
    if (p.bf16) VEC_DISPATCH(k_proj_v, cfg, vec_grid(cdiv(d.R, 16), d.S, cfg), st, pr, va, COMMA true);
    VEC_DISPATCH(k_proj_v, cfg, vec_grid(cdiv(d.R, 16), d.S, cfg), st, pr, va, COMMA false);
  }
  const int nt = pick_nt(d.S);
  const dim3 grid(cdiv(d.R, 16), cdiv(d.S, 16 * nt));
  GEN_DISPATCH(k_proj_step, nt, grid, st, pr, a, );
}

void set_small_max(int s) { g_small_max = s; }
void set_small_nt2(int v) { g_small_nt2 = v; }
void set_fat_fine(int v) { g_fat_fine = v; }
int get_small_max() { return g_small_max; }

int dr_split_k(const Dims &d) {
  // enough (R/16 x KS) workgroups to spread the 4C-long contraction over the chip at small S
  const int K = 4 * d.C;
  int ks = KSMAX;
  while (ks > 1 && cdiv(K, ks) < NW * KCH) ks >>= 1;
  return ks;
}

hipError_t launch_dr_step(const Dims &d, const BwdPtrs &p, int t, float *in_diff, int id_stride,
                          hipStream_t st, LaunchProbe pr) {
  DrArgs a;
  a.C = d.C; a.R = d.R; a.I = d.I; a.S = d.S; a.t = t;
  a.wrT = p.wrT; a.wxT = p.wxT; a.dgifo = p.dgifo; a.part = p.dr_part;
  const int K = 4 * d.C;
  const int ks = t == 0 ? 1 : p.ks;
  a.klen = cdiv(cdiv(K, ks), KCH) * KCH;
  a.ntr = t == 0 ? 0 : cdiv(d.R, 16);
  const int ntx = in_diff ? cdiv(d.I, 16) : 0;
  if (t == 0) { a.xpart = in_diff; a.x_ld = id_stride; }        // frame 1, written in place
  else { a.xpart = p.dx_part; a.x_ld = d.I; }
  a.vecX = aligned16(a.xpart) && a.x_ld % 4 == 0 && d.I % 4 == 0;
  if (a.ntr + ntx == 0) return hipSuccess;
  const bool vec = p.pk_dr != nullptr && aligned16(p.dgifo) && aligned16(p.dr_part);
  if (vec) {
    DrVArgs va; va.g = a; va.wpk = p.pk_dr; va.nch_total = cdiv(K, KCH);
    if (p.fat && d.S > 16) {
      va.g.ntr = t == 0 ? 0 : cdiv(d.R, 32);                       // 32-row groups (2 row tiles x 4 K splits)
      const int gx = cdiv(va.g.ntr + (in_diff ? cdiv(d.I, 32) : 0), 8) * 8;
      va.gx = va.g.ntr + (in_diff ? cdiv(d.I, 32) : 0);
      // (a 32-chunk slab was measured slower here: 66 KB of LDS and 16 weight registers per lane cost more in
      //  occupancy than the second staging round trip they save)
      if (fat_fine(d.S)) {                       // 16-row groups (one row tile x 8 K splits)
        va.g.ntr = t == 0 ? 0 : cdiv(d.R, 16);
        va.gx = va.g.ntr + (in_diff ? cdiv(d.I, 16) : 0);
        const int gxf = cdiv(va.gx, 8) * 8;
        if (p.bf16) KLAUNCH((k_dr_f<1, 8, 16, true>), dim3(gxf, cdiv(d.S, FST), ks), dim3(NW * 64), st, pr, va);
        KLAUNCH((k_dr_f<1, 8, 16, false>), dim3(gxf, cdiv(d.S, FST), ks), dim3(NW * 64), st, pr, va);
      }
      if (p.bf16) KLAUNCH((k_dr_f<2, 4, 16, true>), dim3(gx, cdiv(d.S, FST), ks), dim3(NW * 64), st, pr, va);
      KLAUNCH((k_dr_f<2, 4, 16, false>), dim3(gx, cdiv(d.S, FST), ks), dim3(NW * 64), st, pr, va);
    }
    const VecCfg cfg = pick_vec(d.S, cdiv(a.klen, KCH));
    if (p.bf16) VEC_DISPATCH(k_dr_v, cfg, vec_grid(a.ntr + ntx, d.S, cfg, ks), st, pr, va, COMMA true);
    VEC_DISPATCH(k_dr_v, cfg, vec_grid(a.ntr + ntx, d.S, cfg, ks), st, pr, va, COMMA false);
  }
  const int nt = pick_nt(d.S);
  const dim3 grid(a.ntr + ntx, cdiv(d.S, 16 * nt), ks);
  GEN_DISPATCH(k_dr_step, nt, grid, st, pr, a, );
}

hipError_t launch_dm_step(const Dims &d, const BwdPtrs &p, int t, const float *out_diff, int od_stride,
                          float *in_diff, int id_stride, hipStream_t st, LaunchProbe pr) {
  DmArgs a;
  a.C = d.C; a.R = d.R; a.I = d.I; a.S = d.S; a.T = d.T; a.t = t;
  a.wmT = p.wmT; a.pi = p.pi; a.pf = p.pf; a.po = p.po;
  a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh;
  a.dgifo = p.dgifo; a.dc = p.dc; a.dr = p.dr;
  a.part = p.dr_part; a.nslab = (t == d.T) ? 0 : p.ks;
  a.out_diff = out_diff; a.od_stride = od_stride;
  const bool red_x = in_diff != nullptr && t < d.T;
  a.xpart = red_x ? p.dx_part : nullptr;
  a.in_diff = red_x ? in_diff + (size_t)t * d.S * id_stride : nullptr;     // rows of frame t+1
  a.id_stride = id_stride;
  const bool vec = p.pk_dm != nullptr && aligned16(p.dr_part) && aligned16(p.dr) && aligned16(out_diff) &&
                   od_stride % 4 == 0 && aligned16(p.dgifo) && aligned16(p.dc) && aligned16(p.gifo) &&
                   aligned16(p.cc) && aligned16(p.hh) && aligned16(p.pi) && aligned16(p.pf) && aligned16(p.po);
  if (vec) {
    DmVArgs va; va.g = a; va.wpk = p.pk_dm;
    if (p.fat && d.S > 16) {
      if (fat_fine(d.S)) {
        va.gx = cdiv(d.C, 16);
        const dim3 fg(cdiv(va.gx, 8) * 8, cdiv(d.S, FST));
        if (p.bf16) {
          if (cdiv(d.R, KCH) > 16) KLAUNCH((k_dm_f<1, 8, 32, true>), fg, dim3(NW * 64), st, pr, va);
          KLAUNCH((k_dm_f<1, 8, 16, true>), fg, dim3(NW * 64), st, pr, va);
        }
        if (cdiv(d.R, KCH) > 16) KLAUNCH((k_dm_f<1, 8, 32, false>), fg, dim3(NW * 64), st, pr, va);
        KLAUNCH((k_dm_f<1, 8, 16, false>), fg, dim3(NW * 64), st, pr, va);
      }
      va.gx = cdiv(d.C, 32);
      const dim3 fgrid(cdiv(va.gx, 8) * 8, cdiv(d.S, FST));
      if (p.bf16) {
        if (cdiv(d.R, KCH) > 16) KLAUNCH((k_dm_f<2, 4, 32, true>), fgrid, dim3(NW * 64), st, pr, va);
        KLAUNCH((k_dm_f<2, 4, 16, true>), fgrid, dim3(NW * 64), st, pr, va);
      }
      if (cdiv(d.R, KCH) > 16) KLAUNCH((k_dm_f<2, 4, 32, false>), fgrid, dim3(NW * 64), st, pr, va);
      KLAUNCH((k_dm_f<2, 4, 16, false>), fgrid, dim3(NW * 64), st, pr, va);
    }
    const VecCfg cfg = pick_vec(d.S, cdiv(d.R, KCH));
    if (p.bf16) VEC_DISPATCH(k_dm_v, cfg, vec_grid(cdiv(d.C, 16), d.S, cfg), st, pr, va, COMMA true);
    VEC_DISPATCH(k_dm_v, cfg, vec_grid(cdiv(d.C, 16), d.S, cfg), st, pr, va, COMMA false);
  }
  const int nt = pick_nt(d.S);
  const dim3 grid(cdiv(d.C, 16), cdiv(d.S, 16 * nt));
  GEN_DISPATCH(k_dm_step, nt, grid, st, pr, a, );
}

static GemmJob make_job(bool transA, bool transB, int M, int N, int K, const float *A, int lda, const float *B,
                        int ldb, float beta, float *Cm, int ldc, const float *bias);

hipError_t launch_dmf_step(const Dims &d, const BwdPtrs &p, int t, const float *P, hipStream_t st, LaunchProbe pr) {
  DmfArgs a;
  a.C = d.C; a.S = d.S; a.T = d.T; a.t = t;
  a.pi = p.pi; a.pf = p.pf; a.po = p.po; a.gifo = p.gifo; a.cc = p.cc; a.hh = p.hh;
  a.dgifo = p.dgifo; a.dc = p.dc; a.P = P; a.wpk = p.pk_fold;
  a.nch_total = cdiv(4 * d.C, KCH4);
  a.nch = t == d.T ? 0 : a.nch_total;
  const dim3 blk(NW * 64);
  const int need = cdiv(a.nch_total, NW);
  if (d.S > 4) {                               // two stream groups per workgroup share the weight fetch
    const dim3 grid(cdiv(d.C, 4), cdiv(d.S, 8));
    if (need <= 1) KLAUNCH((k_dmf_v<1, 2>), grid, blk, st, pr, a);
    if (need == 2) KLAUNCH((k_dmf_v<2, 2>), grid, blk, st, pr, a);
    KLAUNCH((k_dmf_v<4, 2>), grid, blk, st, pr, a);
  }
  const dim3 grid(cdiv(d.C, 4), cdiv(d.S, 4));
  if (need <= 1) KLAUNCH((k_dmf_v<1, 1>), grid, blk, st, pr, a);
  if (need == 2) KLAUNCH((k_dmf_v<2, 1>), grid, blk, st, pr, a);
  KLAUNCH((k_dmf_v<4, 1>), grid, blk, st, pr, a);
}

// x chunks of the folded gates array (the W_rm chunks are written by the fold product itself)
__global__ __launch_bounds__(256) void k_pack_foldx(const float *__restrict__ wx, float4 *__restrict__ pk, int C, int I, int nch1) {
  const int nchM = (C + KCH - 1) / KCH, nchX = nch1 - nchM;
  const long total = (long)((C + 3) / 4) * nchX * 128;
  for (long id = blockIdx.x * 256L + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int lane = (int)(id & 63), h = (int)((id >> 6) & 1);
    const long tc = id >> 7;
    const int tile = (int)(tc / nchX), chx = (int)(tc - (long)tile * nchX);
    const int i = lane & 15, cell = tile * 4 + (i >> 2), gate = i & 3;
    const int k = chx * KCH + (lane >> 4) * 8 + h * 4;
    float4 v = f4zero();
    if (cell < C && k + 4 <= I) v = ldg4(wx + ((size_t)gate * C + cell) * I + k);
    pk[(((size_t)tile * nch1 + nchM + chx) * 2 + h) * 64 + lane] = v;
  }
}

// W_rm = W_gifo_r [4C x R] * W_r_m [R x C] once per Update, written directly as the two packed operands of the folded
// step kernels (NT form on the transposed copy W_r_m^T [C x R]: both operands k-contiguous); then the x chunks.
// pk_fold[0/1] must have been zero-filled once (padding rows / k tails are never written).
hipError_t launch_fold(const Dims &d, const float *param_blob, const float *wmT, float *pk_fold[2], bool pack_x,
                       hipStream_t st, LaunchProbe pr, LaunchProbe pr2, void *scratch3, LaunchProbe pr3, bool planes_fresh, int mode3) {
  const long o_wr = (long)4 * d.C * d.I;
  GemmJob g = make_job(false, true, 4 * d.C, d.C, d.R, param_blob + o_wr, d.R, wmT, d.R, 0.f, nullptr, d.C, nullptr);
  g.gperm = d.C;
  g.pk1 = reinterpret_cast<float4 *>(pk_fold[0]); g.nch1 = cdiv(d.C, KCH) + cdiv(d.I, KCH);
  g.pk2 = reinterpret_cast<float4 *>(pk_fold[1]); g.nch2 = cdiv(4 * d.C, KCH4);
  const dim3 grid(cdiv(cdiv(d.C, GT) * cdiv(4 * d.C, GT), 8) * 8), block(256);
  auto first = [&]() -> hipError_t {
    if (scratch3 && fold_bf16x3_supported(d, mode3))
      return launch_fold_bf16x3(d, mode3, param_blob + o_wr, wmT, scratch3, pk_fold, g.nch1, g.nch2, st, pr3, pr, planes_fresh);   // klstm_fold3.hip
    if (fold_direct_supported(d)) return launch_fold_direct(d, param_blob + o_wr, wmT, pk_fold, g.nch1, g.nch2, st, pr);   // klstm_fold.hip
    KLAUNCH((k_gemm<false, true>), grid, block, st, pr, g);
  };
  hipError_t err = first();
  if (err != hipSuccess || !pack_x) return err;
  const long nx = (long)cdiv(d.C, 4) * cdiv(d.I, KCH) * 128;
  KLAUNCH(k_pack_foldx, dim3((unsigned)cdiv((int)nx, 256)), block, st, pr2, param_blob, g.pk1, d.C, d.I, g.nch1);
}

// r(1..T) = m(1..T) W_r_m^T (:312) for all frames at once -> rr rows, out rows (:328), last block -> prev_r (:331)
hipError_t launch_rbatch(const Dims &d, const FwdPtrs &p, float *out, int out_stride, float *ws, hipStream_t st,
                         LaunchProbe pr, LaunchProbe pr2, const unsigned *guard) {
  const int M = d.T * d.S;
  int kl = 0;
  const int ks = gemm_splitk_plan(M, d.R, d.C, &kl);
  if (ks > 1 && ws)
    return launch_gemm_splitk(false, true, M, d.R, d.C, p.mm + (size_t)d.S * d.C, d.C, p.wm, d.C, 0.f, p.rr + (size_t)d.S * d.R,
                              d.R, nullptr, ws, ks, kl, st, nullptr, 0, pr, pr2, out, out_stride, p.next_r, M - d.S, guard);
  GemmJob g = make_job(false, true, M, d.R, d.C, p.mm + (size_t)d.S * d.C, d.C, p.wm, d.C, 0.f,
                       p.rr + (size_t)d.S * d.R, d.R, nullptr);
  g.C2 = out; g.ldc2 = out_stride;
  g.C3 = p.next_r; g.tail0 = M - d.S;
  g.guard = guard;
  const dim3 grid(cdiv(cdiv(d.R, GT) * cdiv(M, GT), 8) * 8), block(256);
  KLAUNCH((k_gemm<false, true>), grid, block, st, pr, g);
}

bool pack_supported(const Dims &d) { return d.R % 8 == 0 && d.I % 8 == 0 && d.C % 8 == 0; }
void pack_sizes_fold(const Dims &d, long n4[2]) {
  n4[0] = (long)cdiv(d.C, 16) * 4 * (cdiv(d.C, KCH) + cdiv(d.I, KCH)) * 128;
  n4[1] = (long)cdiv(d.C, 4) * cdiv(4 * d.C, KCH4) * 128;
}
void pack_sizes(const Dims &d, long n4[4]) {
  // gates tiles are 4 cells each; a fat workgroup walks 4 of them, so round up to 16 cells (zero rows)
  n4[0] = (long)cdiv(d.C, 16) * 4 * (cdiv(d.R, KCH) + cdiv(d.I, KCH)) * 128;
  n4[1] = (long)cdiv(d.R, 16) * cdiv(d.C, KCH) * 128;
  n4[2] = (long)(cdiv(d.R, 16) + cdiv(d.I, 16)) * cdiv(4 * d.C, KCH) * 128;
  n4[3] = (long)cdiv(d.C, 16) * cdiv(d.R, KCH) * 128;
}
hipError_t launch_pack(const Dims &d, const float *param_blob, const float *wrT, const float *wmT, const float *wxT,
                       float *pk[4], int mask, bool bf16, hipStream_t st, LaunchProbe pr, float *foldx) {
  PackArgs a;
  a.bf16 = bf16 ? 1 : 0;
  a.foldx = reinterpret_cast<float4 *>(foldx);
  a.nchm_fold = cdiv(d.C, KCH); a.nch_fold = a.nchm_fold + cdiv(d.I, KCH);
  a.C = d.C; a.R = d.R; a.I = d.I;
  const long o_wr = (long)4 * d.C * d.I, o_wm = o_wr + (long)4 * d.C * d.R + 7 * d.C;
  a.wx = param_blob; a.wr = param_blob + o_wr; a.wm = param_blob + o_wm;
  a.wrT = wrT; a.wmT = wmT; a.wxT = wxT;
  pack_sizes(d, a.n4);
  for (int i = 0; i < 4; i++) { if (!(mask & (1 << i))) a.n4[i] = 0; if (bf16) a.n4[i] /= 2; }
  a.nch[0] = cdiv(d.R, KCH) + cdiv(d.I, KCH); a.nch[1] = cdiv(d.C, KCH); a.nch[2] = cdiv(4 * d.C, KCH); a.nch[3] = cdiv(d.R, KCH);
  long total = 0;
  for (int i = 0; i < 4; i++) { a.pk[i] = reinterpret_cast<float4 *>(pk[i]); total += a.n4[i]; }
  if (total == 0) return hipSuccess;
  const long nb = (total + 255) / 256;
  KLAUNCH(k_pack, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), st, pr, a);
}

static GemmJob make_job(bool transA, bool transB, int M, int N, int K, const float *A, int lda, const float *B,
                        int ldb, float beta, float *Cm, int ldc, const float *bias) {
  GemmJob g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.beta = beta;
  g.Cm = Cm; g.ldc = ldc; g.bias = bias;
  g.Ct = nullptr; g.ldct = 0; g.C2 = nullptr; g.ldc2 = 0; g.C3 = nullptr; g.tail0 = 0;
  g.gperm = 0; g.pk1 = nullptr; g.nch1 = 0; g.pk2 = nullptr; g.nch2 = 0;
  g.P = nullptr; g.lr = 0.f; g.clip = 0.f; g.coal = 0; g.s3 = nullptr; g.s3pl = 0; g.s3t = 0; g.s3mode = 1;
  // branch-free 8-wide fetches need aligned rows and a contiguous extent that is a multiple of 8
  g.vecA = aligned16(A) && lda % 4 == 0 && (transA ? M : K) % 8 == 0;
  g.vecB = aligned16(B) && ldb % 4 == 0 && (transB ? K : N) % 8 == 0;
  return g;
}

hipError_t launch_gemm(bool transA, bool transB, int M, int N, int K, const float *A, int lda,
                       const float *B, int ldb, float beta, float *Cm, int ldc, const float *bias,
                       hipStream_t st, LaunchProbe pr) {
  const GemmJob g = make_job(transA, transB, M, N, K, A, lda, B, ldb, beta, Cm, ldc, bias);
  const dim3 grid(cdiv(cdiv(N, GT) * cdiv(M, GT), 8) * 8), block(256);
  if (transA && transB) KLAUNCH((k_gemm<true, true>), grid, block, st, pr, g);
  if (transA && !transB) KLAUNCH((k_gemm<true, false>), grid, block, st, pr, g);
  if (!transA && transB) KLAUNCH((k_gemm<false, true>), grid, block, st, pr, g);
  KLAUNCH((k_gemm<false, false>), grid, block, st, pr, g);
}

// C = A B^T + bias with both operands rounded to bf16 (bf16 operand mode; M >= GRADS_BF16_MIN_ROWS rows, K >= 128 -- at K = 40 the
// 64x64 fp32 tiles are faster: 9.8 vs 12.1 us at 640 x 4096 --, K % 8 == 0, 16-byte aligned rows)
// C = beta C + add + A B^T with K split into ks slices of klen (a multiple of 64): the partial products go to ws ([ks][M x N] floats),
// k_splitk_reduce adds them in slice order.  For few output tiles and a long K (640 x 512 over K = 4096: 40 tiles of 128 x 64 took 80 us).
hipError_t launch_gemm_bf16_nt_splitk(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float beta, float *Cm, int ldc,
                                      const float *add, int add_ld, float *ws, int ks, int klen, hipStream_t st, LaunchProbe pr, LaunchProbe pr2) {
  GemmJob g = make_job(false, true, M, N, K, A, lda, B, ldb, 0.f, ws, N, nullptr);
  g.kslice = klen;
  auto first = [&]() -> hipError_t {
    if (N % 32 == 0) {                                // (128 x 32 tiles: more workgroups per CU to cover the load latency of a K tile)
      const int nt1 = cdiv(M, BT) * cdiv(N, BT / 4);
      KLAUNCH(k_gemm_bf16_nt<1>, dim3(cdiv(nt1, 8) * 8, ks), dim3(256), st, pr, g);
    }
    const int ntm = cdiv(M, BT), ntn = cdiv(N, BT / 2), nt = ntm * ntn;
    KLAUNCH(k_gemm_bf16_nt<2>, dim3(cdiv(nt, 8) * 8, ks), dim3(256), st, pr, g);
  };
  hipError_t err = first();
  if (err != hipSuccess) return err;
  ReduceArgs r;
  r.ws = ws; r.ks = ks; r.M = M; r.N = N; r.beta = beta; r.Cm = Cm; r.ldc = ldc; r.bias = nullptr; r.add = add; r.add_ld = add_ld;
  r.C2 = nullptr; r.ldc2 = 0; r.C3 = nullptr; r.tail0 = 0;
  const long nb = ((long)M * N + 255) / 256;
  KLAUNCH(k_splitk_reduce, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), st, pr2, r);
}
bool gemm_bf16_nt_supported(int M, int K, const float *A, int lda, const float *B, int ldb) {
  return M >= GRADS_BF16_MIN_ROWS && K >= 128 && K % 8 == 0 && lda % 4 == 0 && ldb % 4 == 0 && aligned16(A) && aligned16(B);
}
hipError_t launch_gemm_bf16_nt(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *Cm, int ldc,
                               const float *bias, hipStream_t st, LaunchProbe pr, float *C2, int ldc2, float *C3, int tail0, float beta) {
  GemmJob g = make_job(false, true, M, N, K, A, lda, B, ldb, beta, Cm, ldc, bias);
  g.C2 = C2; g.ldc2 = ldc2; g.C3 = C3; g.tail0 = tail0;
  const dim3 block(256);
  if (cdiv(N, BT / 2) * cdiv(M, BT) < 512 && N % 32 == 0) {   // few 128 x 64 tiles too: 128 x 32 -- every K tile of 64 exposes a memory latency, more
    const dim3 grid(cdiv(cdiv(N, BT / 4) * cdiv(M, BT), 8) * 8);   // workgroups per CU cover it (640 x 1024 over K = 512: 15.9 -> 11 us; 640 x 4096: no change)
    KLAUNCH(k_gemm_bf16_nt<1>, grid, block, st, pr, g);
  }
  if (cdiv(N, BT) * cdiv(M, BT) < 384) {             // few 128 x 128 tiles (640 x 4096: 160 on 256 CUs): 128 x 64
    const dim3 grid(cdiv(cdiv(N, BT / 2) * cdiv(M, BT), 8) * 8);
    KLAUNCH(k_gemm_bf16_nt<2>, grid, block, st, pr, g);
  }
  const dim3 grid(cdiv(cdiv(N, BT) * cdiv(M, BT), 8) * 8);
  KLAUNCH(k_gemm_bf16_nt<4>, grid, block, st, pr, g);
}

// Cm = beta*Cm + A^T B (gradient of a weight matrix, K = frames), then P -= lr*Cm in the same pass (GemmJob::P):
// AffineTransform::Update with the momentum folded into the product like ...streams.h:468-487.  N, ldc multiples of 4,
// 16-byte aligned Cm / P.
hipError_t launch_gemm_tn_update(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float beta, float *Cm,
                                 float *P, int ldc, float lr, hipStream_t st, LaunchProbe pr) {
  GemmJob g = make_job(true, false, M, N, K, A, lda, B, ldb, beta, Cm, ldc, nullptr);
  g.P = P; g.lr = lr; g.clip = 0.f;
  const dim3 grid(cdiv(cdiv(N, GT) * cdiv(M, GT), 8) * 8), block(256);
  KLAUNCH((k_gemm<true, false>), grid, block, st, pr, g);
}

// Cm = beta*Cm + A^T B through the coalesced epilogue (N, ldc % 4 == 0, 16-byte aligned Cm)
hipError_t launch_gemm_tn_coal(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float beta, float *Cm,
                               int ldc, hipStream_t st, LaunchProbe pr) {
  GemmJob g = make_job(true, false, M, N, K, A, lda, B, ldb, beta, Cm, ldc, nullptr);
  g.coal = 1;
  const dim3 grid(cdiv(cdiv(N, GT) * cdiv(M, GT), 8) * 8), block(256);
  KLAUNCH((k_gemm<true, false>), grid, block, st, pr, g);
}

// Split-K plan: worth it when the output tiles cover less than half the chip and K is long.
int gemm_splitk_plan(int M, int N, int K, int *klen) {
  const int tiles = cdiv(M, GT) * cdiv(N, GT);
  if (tiles >= 128 || K <= GK) { *klen = K; return 1; }
  const int ks = cdiv(768, tiles);                   // ~3 workgroups per CU
  const int kl = cdiv(cdiv(K, ks), GK) * GK;         // whole K tiles; a slice is at least one
  *klen = kl;
  return cdiv(K, kl);
}
hipError_t launch_gemm_splitk(bool transA, bool transB, int M, int N, int K, const float *A, int lda, const float *B,
                              int ldb, float beta, float *Cm, int ldc, const float *bias, float *ws, int ks, int klen,
                              hipStream_t st, const float *add, int add_ld, LaunchProbe pr, LaunchProbe pr2, float *C2,
                              int ldc2, float *C3, int tail0, const unsigned *guard) {
  const GemmJob g = make_job(transA, transB, M, N, K, A, lda, B, ldb, 0.f, nullptr, N, nullptr);
  const dim3 grid(cdiv(N, GT), cdiv(M, GT), ks), block(256);
  auto first = [&]() -> hipError_t {
    if (transA && transB) KLAUNCH((k_gemm_splitk<true, true>), grid, block, st, pr, g, klen, ws);
    if (transA) KLAUNCH((k_gemm_splitk<true, false>), grid, block, st, pr, g, klen, ws);
    if (transB) KLAUNCH((k_gemm_splitk<false, true>), grid, block, st, pr, g, klen, ws);
    KLAUNCH((k_gemm_splitk<false, false>), grid, block, st, pr, g, klen, ws);
  };
  hipError_t err = first();
  if (err != hipSuccess) return err;
  ReduceArgs r;
  r.ws = ws; r.ks = ks; r.M = M; r.N = N; r.beta = beta; r.Cm = Cm; r.ldc = ldc; r.bias = bias; r.add = add; r.add_ld = add_ld;
  r.C2 = C2; r.ldc2 = ldc2; r.C3 = C3; r.tail0 = tail0; r.guard = guard;
  const long nb = ((long)M * N + 255) / 256;
  KLAUNCH(k_splitk_reduce, dim3((unsigned)(nb > 2048 ? 2048 : nb)), block, st, pr2, r);
}

// Folded BPTT tail: d_r(1..T) = out_diff + dgifo(2..T+1) W_gifo_r (:391) and, if in_diff != nullptr,
// in_diff = dgifo(1..T) W_gifo_x (:457).  ws: bwd_tail_ws_floats(d) floats.
size_t bwd_tail_ws_floats(const Dims &d) {
  int kl = 0;
  const int ks = gemm_splitk_plan(d.T * d.S, d.R, 4 * d.C, &kl);
  return (size_t)ks * d.T * d.S * (d.R + d.I);
}
hipError_t launch_bwd_tail(const Dims &d, const float *dgifo, const float *wr, const float *wx, const float *out_diff,
                           int od_stride, float *dr, float *in_diff, int id_stride, float *ws, hipStream_t st, LaunchProbe pr,
                           LaunchProbe pr2) {
  const int M = d.T * d.S, K4 = 4 * d.C;
  int kl = 0;
  int ks = gemm_splitk_plan(M, d.R, K4, &kl);
  // few frames: both products on the f16 matrix cores at fp32 accuracy, all rows per wave (klstm_fold.hip k_skinny_nn16; 80 frames,
  // 800/512/512: 16.3 us for the tiled split-K launch -> see docs/DESIGN_rounds_1-4.md 10)
  const int G16 = aligned16(dgifo) && aligned16(wr) && aligned16(wx) ? skinny16_pair_groups(M, d.R, in_diff ? d.I : 0, K4, ks) : 0;
  const GemmJob g1 = make_job(false, false, M, d.R, K4, dgifo + (size_t)2 * d.S * K4, K4, wr, d.R, 0.f, nullptr, d.R, nullptr);
  const GemmJob g2 = make_job(false, false, M, d.I, K4, dgifo + (size_t)d.S * K4, K4, wx, d.I, 0.f, nullptr, d.I, nullptr);
  const int nb1 = cdiv(M, GT) * cdiv(d.R, GT), nb2 = in_diff ? cdiv(M, GT) * cdiv(d.I, GT) : 0;
  float *ws2 = ws + (size_t)ks * M * d.R;
  auto first = [&]() -> hipError_t {
    if (G16 > 0) {
      const int G = G16;
      ks = G;                                            // (ws2 above was placed for the larger tiled plan: still in range)
      return launch_skinny16_pair(M, K4, dgifo + (size_t)2 * d.S * K4, dgifo + (size_t)d.S * K4, K4, wr, d.R, wx, in_diff ? d.I : 0, ws, ws2, G, st, pr);
    }
    KLAUNCH(k_gemm_splitk2, dim3(nb1 + nb2, 1, ks), dim3(256), st, pr, g1, g2, nb1, kl, ws, ws2);
  };
  hipError_t err = first();
  if (err != hipSuccess) return err;
  ReduceArgs r1, r2;
  r1.ws = ws; r1.ks = ks; r1.M = M; r1.N = d.R; r1.beta = 0.f; r1.Cm = dr + (size_t)d.S * d.R; r1.ldc = d.R; r1.bias = nullptr;
  r1.add = out_diff; r1.add_ld = od_stride; r1.C2 = nullptr; r1.ldc2 = 0; r1.C3 = nullptr; r1.tail0 = 0;
  r2 = r1;
  r2.ws = ws2; r2.N = d.I; r2.Cm = in_diff; r2.ldc = id_stride; r2.add = nullptr; r2.add_ld = 0;
  const int nbr1 = cdiv(M * d.R, 256), nbr2 = in_diff ? cdiv(M * d.I, 256) : 0;
  KLAUNCH(k_splitk_reduce2, dim3(nbr1 + nbr2), dim3(256), st, pr2, r1, r2, nbr1);
}

bool grads_bf16_tiles(const Dims &d, bool bf16) { return bf16 && d.T * d.S >= GRADS_BF16_MIN_ROWS; }

hipError_t launch_grads(const Dims &d, const float *dgifo, const float *dr, const float *in, int in_stride,
                        const float *rr, const float *mm, const float *cc, float beta, float *dst,
                        hipStream_t st, LaunchProbe pr, bool bf16, const GradsUpdate *upd, const unsigned *guard, float *mark,
                        const TailReduceJob *tr) {
  const int S = d.S, C = d.C, R = d.R, I = d.I, TS = d.T * d.S;
  const long o_wx = 0, o_wr = (long)4 * C * I, o_b = o_wr + (long)4 * C * R, o_pi = o_b + 4 * C,
             o_pf = o_pi + C, o_po = o_pf + C, o_wm = o_po + C;
  const float *dg1 = dgifo + (size_t)S * 4 * C;                      // DGIFO[1..T]
  GradsArgs a;
  a.guard = guard;
  a.mark = mark;
  a.bf16_narrow = 0;
  a.wx = make_job(true, false, 4 * C, I, TS, dg1, 4 * C, in, in_stride, beta, dst + o_wx, I, nullptr);             // :468
  a.wr = make_job(true, false, 4 * C, R, TS, dg1, 4 * C, rr, R, beta, dst + o_wr, R, nullptr);                      // :471 (YR[0..T-1])
  a.wm = make_job(true, false, R, C, TS, dr + (size_t)S * R, R, mm + (size_t)S * C, C, beta, dst + o_wm, C, nullptr); // :486
  a.nb0 = cdiv(4 * C, GT) * cdiv(I, GT);
  a.nb1 = a.nb0 + cdiv(4 * C, GT) * cdiv(R, GT);
  a.nb2 = a.nb1 + cdiv(R, GT) * cdiv(C, GT);
  a.C = C; a.S = S; a.T = d.T; a.dgifo = dgifo; a.cc = cc; a.beta = beta;
  a.g_bias = dst + o_b; a.g_pi = dst + o_pi; a.g_pf = dst + o_pf; a.g_po = dst + o_po;
  a.p_bias = a.p_pi = a.p_pf = a.p_po = nullptr; a.lr = 0.f; a.clip = 0.f;
  if (upd) {                                          // Update folded into the same pass (fp32 tiles and bf16 tiles alike)
    float *pb = upd->params;
    a.wx.P = pb + o_wx; a.wr.P = pb + o_wr; a.wm.P = pb + o_wm;
    a.wx.lr = a.wr.lr = a.wm.lr = a.lr = upd->lr;
    a.wx.clip = a.wr.clip = a.wm.clip = a.clip = upd->clip;
    a.wx.Ct = upd->wxT; a.wx.ldct = 4 * C;           // [I x 4C]
    a.wr.Ct = upd->wrT; a.wr.ldct = 4 * C;           // [R x 4C]
    a.wm.Ct = upd->wmT; a.wm.ldct = R;               // [C x R]
    if (upd->a3) { a.wr.s3 = upd->a3; a.wr.s3pl = upd->a_plane; a.wr.s3t = 0; a.wr.s3mode = upd->split_mode; }
    if (upd->b3) { a.wm.s3 = upd->b3; a.wm.s3pl = upd->b_plane; a.wm.s3t = 1; a.wm.s3mode = upd->split_mode; }
    a.wx.cth = upd->wxTh; a.wr.cth = upd->wrTh;      // (null: none)
    a.p_bias = pb + o_b; a.p_pi = pb + o_pi; a.p_pf = pb + o_pf; a.p_po = pb + o_po;
  }
  if (!upd && aligned16(dst) && C % 4 == 0 && R % 4 == 0 && I % 4 == 0) a.wx.coal = a.wr.coal = a.wm.coal = 1;
  a.nvec = cdiv(4 * C, 64);
  // below ~256 frames per minibatch the products are write-bound and the 64x64 fp32 tiles are faster (80 frames: 13.4 vs
  // 16.5 us); from there on the bf16 tiles win (640 frames at 512/1024/512: 99 -> 55 us)
  const bool bf_ok = bf16 && TS >= GRADS_BF16_MIN_ROWS && aligned16(dgifo) && aligned16(dr) && aligned16(in) && aligned16(rr) && aligned16(mm) &&
                     in_stride % 4 == 0 && C % 4 == 0 && R % 4 == 0 && I % 4 == 0;
  if (bf_ok && upd && !(aligned16(dst) && aligned16(upd->params) && aligned16(upd->wrT) && aligned16(upd->wmT) && aligned16(upd->wxT)))
    return hipErrorInvalidValue;                      // (the fused epilogue moves 16-byte pieces; the engine's blobs are aligned)
  if (bf_ok && tr) return hipErrorInvalidValue;       // (the merged reduction rides on the fp32 tiles only)
  if (bf_ok) {                                        // 128x128 tiles on the bf16 pipe
    // 128 x 64 tiles while 128 x 128 ones would not even give every CU a workgroup (measured at 1024/512, 640 frames: 192 tiles
    // (40 inputs) 24.8 -> 21.3 us; 288 tiles (512 inputs) 30 -> 33.5 us: stays wide)
    // ... and, round 6, whenever the 128 x 64 tiles are ALL resident at the three waves per SIMD their own kernel instance runs at (768
    // workgroups): every tile is a chain of dependent memory round trips, so half the epilogue per tile with every tile in flight wins --
    // 576 tiles at 512 inputs: 41.9 -> 38-39 us per layer, configs[4] 0.659 -> 0.649 ms (A-B of builds, profiles/r06_grads_bf16_narrow_ab.txt;
    // the measurement above was taken when both widths shared one kernel at two waves per SIMD: 576 tiles in 512 slots)
    const int narrow_tiles = cdiv(4 * C, BT) * (cdiv(I, BT / 2) + cdiv(R, BT / 2)) + cdiv(R, BT) * cdiv(C, BT / 2);
    a.bf16_narrow = (cdiv(4 * C, BT) * (cdiv(I, BT) + cdiv(R, BT)) + cdiv(R, BT) * cdiv(C, BT) < 256 || narrow_tiles + a.nvec <= 768) ? 1 : 0;
    const int btn = a.bf16_narrow ? BT / 2 : BT;
    a.nb0 = cdiv(4 * C, BT) * cdiv(I, btn);
    a.nb1 = a.nb0 + cdiv(4 * C, BT) * cdiv(R, btn);
    a.nb2 = a.nb1 + cdiv(R, BT) * cdiv(C, btn);
    if (a.bf16_narrow) {
      if (upd && upd->no_wT32) KLAUNCH((k_grads_bf16<true, true>), dim3(cdiv(a.nb2 + a.nvec, 8) * 8), dim3(256), st, pr, a);
      KLAUNCH((k_grads_bf16<false, true>), dim3(cdiv(a.nb2 + a.nvec, 8) * 8), dim3(256), st, pr, a);
    }
    if (upd && upd->no_wT32) KLAUNCH(k_grads_bf16<true>, dim3(cdiv(a.nb2 + a.nvec, 8) * 8), dim3(256), st, pr, a);
    KLAUNCH(k_grads_bf16<false>, dim3(cdiv(a.nb2 + a.nvec, 8) * 8), dim3(256), st, pr, a);
  }
  if (tr) {   // the reduction of the tail workgroups' partial rows on the first workgroups of this launch
    if (!(a.wm.vecA && a.wm.vecB) || !tr->ctr || !tr->tws) return hipErrorInvalidValue;   // (16-byte rows: the engine's own buffers)
    a.tr = *tr; a.nred = tail_reduce_blocks(*tr); a.tr_target = tr->seq * (unsigned)a.nred;
    KLAUNCH(k_grads_tm, dim3(a.nred + 8 * (cdiv(a.nb1, 8) + cdiv(a.nb2 + a.nvec - a.nb1, 8))), dim3(256), st, pr, a);
  }
  KLAUNCH(k_grads, dim3(cdiv(a.nb2 + a.nvec, 8) * 8), dim3(256), st, pr, a);
}

bool update_repack_vectorised(const Dims &d, const float *param_blob, const float *corr_blob, const float *grad_blob, const float *wrT,
                              const float *wmT, const float *wxT) {
  return d.C % 4 == 0 && d.R % 4 == 0 && d.I % 4 == 0 && aligned16(param_blob) && aligned16(corr_blob) &&
         (!grad_blob || aligned16(grad_blob)) && aligned16(wrT) && aligned16(wmT) && aligned16(wxT);
}
hipError_t launch_update_repack(const Dims &d, float *param_blob, float *corr_blob, const float *grad_blob,
                                float mmt, float lr, float clip, float *wrT, float *wmT, float *wxT,
                                hipStream_t st, LaunchProbe pr, const unsigned *guard, const GradsUpdate *planes, const float *mark,
                                unsigned *peer_skip) {
  const int C = d.C, R = d.R, I = d.I;
  UpdArgs a;
  a.guard = guard;
  a.mark = mark; a.peer_skip = peer_skip;
  a.a3 = a.b3 = nullptr; a.a_plane = a.b_plane = 0; a.split_mode = 1;
  a.dstTh[0] = a.dstTh[1] = a.dstTh[2] = nullptr;
  a.param = param_blob; a.corr = corr_blob; a.grad = grad_blob; a.mmt = mmt; a.lr = lr; a.clip = clip;
  a.touch = (lr != 0.f || grad_blob != nullptr || clip > 0.f) ? 1 : 0;
  const long o_wr = (long)4 * C * I, o_b = o_wr + (long)4 * C * R, o_wm = o_b + 7 * C;
  a.off[0] = 0;    a.rows[0] = 4 * C; a.cols[0] = I; a.dstT[0] = wxT;
  a.off[1] = o_wr; a.rows[1] = 4 * C; a.cols[1] = R; a.dstT[1] = wrT;
  a.off[2] = o_wm; a.rows[2] = R;     a.cols[2] = C; a.dstT[2] = wmT;
  a.voff = o_b; a.vlen = 7 * C;
  const bool vec = C % 4 == 0 && R % 4 == 0 && I % 4 == 0 && aligned16(param_blob) && aligned16(corr_blob) &&
                   (!grad_blob || aligned16(grad_blob)) && aligned16(wrT) && aligned16(wmT) && aligned16(wxT);
  const int tsz = vec ? 64 : 32;
  int nb = 0;
  for (int i = 0; i < 3; i++) { a.tb[i] = nb; nb += cdiv(a.rows[i], tsz) * cdiv(a.cols[i], tsz); }
  a.tb_vec = nb;
  nb += cdiv(7 * C, 1024);
  if (vec && planes) {
    a.a3 = planes->a3; a.b3 = planes->b3; a.a_plane = planes->a_plane; a.b_plane = planes->b_plane; a.split_mode = planes->split_mode;
    a.dstTh[0] = planes->wxTh; a.dstTh[1] = planes->wrTh;
  }
  if (vec) KLAUNCH(k_update_repack_v, dim3(nb), dim3(256), st, pr, a);
  KLAUNCH(k_update_repack, dim3(nb), dim3(256), st, pr, a);
}

hipError_t launch_time_shift(const float *in, int rows, int cols, int in_stride, float *out, int out_stride, int shift,
                             hipStream_t st, LaunchProbe pr) {
  const int vec = aligned16(in) && aligned16(out) && in_stride % 4 == 0 && out_stride % 4 == 0 && cols % 4 == 0;
  const int per = vec ? 4 : 1;
  KLAUNCH(k_time_shift, dim3(cdiv(cdiv(cols, per), 256), rows), dim3(256), st, pr, in, rows, cols, in_stride, out, out_stride,
          shift, vec);
}

hipError_t launch_softmax(const float *in, int rows, int cols, int in_stride, float *out, int out_stride, hipStream_t st) {
  LaunchProbe pr;
  const bool wide = cols % 4 == 0 && cols <= 32768 && cols >= 2048 && in_stride % 4 == 0 && out_stride % 4 == 0 &&
                    aligned16(in) && aligned16(out);
  if (wide) KLAUNCH(k_softmax_rows_v, dim3(rows), dim3(1024), st, pr, in, cols, in_stride, out, out_stride);
  KLAUNCH(k_softmax_rows, dim3(rows), dim3(256), st, pr, in, cols, in_stride, out, out_stride);
}
hipError_t launch_xent(const float *y, int rows, int cols, int stride, const int *target, const float *mask, float *diff,
                       int diff_stride, float *row_xent, float *row_correct, hipStream_t st) {
  LaunchProbe pr;
  const bool wide = cols % 4 == 0 && cols <= 32768 && cols >= 2048 && stride % 4 == 0 && diff_stride % 4 == 0 &&
                    aligned16(y) && aligned16(diff);
  if (wide) KLAUNCH(k_xent_rows_v, dim3(rows), dim3(1024), st, pr, y, cols, stride, target, mask, diff, diff_stride, row_xent, row_correct);
  KLAUNCH(k_xent_rows, dim3(rows), dim3(256), st, pr, y, cols, stride, target, mask, diff, diff_stride, row_xent, row_correct);
}
// one pass when the row fits the registers of 1024 threads (returns hipErrorNotSupported otherwise: the caller runs the pair)
hipError_t launch_softmax_xent(const float *in, int rows, int cols, int in_stride, float *post, int post_stride, const int *target,
                               const float *mask, float *diff, int diff_stride, float *row_xent, float *row_correct, double *totals,
                               unsigned *ticket, hipStream_t st) {
  LaunchProbe pr;
  const bool wide = cols % 4 == 0 && cols <= 32768 && cols >= 2048 && in_stride % 4 == 0 && diff_stride % 4 == 0 && aligned16(in) &&
                    aligned16(diff) && (!post || (post_stride % 4 == 0 && aligned16(post)));
  if (!wide) return hipErrorNotSupported;
  KLAUNCH(k_softmax_xent_rows_v, dim3(rows), dim3(1024), st, pr, in, cols, in_stride, post, post_stride, target, mask, diff, diff_stride,
          row_xent, row_correct, totals, ticket);
}
hipError_t launch_xent_post(const float *y, int rows, int cols, int stride, const int *post_off, const int *post_pdf, const float *post_w,
                            const float *mask, float *diff, int diff_stride, float *row_xent, float *row_ent, float *row_correct,
                            hipStream_t st) {
  LaunchProbe pr;
  KLAUNCH(k_xent_post_rows, dim3(rows), dim3(256), st, pr, y, cols, stride, post_off, post_pdf, post_w, mask, diff, diff_stride, row_xent,
          row_ent, row_correct);
}
// Loss statistics of a minibatch onto device totals (Xent::EvalMasked adds its three scalars to loss_, correct_, frames_ on the host,
// nnet-loss.cc:136-141; a trainer that reports every N minibatches reads the totals once): totals[0] += sum row_xent (double),
// totals[1] += sum row_correct, totals[2] += sum mask.  One workgroup, fixed summation order.
__global__ __launch_bounds__(256) void k_xent_accumulate(const float *__restrict__ row_xent, const float *__restrict__ row_correct,
                                                          const float *__restrict__ mask, int rows, double *totals) {
  __shared__ double red[3][256];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) { s0 += (double)row_xent[r]; s1 += (double)row_correct[r]; s2 += (double)mask[r]; }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int q = 0; q < 3; q++) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 3) totals[threadIdx.x] += red[threadIdx.x][0];
}
hipError_t launch_xent_accumulate(const float *row_xent, const float *row_correct, const float *mask, int rows, double *totals, hipStream_t st) {
  LaunchProbe pr;
  KLAUNCH(k_xent_accumulate, dim3(1), dim3(256), st, pr, row_xent, row_correct, mask, rows, totals);
}
hipError_t launch_col_sum(const float *src, int rows, int cols, int stride, float beta, float *dst, hipStream_t st) {
  LaunchProbe pr;
  KLAUNCH(k_col_sum, dim3(cdiv(cols, 64)), dim3(256), st, pr, src, rows, cols, stride, beta, dst);
}
static inline int ew_grid(long n);
hipError_t launch_axpy(float *y, const float *x, float a, long n, hipStream_t st) {
  LaunchProbe pr;
  KLAUNCH(k_axpy, dim3(ew_grid(n)), dim3(256), st, pr, y, x, a, n);
}

static inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g)); }

hipError_t launch_sgd_momentum(float *param, float *corr, const float *grad, float mmt, float lr, long n, hipStream_t st) {
  LaunchProbe pr;
  KLAUNCH(k_sgd_momentum, dim3(ew_grid(n)), dim3(256), st, pr, param, corr, grad, mmt, lr, n);
}
hipError_t launch_apply_momentum(float *corr, const float *grad, float mmt, long n, hipStream_t st, LaunchProbe pr, const unsigned *guard,
                                 const float *mark) {
  KLAUNCH(k_apply_momentum, dim3(ew_grid(n)), dim3(256), st, pr, corr, grad, mmt, n, guard, mark);
}

}  // namespace klstm
