// kaldi-lstm_amd/csrc/klstm_outer.hip -- gradient of a wide AffineTransform at few frames: G = out_diff^T in  (M = out_dim rows,
// N = in_dim columns, contraction over K = frames of the minibatch, 80 at 4 streams x 20 frames): AffineTransform::Update of the
// output layer 512 -> 16624 ([UPSTREAM] nnet-affine-transform.h, not vendored in the reference -- include/klstm.h, output tail:
// linearity_corr = mmt * linearity_corr + out_diff^T in, bias_corr = mmt * bias_corr + column sums of out_diff).
//
// The product is 1.36 GFLOP writing 34 MB; the 64 x 64-tile fp32 kernel (klstm_kernels.hip k_gemm<true, false>) took 27 us for it
// (+ 3 us for the column sums in a launch of their own): 2080 tiles whose 80-deep contraction is over before the loads of the tile
// are amortised, and 10 us of fp32 MFMA issue on top.
//
// k_outer16: the product on the f16 matrix cores at fp32 accuracy (klstm_math.h f16_split2_pair: x = h1 + h2 / 2048, both normal
// fp16 down to |x| = 2^-14; three products a1 b1 + (a1 b2 + a2 b1) / 2048 with the cross terms in their own accumulators, as the
// fold product, klstm_fold3.hip; dropped a2 b2 ~ 2^-22 relative), both operands split in registers, no LDS in the product, no
// workspace.  Both operands are stored k-major ([frame][column]), the MFMA wants 8 consecutive k of one row per lane: lane
// (i16, kg) loads the float4 at columns 4 i16 .. 4 i16 + 3 of frames 32c + 8 kg + e, e = 0..7 (every load instruction = four
// 256-byte runs), and the e-th components of the eight registers ARE the operands of the four "virtual" 16-row blocks
// {4 i + cm : i = 0..15}, cm = 0..3 -- the transpose costs nothing.  With the same assignment on the n side the four accumulators
// of a lane are one 16-byte piece of a row of G and sixteen lanes write 256 contiguous bytes.
// A workgroup = one strip of 64 rows of G, one wave per SIMD (16624 rows: 256 strips in ONE round of the chip, the 240 rows past them
// ride along one per workgroup on the vector ALU -- see the kernel); each of the
// four waves splits the strip's 64 columns of out_diff ONCE (all K <= 96 frames, kept as fp16 planes in registers: 32 per
// chunk of 32 frames) and walks its 64-column tiles of `in` (two at N = 512) with the raw rows of the next tile in flight
// under the 144 MFMAs of the current one.  Wave 0 also sums the strip's columns of out_diff (the bias gradient: no second launch).
// The epilogue is the store (gradient), or  corr = mmt * corr + G; W -= lr * corr; bias -= lr_b * bias_corr  (Update in the same
// pass: template flag UPD, the tile's rows of corr and W brought in by LDS-DMA under its MFMAs).
// 80 x 16624 x 512: 29.9 -> 16.4 us (gradient), 42.1 -> 33.2 us (Update); DESIGN.md 4f has the ablations and dead ends.
// (First version, kept in the history: planes written k-contiguous by a prep launch, 16-byte operand loads from them: 8.9 + 26 us
//  -- every 128-byte line of the planes fetched twice through a 32 KB L1, 200 MB of operand ingest.)
// Range: every column of out_diff is scaled by its own power of two before the split (see the kernel), `in` is split as it is; an
// entry beyond the fp16 range (|x| >= 65520) makes the wave's accumulators non-finite, which the wave notices and answers by
// recomputing its tile in plain fp32 (klstm_math.h "range guard") -- the launcher then keeps this product on the fp32 tile kernel.
#include "klstm_kernels.h"
#include "klstm_math.h"
#include <hip/hip_ext.h>
#include <type_traits>

namespace klstm {

typedef float of32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 of16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void *outer_gptr;
typedef __attribute__((address_space(3))) void *outer_lptr;

struct OuterArgs {
  const float *diff; int ldd;          // [K][M]
  const float *x; int ldx;             // [K][N]
  int K, M, N;
  int m_main, nextra;                  // strips cover rows [0, m_main); rows m_main .. m_main + nextra - 1 go out one per workgroup (VALU)
  float beta_b; float *bias;           // bias = beta_b * bias + column sums of diff
  float *bias_p; float lr_b;           // bias_p -= lr_b * bias  (nullptr: not)
  float beta; float *Cm; int ldc;      // Cm = beta * Cm + G
  float *P; float lr;                  // P -= lr * Cm  (nullptr: gradient only)
  unsigned *redo;                      // range guard: host-mapped event counter (klstm_kernels.h REDO_OUTER), or null
};

__device__ __forceinline__ of32x4 outer_keep(of32x4 v, bool on) { return on ? v : (of32x4){0.f, 0.f, 0.f, 0.f}; }

// eight rows (frames) x four columns of raw values -> per column cm the eight frames as two fp16 planes
__device__ __forceinline__ void outer_split(const of32x4 (&raw)[8], of16x8 (&p1)[4], of16x8 (&p2)[4]) {
#pragma unroll
  for (int cm = 0; cm < 4; cm++) {
    uint4 u1, u2;
    f16_split2_pair(raw[0][cm], raw[1][cm], u1.x, u2.x);
    f16_split2_pair(raw[2][cm], raw[3][cm], u1.y, u2.y);
    f16_split2_pair(raw[4][cm], raw[5][cm], u1.z, u2.z);
    f16_split2_pair(raw[6][cm], raw[7][cm], u1.w, u2.w);
    p1[cm] = __builtin_bit_cast(of16x8, u1); p2[cm] = __builtin_bit_cast(of16x8, u2);
  }
}

// UPD: the epilogue is AffineTransform::Update.  The tile's 2 x 16 rows of corr and W (32 KB per wave) are requested when the tile
// starts, by LDS-DMA (global_load_lds_dwordx4: no registers, lane-linear destination = each lane later reads back its own 16 bytes),
// and have landed under the tile's MFMAs: 128 KB of dynamic LDS per workgroup.
template <int NCH, bool UPD>
__global__ __launch_bounds__(256) void k_outer16(OuterArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kg = lane >> 4;
  const int m0 = (int)blockIdx.x * 64;
  const int mc = m0 + 4 * i16;                           // this lane's four columns of out_diff (rows of G); M % 4 == 0
  const bool m_in = mc < a.m_main;
  // ---- a row past the strips (16624 = 256 strips of 64 + 240 rows: a 257th..260th workgroup would run alone behind the other 256):
  //      workgroup b also produces row m_main + b, on the vector ALU in exact fp32, from the rows of `in` its waves hold for the
  //      MFMA operands anyway: lane (i16, kg) multiplies its eight frames of a chunk with that row's eight values of out_diff,
  //      the four frame groups meet through two lane exchanges ----
  const bool has_x = (int)blockIdx.x < a.nextra;
  const int mx = has_x ? a.m_main + (int)blockIdx.x : 0;
  __shared__ __attribute__((aligned(16))) float xd[32 * NCH];
  extern __shared__ __attribute__((aligned(16))) char outer_rows[];   // UPD: [4 waves][corr | W][16 rows][64 lanes x 16 bytes]
  char *wrows = outer_rows + wave * 32768;
  constexpr int RB = UPD ? 1 : NCH;                      // (UPD: the row addresses of the tile live across its MFMAs: a one-chunk lead of `in`)
  of32x4 rawb[RB][8];
  auto loadChunk = [&](int j, auto, of32x4 (&dst)[8], int c) {
    const int nc = 64 * (wave + 4 * j) + 4 * i16;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int k = 32 * c + 8 * kg + e;
      dst[e] = *reinterpret_cast<const of32x4 *>(a.x + (size_t)(k < a.K ? k : 0) * a.ldx + (nc < a.N ? nc : 0));
    }
  };
  auto loadTile = [&](int j) {
#pragma unroll
    for (int c = 0; c < RB; c++) loadChunk(j, std::integral_constant<int, 0>(), rawb[c], c);
  };
  // ---- the strip's columns of out_diff: all chunks, split once ----
  of16x8 a1[NCH][4], a2[NCH][4];
  of32x4 colsum = {0.f, 0.f, 0.f, 0.f};
  __shared__ float cinv_s[64];                           // per column of the strip: the inverse of the power of two it was scaled by
  {
    of32x4 raw[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int k = 32 * c + 8 * kg + e;
        raw[c][e] = *reinterpret_cast<const of32x4 *>(a.diff + (size_t)(k < a.K ? k : 0) * a.ldd + (m_in ? mc : 0));
      }
    if (tid < 32 * NCH) xd[tid] = has_x && tid < a.K ? a.diff[(size_t)tid * a.ldd + mx] : 0.f;   // (LDS: 24 registers less per lane)
    loadTile(0);
    of32x4 mx = {0.f, 0.f, 0.f, 0.f}, cscale;
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        raw[c][e] = outer_keep(raw[c][e], m_in && 32 * c + 8 * kg + e < a.K);
        colsum += raw[c][e];
        mx = __builtin_elementwise_max(mx, __builtin_elementwise_abs(raw[c][e]));
      }
    // out_diff is a derivative: late in training its entries are 1e-7 and smaller, where two fp16 planes keep 2^-35 absolute
    // (3e-4 relative).  Every COLUMN of out_diff (= row of G) is brought to [2^11, 2^12) by its own power of two before the split
    // (exact) and the row of G is multiplied back in the epilogue: 22 bits for every entry within 2^25 of its column's largest.
#pragma unroll
    for (int q = 0; q < 4; q++) {
      float m = mx[q];
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      const int eb = (int)((__float_as_uint(m) >> 23) & 0xffu);          // biased exponent of the column's largest entry
      const int sb = eb == 0 ? 127 : min(265 - eb, 253);               // scale = 2^(sb - 127): largest entry -> [2^11, 2^12)
      cscale[q] = __uint_as_float((unsigned)sb << 23);
      if (wave == 0 && kg == 0) cinv_s[4 * i16 + q] = __uint_as_float((unsigned)(254 - sb) << 23);   // (every wave holds the same 64 columns; read behind the barrier below)
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
#pragma unroll
      for (int e = 0; e < 8; e++) raw[c][e] *= cscale;
      outer_split(raw[c], a1[c], a2[c]);
    }
  }
  if (wave == 0 && a.bias) {                             // the four k-groups of a column sit in lanes i16 + 16 kg
    of32x4 v = colsum;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      v[q] += __shfl_xor(v[q], 16);
      v[q] += __shfl_xor(v[q], 32);
    }
    if (kg == 0 && m_in) {
      float4 *bp = reinterpret_cast<float4 *>(a.bias + mc);
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (a.beta_b != 0.f) { const float4 b = *bp; o.x += a.beta_b * b.x; o.y += a.beta_b * b.y; o.z += a.beta_b * b.z; o.w += a.beta_b * b.w; }
      *bp = o;
      if (a.bias_p) {
        float4 *pp = reinterpret_cast<float4 *>(a.bias_p + mc);
        float4 p = *pp;
        p.x -= a.lr_b * o.x; p.y -= a.lr_b * o.y; p.z -= a.lr_b * o.z; p.w -= a.lr_b * o.w;
        *pp = p;
      }
    }
  }
  __syncthreads();                                       // xd
  if (wave == 0 && has_x && a.bias) {
    float bs = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++)
      if (lane < 32) bs += xd[32 * c + lane];
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) bs += __shfl_xor(bs, o);
    if (lane == 0) {
      const float o = (a.beta_b != 0.f ? a.beta_b * a.bias[mx] : 0.f) + bs;
      a.bias[mx] = o;
      if (a.bias_p) a.bias_p[mx] -= a.lr_b * o;
    }
  }
  // ---- this wave's 64-column tiles of `in`: the raw rows of the NEXT tile in flight under the 48 NCH MFMAs of the current one ----
  const int ntile = (a.N + 63) / 64, nmine = wave < ntile ? (ntile - wave + 3) / 4 : 0;
  if (nmine == 0) return;
#pragma unroll 1
  for (int j = 0; j < nmine; j++) {
    const int nc = 64 * (wave + 4 * j) + 4 * i16;
    if (UPD && nc < a.N) {
      asm volatile("" ::: "memory");                     // (the previous tile's rows have been read back)
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int m = min(m0 + 16 * kg + 4 * r + mi, a.m_main - 1);
          __builtin_amdgcn_global_load_lds((outer_gptr)(a.Cm + (size_t)m * a.ldc + nc), (outer_lptr)(wrows + (mi * 4 + r) * 1024), 16, 0, 0);
          if (a.P)
            __builtin_amdgcn_global_load_lds((outer_gptr)(a.P + (size_t)m * a.ldc + nc), (outer_lptr)(wrows + 16384 + (mi * 4 + r) * 1024), 16, 0, 0);
        }
    }
    of32x4 acc[4][4], accx[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
      for (int cn = 0; cn < 4; cn++) { acc[mi][cn] = (of32x4){0, 0, 0, 0}; accx[mi][cn] = (of32x4){0, 0, 0, 0}; }
    of32x4 xg = {0.f, 0.f, 0.f, 0.f};
    const int jn = j + 1 < nmine ? j + 1 : j;            // (unconditional: past the end the same rows again, never used)
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      of16x8 b1[4], b2[4];
      const int cb = UPD ? 0 : c;                        // (a constant once the loop is unrolled)
#pragma unroll
      for (int e = 0; e < 8; e++) rawb[cb][e] = outer_keep(rawb[cb][e], nc < a.N && 32 * c + 8 * kg + e < a.K);
      outer_split(rawb[cb], b1, b2);
      if (has_x) {
        const of32x4 d0 = *reinterpret_cast<const of32x4 *>(&xd[32 * c + 8 * kg]), d1 = *reinterpret_cast<const of32x4 *>(&xd[32 * c + 8 * kg + 4]);
#pragma unroll
        for (int e = 0; e < 4; e++) xg += d0[e] * rawb[cb][e] + d1[e] * rawb[cb][e + 4];
      }
      if (UPD) loadChunk(c + 1 < NCH ? j : jn, std::integral_constant<int, 0>(), rawb[0], c + 1 < NCH ? c + 1 : 0);
      else loadChunk(jn, std::integral_constant<int, 0>(), rawb[cb], c);
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int cn = 0; cn < 4; cn++) {
          accx[mi][cn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[c][mi], b2[cn], accx[mi][cn], 0, 0, 0);
          accx[mi][cn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[c][mi], b1[cn], accx[mi][cn], 0, 0, 0);
          acc[mi][cn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[c][mi], b1[cn], acc[mi][cn], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);                 // (the next chunk's planes are not built ahead: registers)
    }
    if (nc >= a.N) continue;
    if (has_x) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        xg[q] += __shfl_xor(xg[q], 16);
        xg[q] += __shfl_xor(xg[q], 32);
      }
      if (kg == 0) {
        float4 g = make_float4(xg[0], xg[1], xg[2], xg[3]);
        float4 *cp = reinterpret_cast<float4 *>(a.Cm + (size_t)mx * a.ldc + nc);
        if (a.beta != 0.f) {
          const float4 o = *cp;
          g.x += a.beta * o.x; g.y += a.beta * o.y; g.z += a.beta * o.z; g.w += a.beta * o.w;
        }
        *cp = g;
        if (a.P) {
          float4 *pp = reinterpret_cast<float4 *>(a.P + (size_t)mx * a.ldc + nc);
          float4 p = *pp;
          p.x -= a.lr * g.x; p.y -= a.lr * g.y; p.z -= a.lr * g.z; p.w -= a.lr * g.w;
          *pp = p;
        }
      }
    }
    // accumulator (mi, cn)[r] = G[m0 + 4 (4 kg + r) + mi][nc + cn], still scaled by the row's power of two (row 4 (4 kg + r) + mi
    // = column mi of lane i16 = 4 kg + r: cinv_s)
    float probe = 0.f;                                   // range guard (klstm_math.h): NaN as soon as one entry of the tile is Inf / NaN
    if (!UPD) {                                          // gradient only: sixteen stores
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int m = m0 + 16 * kg + 4 * r + mi;
          if (m >= a.m_main) continue;
          const float inv = cinv_s[4 * (4 * kg + r) + mi];
          const of32x4 g = {(acc[mi][0][r] + accx[mi][0][r] * (1.f / 2048.f)) * inv, (acc[mi][1][r] + accx[mi][1][r] * (1.f / 2048.f)) * inv,
                            (acc[mi][2][r] + accx[mi][2][r] * (1.f / 2048.f)) * inv, (acc[mi][3][r] + accx[mi][3][r] * (1.f / 2048.f)) * inv};
          probe = nonfinite_probe(nonfinite_probe(nonfinite_probe(nonfinite_probe(probe, g[0]), g[1]), g[2]), g[3]);
          *reinterpret_cast<of32x4 *>(a.Cm + (size_t)m * a.ldc + nc) = g;
        }
    } else {
      // Update in the same pass
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's rows of corr and W are in LDS (so are the next tile's rows of `in`)
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int m = m0 + 16 * kg + 4 * r + mi;
          if (m >= a.m_main) continue;
          const of32x4 oc = *reinterpret_cast<const of32x4 *>(wrows + (mi * 4 + r) * 1024 + lane * 16);
          const float inv = cinv_s[4 * (4 * kg + r) + mi];
          of32x4 g = {(acc[mi][0][r] + accx[mi][0][r] * (1.f / 2048.f)) * inv, (acc[mi][1][r] + accx[mi][1][r] * (1.f / 2048.f)) * inv,
                      (acc[mi][2][r] + accx[mi][2][r] * (1.f / 2048.f)) * inv, (acc[mi][3][r] + accx[mi][3][r] * (1.f / 2048.f)) * inv};
          probe = nonfinite_probe(nonfinite_probe(nonfinite_probe(nonfinite_probe(probe, g[0]), g[1]), g[2]), g[3]);
          g += a.beta * oc;
          *reinterpret_cast<of32x4 *>(a.Cm + (size_t)m * a.ldc + nc) = g;
          if (a.P) {
            const of32x4 op = *reinterpret_cast<const of32x4 *>(wrows + 16384 + (mi * 4 + r) * 1024 + lane * 16);
            *reinterpret_cast<of32x4 *>(a.P + (size_t)m * a.ldc + nc) = op - a.lr * g;
          }
        }
    }
    // An entry of `in` or of out_diff beyond the fp16 range left Inf / NaN in this wave's tile: the tile again in plain fp32, element
    // by element, written over what went out above (same lane, same address: ordered).  The Update form takes the old corr / W
    // values from the LDS copies the tile's epilogue used, which are still in place.
    if (wave_any(probe != probe)) {
      redo_note(a.redo);
#pragma unroll 1
      for (int q = 0; q < 64; q++) {
        const int mi = q >> 4, cn = (q >> 2) & 3, r = q & 3;
        const int m = m0 + 16 * kg + 4 * r + mi, n = nc + cn;
        if (m >= a.m_main || n >= a.N) continue;
        float g = redo_dot(a.diff + m, a.ldd, a.x + n, a.ldx, a.K);
        if (UPD) g += a.beta * *reinterpret_cast<const float *>(wrows + (mi * 4 + r) * 1024 + lane * 16 + cn * 4);
        a.Cm[(size_t)m * a.ldc + n] = g;
        if (UPD && a.P) a.P[(size_t)m * a.ldc + n] = *reinterpret_cast<const float *>(wrows + 16384 + (mi * 4 + r) * 1024 + lane * 16 + cn * 4) - a.lr * g;
      }
    }
  }
}

static int g_outer_f16 = 1;
void set_outer_f16(int on) { g_outer_f16 = on; }

// few frames (the contraction: three chunks of 32 are held in registers), a wide result: below ~2k rows of G the strips do not fill the chip
bool outer_f16_supported(int M, int N, int K, const float *diff, int ldd, const float *x, int ldx, const float *Cm, int ldc,
                         const float *P, const float *bias) {
  return g_outer_f16 != 0 && redo_count(REDO_OUTER) == 0 && K >= 1 && K <= 96 && M >= 2048 && M % 4 == 0 && N >= 64 && N % 4 == 0 && ldc % 4 == 0 && ldd % 4 == 0 &&
         ldx % 4 == 0 && ((reinterpret_cast<uintptr_t>(Cm) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(diff) |
                           reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0 && diff && x;
}
hipError_t launch_outer_f16(int M, int N, int K, const float *diff, int ldd, const float *x, int ldx, float beta, float *Cm, int ldc,
                            float *P, float lr, float beta_b, float *bias, float *bias_p, float lr_b, hipStream_t st, LaunchProbe pr) {
  OuterArgs a;
  a.diff = diff; a.ldd = ldd; a.x = x; a.ldx = ldx; a.K = K; a.M = M; a.N = N;
  a.bias_p = bias ? bias_p : nullptr; a.lr_b = lr_b;
  a.beta_b = beta_b; a.bias = bias; a.beta = beta; a.Cm = Cm; a.ldc = ldc; a.P = P; a.lr = lr;
  a.redo = redo_counters() ? redo_counters() + REDO_OUTER : nullptr;
  // one round of workgroups where a few rows past a whole number of strips per CU would start a second one
  static int ncu_of[64];                                 // per device, asked once
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
    if (ncu_of[dev] == 0) {
      int v = 0;
      ncu_of[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256;
    }
    ncu = ncu_of[dev];
  }
  int nstrip = (M + 63) / 64;
  a.m_main = M; a.nextra = 0;
  if (nstrip > ncu && M - 64 * ncu <= ncu) { nstrip = ncu; a.m_main = 64 * ncu; a.nextra = M - a.m_main; }
  const dim3 grid(nstrip), block(256);
#define OUTER_GO2(NCH_, UPD_) do { const unsigned shm = UPD_ ? 131072u : 0u; \
                                   if (UPD_) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_outer16<NCH_, UPD_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
                                   if (pr.start) hipExtLaunchKernelGGL((k_outer16<NCH_, UPD_>), grid, block, shm, st, pr.start, pr.stop, 0, a); \
                                   else hipLaunchKernelGGL((k_outer16<NCH_, UPD_>), grid, block, shm, st, a); } while (0)
#define OUTER_GO(NCH_) do { if (beta != 0.f || P) OUTER_GO2(NCH_, true); else OUTER_GO2(NCH_, false); } while (0)
  switch ((K + 31) / 32) {
    case 1: OUTER_GO(1); break;
    case 2: OUTER_GO(2); break;
    case 3: OUTER_GO(3); break;
    default: return hipErrorInvalidValue;
  }
#undef OUTER_GO
#undef OUTER_GO2
  return hipGetLastError();
}

}  // namespace klstm
