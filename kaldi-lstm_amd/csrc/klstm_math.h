// kaldi-lstm_amd/csrc/klstm_math.h -- device-side types, tile constants and the scalar LSTM math shared by
// klstm_kernels.hip (one launch per recurrence step) and klstm_persist.hip (weights-resident persistent chain): both
// must evaluate every elementwise formula with the SAME instruction sequence, so there is exactly one definition.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace klstm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// bf16 operand mode (engine option "bf16"): weights packed as bf16, activations rounded to bf16 (RNE) when
// they are staged into LDS, fp32 accumulate; masters, planes and gradients stay fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 pack_bf16x4(const float4 &v) {
  const bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  return __builtin_bit_cast(uint2, h);
}

constexpr int NW = 8;        // waves per workgroup in the step kernels (K split)
constexpr int KCH = 32;      // K chunk one wave consumes per MFMA group (4 k-groups x 8)
constexpr int KSMAX = 4;     // max split-K slabs of k_dr_step

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// scalar math: the overflow-safe forms Kaldi's CPU path uses, no FMA contraction so that the
// elementwise results track the CPU formulation to the last bit where possible.
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// sigmoid / tanh on the hardware transcendental units (v_exp_f32, v_rcp_f32: ~1 ulp each).  The
// reference CPU forms (overflow-safe split, expf, IEEE divide) cost a ~500-cycle dependent chain on
// the 16 lanes that own a tile's cell math; IEEE inf arithmetic makes the single-branch forms below
// saturate correctly (exp2(+big) = inf -> rcp = 0), and the deviation is <= 3e-7 absolute.
__device__ __forceinline__ float k_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float k_tanh(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.8853900817779268f));
}
// DiffSigmoid / DiffTanh with the reference's double literal (kaldi-matrix.cc:2562-2593)
__device__ __forceinline__ float k_diff_sigmoid(float d, float y) {
  return (float)((double)(d * y) * (1.0 - (double)y));
}
__device__ __forceinline__ float k_diff_tanh(float d, float y) {
  return (float)((double)d * (1.0 - (double)(y * y)));
}


// The same two derivatives in fp32 with ONE rounding:  d*y*(1 - y) = p - p*y  with p = fl(d*y), and  d*(1 - y*y) = d - d*t
// with t = fl(y*y), each as a single fused multiply-add (contraction is off in this file, so the FMA is explicit).  The
// reference rounds the double product to double and then to float; that double rounding differs from the single
// rounding here only when the exact value lies within 2^-53 (relative) of a float rounding boundary, ~2^-29 of all
// inputs, by one ulp.  Used where the derivative is REPLICATED (persistent backward chain: every workgroup recomputes the
// whole layer's dgifo, and the fp64 conversions and multiplies of the forms above were 2-3 us per step there).
__device__ __forceinline__ float k_diff_sigmoid_fma(float d, float y) {
  const float p = d * y;
  return __builtin_fmaf(-p, y, p);
}
__device__ __forceinline__ float k_diff_tanh_fma(float d, float y) {
  const float t = y * y;
  return __builtin_fmaf(-d, t, d);
}

// Three-way bf16 split of an fp32 number (klstm_fold3.hip): x = h1 + h2 + h3 up to 2^-24 |x|, both residuals exact.
__device__ __forceinline__ unsigned short bf16_rne(float x) {
  const unsigned u = __float_as_uint(x);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void bf16_split3(float x, unsigned short &h1, unsigned short &h2, unsigned short &h3) {
  h1 = bf16_rne(x);
  const float r1 = x - bf16_f32(h1);
  h2 = bf16_rne(r1);
  h3 = bf16_rne(r1 - bf16_f32(h2));
}
// four consecutive elements -> 8 bytes into each of the three planes
__device__ __forceinline__ void bf16_split3_store4(const float (&v)[4], unsigned short *dst, long plane) {
  unsigned short h[3][4];
#pragma unroll
  for (int e = 0; e < 4; e++) bf16_split3(v[e], h[0][e], h[1][e], h[2][e]);
#pragma unroll
  for (int q = 0; q < 3; q++)
    *reinterpret_cast<uint2 *>(dst + q * plane) = make_uint2(h[q][0] | ((unsigned)h[q][1] << 16), h[q][2] | ((unsigned)h[q][3] << 16));
}

// Two-way fp16 split (the lighter fold product, klstm_fold3.hip NPL = 2; the output layer's products, klstm_outer.hip): x = h1 + h2 / 2048
// up to 2^-22 |x| for 2^-14 <= |x| < 65504; the residual (<= 2^-11 |x|) is scaled by 2^11 before it is rounded -- unscaled it would
// sit in fp16's subnormal range for every |x| < 2^-3 and lose bits (tests/test_f16_split.py: 2^-14 relative around 2^-10).  Below
// 2^-14 the absolute error, < 2^-35, is far under a product's rounding.  The product kernel keeps a1 b1 and (a1 b2 + a2 b1) in separate
// accumulators and folds the 2^-11 in at the end.
__device__ __forceinline__ void f16_split2(float x, unsigned short &h1, unsigned short &h2) {
  const _Float16 a = (_Float16)x;                       // RNE
  const float r = x - (float)a;                         // exact
  const _Float16 b = (_Float16)(r * 2048.f);
  h1 = __builtin_bit_cast(unsigned short, a); h2 = __builtin_bit_cast(unsigned short, b);
}
// two at once: the packed conversions and packed fp32 arithmetic of gfx950 (v_cvt_pk_f16_f32, v_pk_add_f32, v_pk_mul_f32) -- three
// instructions per element instead of six; same bits as f16_split2.  h1 / h2 = the pairs (x0 low half, x1 high half)
typedef _Float16 klstm_h2 __attribute__((ext_vector_type(2)));
typedef float klstm_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void f16_split2_pair(float x0, float x1, unsigned &h1, unsigned &h2) {
  const klstm_f2 f = {x0, x1};
  const klstm_h2 a = __builtin_convertvector(f, klstm_h2);
  const klstm_f2 r = (f - __builtin_convertvector(a, klstm_f2)) * 2048.f;
  const klstm_h2 b = __builtin_convertvector(r, klstm_h2);
  h1 = __builtin_bit_cast(unsigned, a); h2 = __builtin_bit_cast(unsigned, b);
}
// ---- range guard of the fp16-plane products ------------------------------------------------------------------------------
// The reference's products are fp32 (cblas_sgemm / cublasSgemm, kaldi-matrix.cc:160-175): any finite operand is legal.  An
// fp16 plane overflows for |x| >= 65520 (h1 = Inf, h2 = -Inf).  An Inf operand can only produce Inf or NaN -- never a finite
// wrong number -- and it reaches EVERY accumulator whose row / column it sits in, so a look at a wave's accumulators after the
// K loop detects it: v * 0 is NaN exactly when v is Inf or NaN.  The wave then recomputes ITS outputs with plain fp32
// multiply-adds from the fp32 operands (slow and rare: one dot product per accumulator element and lane) and counts the event in
// a host-mapped word; the host takes that product to its fp32-range kernel from the next call on (klstm_kernels.h redo_*).
// A result that overflows fp32 itself takes the same path and comes out non-finite again, as it does in the reference.
__device__ __forceinline__ float nonfinite_probe(float t, float v) { return __builtin_fmaf(v, 0.f, t); }
__device__ __forceinline__ bool wave_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
__device__ __forceinline__ void redo_note(unsigned *ctr) {      // (called from wave-uniform code: lane 0 is active)
  if (ctr && (threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// sequential fp32 dot product of the redo path: n terms, strides in elements
__device__ __forceinline__ float redo_dot(const float *a, long sa, const float *b, long sb, int n) {
  float s = 0.f;
  for (int k = 0; k < n; k++) s = __builtin_fmaf(a[(long)k * sa], b[(long)k * sb], s);
  return s;
}
// Derivative operands (out_diff, dgifo) of the on-the-fly split kernels are multiplied by 2^12 before they are split and the result
// by 2^-12 afterwards (both exact): the two planes then carry 22 bits for 2^-26 <= |x| < 16 instead of 2^-14 <= |x| < 65504 --
// late-training derivatives of 1e-7 keep full precision (unscaled: 3e-4 relative); below 2^-26 the absolute error is < 2^-47;
// at 16 and above the range guard takes over.
constexpr float DERIV_SCALE = 4096.f, DERIV_UNSCALE = 1.f / 4096.f;

// mode 1: three bf16 planes, mode 2: two fp16 planes (plane 1 scaled by 2^11), mode 3: ONE bf16 plane (the operand itself rounded
// to bf16: the fold product of the bf16 operand mode, klstm_persist_ms.hip)
__device__ __forceinline__ void split_store4(int mode, const float (&v)[4], unsigned short *dst, long plane) {
  if (mode == 3) {
    *reinterpret_cast<uint2 *>(dst) = make_uint2(bf16_rne(v[0]) | ((unsigned)bf16_rne(v[1]) << 16), bf16_rne(v[2]) | ((unsigned)bf16_rne(v[3]) << 16));
    return;
  }
  if (mode != 2) { bf16_split3_store4(v, dst, plane); return; }
  unsigned h[2][2];
  f16_split2_pair(v[0], v[1], h[0][0], h[1][0]);
  f16_split2_pair(v[2], v[3], h[0][1], h[1][1]);
#pragma unroll
  for (int q = 0; q < 2; q++) *reinterpret_cast<uint2 *>(dst + q * plane) = make_uint2(h[q][0], h[q][1]);
}

}  // namespace klstm
