// kaldi-lstm_amd/csrc/klstm_gemm16.hip -- the batched products around the many-stream bf16 chains (BASELINE.json configs[4]:
// 32 streams x 20 frames = 640 rows per minibatch and layer) as ONE pipelined kernel family:
//
//     C[M x N] = A[M x K] B[N x K]^T (+ bias[n]) (+ add[m][n])        A, B fp32 in memory, k-contiguous rows,
//                                                                     rounded to bf16 (RNE) when staged, fp32 accumulate
//       x-projection  x(1..T) W_gifo_x^T + bias          ...streams.h:246, :259     640 x 4096 over K = I
//       P             out_diff W_r_m                     :408 (batched part)         640 x 1024 over K = R
//       d_r           out_diff + dgifo(2..T+1) W_gifo_r  :391                        640 x  512 over K = 4C
//       in_diff       dgifo(1..T) W_gifo_x               :457                        640 x    I over K = 4C
//
// Round 4's kernel (k_gemm_bf16_nt, klstm_kernels.hip) ran these at 5 % of the bf16 matrix peak and 0.9 TB/s.  What was measured on the
// way here (tools/gemm16_probe.py, tools/gemm16_anatomy.py = per-workgroup shader clocks of a TIMING instantiation; profiles/r05_gemm16_*.txt),
// in the order it was tried, x-projection 640 x 4096 x 512 as the yardstick (round 4: 16.2 us):
//   1. two register prefetch stages per workgroup instead of one: 19 us -- the compiler folds the stages' registers onto each other and
//      waits for nearly every outstanding load before it issues the next group;
//   2. LDS-DMA (global_load_lds_dwordx4) into fp32 stages by four loader waves, the MFMA waves converting as they read: 16 us -- the
//      loaders wait 80 clocks per stage for their requests, the MFMA waves spend 600 of 900 clocks per stage on their own instruction
//      stream (36 LDS loads and conversions for 8 MFMAs); the epilogue's 4-byte stores take as long as the K loop.  Rotating the K loop
//      per workgroup (L2 channels) and a fourth stage in flight change nothing: not memory latency;
//   3. the loaders also convert (read their own fp32 rows back, write a bf16 stage; the MFMA waves then need ONE 16-byte LDS load per
//      operand block): 17 us, now bound by LDS bandwidth -- every element crosses the LDS four times (DMA in, fp32 out, bf16 in, bf16
//      out: ~84 KB per stage and workgroup).  (On the way: LDS accesses the compiler can see make it drain ALL LDS-DMA requests first,
//      vmcnt(0); hand-counted waits around inline-asm loads are not safe either -- it moves the "loaded" registers before the wait.)
//   4. THIS FILE: the loaders load into REGISTERS (plain 16-byte loads, two stages per wave in flight), round to bf16 and write the
//      bf16 stage; the MFMA waves read one 16-byte LDS load per operand block.  Every load is issued unconditionally and the loop runs
//      whole rounds of two steps -- with a load behind a condition the compiler cannot count what is outstanding at the join and waits
//      for everything.  12.7-13.4 us (P: 11.0 -> 7.3, d_r + in_diff: 2 x (20.2 + 4.3) -> 25.5-27.6 in one launch).  What bounds it now:
//      fp32 operands are 4 bytes per element through the compute unit's 64-byte-per-clock vector-memory path -- (128 + 64) rows x 256
//      bytes per 64-k stage = 768 clocks per stage and workgroup, two workgroups per compute unit; K stages of 64 instead of 32 leave
//      the time per k unchanged.  The next factor of two is bf16 copies of the operands in memory (written by their producers), not
//      this kernel.
// The shape that came out:
//   * FOUR LOADER WAVES (one per SIMD) own 8-row groups of the (128 + BTN)-row stage: lane = (row, k-group of 8): 32 contiguous bytes
//     -> eight bf16 -> one 16-byte LDS write; two register stages; one workgroup barrier per 64-k stage hands a bf16 buffer (of two)
//     to the four MFMA waves (2 x 2 waves of 64 x 16 NJ outputs, v_mfma_f32_16x16x32_bf16);
//   * bf16 stage rows are 128 bytes, k-group q of row r in slot q ^ (((r >> 1) & 3) * 2): every ds_read_b128 lane group touches 16
//     distinct 16-byte slots of the bank window;
//   * the result leaves through LDS: accumulator blocks are transposed into a row-major tile, then every thread stores 16 bytes of a
//     row (+ bias, + add), all loads of all passes issued before the first store;
//   * split K WITHOUT a second launch: a slice stores its partial tile with 16-byte write-through (sc1) stores in accumulator
//     order (the slab layout is private), drains them, takes a ticket for its output tile; the workgroup that takes the last
//     ticket adds the ks slabs IN SLICE ORDER (deterministic, the same order whoever comes last) and writes the result --
//     MI355X_MICROARCH.md's in-launch hand-off recipe (write-through payload + per-wave drain + barrier + counter; the consumer
//     acquires at agent scope and reads with sc1 loads);
//   * K slice z of every output tile runs on XCD z (workgroup w lands on XCD w % 8): the slice's rows of A and B are read from HBM
//     once by that XCD and stay in its 4 MB L2 for all its output tiles;
//   * two products that contract the same rows (d_r and in_diff: dgifo shifted by one time block) share ONE launch.
#include "klstm_kernels.h"
#include "klstm_persist_dev.h"

#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

namespace klstm {

typedef __attribute__((address_space(1))) const void *g16_gptr;
typedef __attribute__((address_space(3))) void *g16_lptr;

namespace {

constexpr int G16_BT = 128, G16_BK = 64;            // rows of a tile, k of a stage (two MFMA k-steps: the per-stage hand-over latency, ~700 clocks measured, is paid once per 64)
template <int NJ, int NF> struct G16Geo {
  static constexpr int BTN = 32 * NJ;                                       // columns of the tile
  static constexpr int ROWS = G16_BT + BTN;                                 // LDS rows of a stage: the A rows, then the B rows
  static constexpr int STGH = ROWS * 128;                                   // ... of a bf16 stage (rows of 64 bf16 = 128 bytes)
  static constexpr int NQ = ROWS / 8;                                       // DMA instructions of a stage (8 rows = 1 KB each)
  static constexpr int NJL = NQ / 4;                                        // ... per loader wave (ROWS % 32 == 0)
  static constexpr int NCV = NJL;                                           // (row, k-group) conversion units per loader lane and stage: 8 rows x 8 k-groups per 64 lanes
  static constexpr int EPI = G16_BT * (BTN + 4) * 4;                        // the epilogue's row-major tile
  static constexpr int LDS = 2 * STGH > EPI ? 2 * STGH : EPI;               // two bf16 stage buffers; the epilogue's tile afterwards
  static constexpr int LDSH = NF * STGH > EPI ? NF * STGH : EPI;            // operands that ARE bf16 in memory: NF stage buffers filled by LDS-DMA
  static_assert(ROWS % 32 == 0, "loader geometry");
};

// bf16 stage: k-group q (16 bytes = 8 bf16; q = 0..7) of tile row r lives in slot q ^ g16_swh(r) of the 128-byte row.  A ds_read_b128 lane
// group (MI355X_MICROARCH.md: {0-3, 12-15, 20-27}, ...) holds rows {0-3, 12-15} at k-group q and rows {4-11} at q ^ 1: with this swizzle
// their 16 accesses fall on 16 distinct 16-byte slots of the 256-byte bank window (two rows).
__device__ __forceinline__ int g16_swh(int r) { return ((r >> 1) & 3) * 2; }

__device__ __forceinline__ uint4 g16_cvt8(const float4 &lo, const float4 &hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 p0 = {(__bf16)lo.x, (__bf16)lo.y}, p1 = {(__bf16)lo.z, (__bf16)lo.w}, p2 = {(__bf16)hi.x, (__bf16)hi.y}, p3 = {(__bf16)hi.z, (__bf16)hi.w};
  uint4 w;
  w.x = __builtin_bit_cast(unsigned, p0); w.y = __builtin_bit_cast(unsigned, p1); w.z = __builtin_bit_cast(unsigned, p2); w.w = __builtin_bit_cast(unsigned, p3);
  return w;
}

}  // namespace

struct Nt2Args {
  Nt2Job j[2];
  int ntm[2];                 // 128-row tiles per job
  int nt0, nt_all;            // output tiles of job 0, of both
  int ks, kslice;             // K slices (1, 2, 4 or 8) and their length (a multiple of 32)
  int cpg;                    // output tiles per XCD group (8 / ks groups share a slice's tiles)
  float *ws; unsigned ws_bytes;   // [ks][nt_all][128 x BTN] partial tiles in accumulator order
  unsigned *tickets;          // one per output tile, zero between launches
  long long *dbg;             // TIMING instantiations only (tools/gemm16_anatomy.py): 8 shader-clock sums per workgroup
};

// 512 threads: waves 0..3 contract (2 x 2 waves of 64 x 16 NJ), waves 4..7 bring the operands in and round them to bf16.
template <int N> __device__ __forceinline__ void g16_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `rem` stages of PER instructions each are outstanding (rem is wave-uniform, 0 .. MAXR)
template <int PER, int MAXR> __device__ __forceinline__ void g16_wait_stages(int rem) {
  if constexpr (MAXR > 0) {
    if (rem >= MAXR) { g16_wait_vm<MAXR * PER>(); return; }
    g16_wait_stages<PER, MAXR - 1>(rem);
  } else {
    g16_wait_vm<0>();
  }
}

// H: both operands exist as bf16 copies in memory (Nt2Job::Ah / Bh, written by their producers: the BPTT chain's dgifo rows, the
// Update's transposed weight copies) -- the loaders then move them straight into the bf16 stage by LDS-DMA (global_load_lds_dwordx4:
// 64 lanes x 16 bytes = 8 rows x 128 bytes per instruction; the stage's swizzle goes on the SOURCE address: lane l fills slot l & 7 of
// row l >> 3, so it fetches k-group (l & 7) ^ swizzle(row)), NF stage buffers, NF - 1 stages in flight with hand-counted vmcnt (the
// loader waves issue nothing else on the VM counter and touch no LDS themselves: cdna_hip_programming.md "pipelining across
// barriers").  Half the bytes per element through the vector-memory path, no conversion pass, no staging registers.  Same stages, same
// slices, same MFMA order as the fp32-operand form: BIT-IDENTICAL results when the copies are the RNE roundings of A and B.
template <int NJ, int NF, bool TIMING = false, bool H = false>
__global__ __launch_bounds__(512, NJ >= 4 ? 2 : 4) void k_gemm_bf16_nt2(Nt2Args a) {   // (waves per SIMD: two workgroups per compute unit up to 128 x 64 tiles)
  typedef G16Geo<NJ, NF> Geo;
  constexpr int BTN = Geo::BTN, STGH = Geo::STGH, NCV = Geo::NCV;
  constexpr int NSB = H ? NF : 2;                      // bf16 stage buffers in LDS
  static_assert(H || (NF >= 2 && NF <= 3 && (NF - 1) * 2 * NCV <= 63), "vmcnt is a 6-bit counter");
  static_assert(!H || (NF >= 3 && NF <= 6 && (NF - 2) * Geo::NJL <= 63 && NF * STGH <= 152 * 1024), "LDS-DMA form: three to six stage buffers");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned last_s;
  char *const hbuf = smem;                             // the two bf16 stage buffers
  // workgroup -> (K slice, output tile): slice z on XCDs z, z + ks, ... (w % 8 = XCD); a slice's tiles in contiguous m-fastest runs per XCD
  const int xcd = (int)(blockIdx.x & 7), l = (int)(blockIdx.x >> 3);
  const int z = xcd % a.ks, grp = xcd / a.ks;
  const int tile = grp * a.cpg + l;
  if (tile >= a.nt_all) return;
  const int jb = tile >= a.nt0 ? 1 : 0;
  const Nt2Job &g = a.j[jb];
  const int lt = tile - (jb ? a.nt0 : 0);
  const int m0 = (lt % a.ntm[jb]) * G16_BT, n0 = (lt / a.ntm[jb]) * BTN;
  const int kbeg = z * a.kslice, kend = min(g.K, kbeg + a.kslice);
  const int nstage = (kend - kbeg) / G16_BK;                                // (K % 64 == 0 on this path; every slice has a stage: the launcher's plan)
  const int nround = (nstage + NF - 1) / NF * NF;                           // the loaders' steps: whole rounds of NF
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kg = lane >> 4;
  const bool loader = wave >= 4;
  const int tw = wave & 3, wr = tw >> 1, wc = tw & 1;

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0, 0, 0, 0};

  if (loader && H) {
    // ---- LDS-DMA: instruction t of loader wave tw fills tile rows 8 (tw + 4 t) .. + 7 of the stage buffer (1 KB, lane-linear)
    constexpr int NJL = Geo::NJL, D = NF - 1;
    const unsigned short *srch[NJL];
#pragma unroll
    for (int t = 0; t < NJL; t++) {
      const int rt = (tw + 4 * t) * 8 + (lane >> 3), q = (lane & 7) ^ g16_swh(rt & 15);
      if (rt < G16_BT) srch[t] = g.Ah + (size_t)min(m0 + rt, g.M - 1) * g.lda + kbeg + 8 * q;
      else srch[t] = g.Bh + (size_t)min(n0 + rt - G16_BT, g.N - 1) * g.ldb + kbeg + 8 * q;
    }
    auto issue = [&](int st, int buf) {
      char *lb = hbuf + buf * STGH + tw * 1024;
#pragma unroll
      for (int t = 0; t < NJL; t++)
        __builtin_amdgcn_global_load_lds((g16_gptr)(srch[t] + (size_t)st * G16_BK), (g16_lptr)(lb + t * 4096), 16, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < D; st++) if (st < nstage) issue(st, st);
    int nb = D;                                        // buffer of the next stage to issue: (k + D) % NF
    long long th_wait = 0, th_bar = 0, th_issue = 0, th_t0 = TIMING ? clock64() : 0;
    for (int k = 0; k < nstage; k++) {
      // stages issued beyond k: min(D - 1, nstage - 1 - k); everything older has landed once the count is down to theirs
      const int rem = min(D - 1, nstage - 1 - k);
      const long long c0 = TIMING ? clock64() : 0;
      g16_wait_stages<NJL, D - 1>(rem);
      const long long c1 = TIMING ? clock64() : 0;
      asm volatile("s_barrier" ::: "memory");          // barrier k: stage k is in LDS (every loader waited); the MFMA waves are done with stage k - 1
      const long long c2 = TIMING ? clock64() : 0;
      if (k + D < nstage) issue(k + D, nb);            // ... whose buffer takes stage k + D
      nb = nb + 1 == NF ? 0 : nb + 1;
      if (TIMING) { th_wait += c1 - c0; th_bar += c2 - c1; th_issue += clock64() - c2; }
    }
    if (TIMING && wave == 4 && lane == 0) {
      long long *d = a.dbg + (size_t)blockIdx.x * 16;
      d[4] = th_wait; d[5] = clock64() - th_t0; d[8] = th_wait; d[9] = th_issue; d[10] = th_bar;
    }
  } else if (loader) {
    // ---- unit t of a lane = (row group t of this wave, row lane >> 3 of the group, k-group lane & 7): 32 contiguous bytes of an operand
    // row (two 16-byte loads; eight lanes cover 256 contiguous bytes) -> eight bf16 -> one 16-byte write into the bf16 stage.  Rows
    // beyond the operand read its last row (their results are never stored).  Loader wave tw owns tile rows 8 (tw + 4 t) .. + 7.
    const float *src[NCV];
    int ch[NCV];
#pragma unroll
    for (int t = 0; t < NCV; t++) {
      const int rt = (tw + 4 * t) * 8 + (lane >> 3), kq = lane & 7;
      if (rt < G16_BT) src[t] = g.A + (size_t)min(m0 + rt, g.M - 1) * g.lda + kbeg + 8 * kq;
      else src[t] = g.B + (size_t)min(n0 + rt - G16_BT, g.N - 1) * g.ldb + kbeg + 8 * kq;
      ch[t] = rt * 128 + (kq ^ g16_swh(rt & 15)) * 16;
    }
    // NF register stages, each 2 NCV x 16 bytes per lane; stage s of the K slice lives in set s % NF.  The loop below is unrolled by NF
    // so that every set is loaded and consumed by the same static code.  Plain loads: the compiler counts vmcnt itself (in a wave that
    // does nothing else it keeps the sets apart and waits for exactly the oldest stage -- checked in the ISA: s_waitcnt vmcnt(2 NCV (NF - 1))
    // in front of the conversion; hand-counted waits around inline-asm loads are NOT safe here: the compiler moves "loaded" registers
    // around before the wait it cannot see).
    f32x4 rl[NF][NCV], rh[NF][NCV];
    auto fetch = [&](int s, f32x4 (&lo)[NCV], f32x4 (&hi)[NCV]) {
#pragma unroll
      for (int t = 0; t < NCV; t++) {
        const float *p = src[t] + (size_t)s * G16_BK;
        lo[t] = *reinterpret_cast<const f32x4 *>(p);
        hi[t] = *reinterpret_cast<const f32x4 *>(p + 4);
      }
    };
    auto cvw = [&](int hb, const f32x4 (&lo)[NCV], const f32x4 (&hi)[NCV]) {
#pragma unroll
      for (int t = 0; t < NCV; t++) {
        const float4 l4 = {lo[t].x, lo[t].y, lo[t].z, lo[t].w}, h4 = {hi[t].x, hi[t].y, hi[t].z, hi[t].w};
        *reinterpret_cast<uint4 *>(smem + ch[t] + hb * STGH) = g16_cvt8(l4, h4);
      }
    };
    long long tl_wait = 0, tl_conv = 0, tl_issue = 0, tl_bar = 0, tl_t0 = TIMING ? clock64() : 0;
    // Every load below is issued UNCONDITIONALLY (stage indices past the end are clamped to the last stage: a redundant load of bytes
    // that are in the L2 anyway) and the loop runs a whole number of rounds of NF steps: with a load behind a condition the compiler
    // cannot know how many are outstanding at the join and waits for all of them -- vmcnt(5) ... vmcnt(0) in front of every conversion,
    // the pipeline gone (seen in the ISA).  The MFMA waves make up for the padding steps with bare barriers.
    const int last = nstage - 1;
#pragma unroll
    for (int s = 0; s < NF; s++) fetch(s < last ? s : last, rl[s], rh[s]);
    // step k: stage k has landed in set k % NF -> round, write bf16 buffer k & 1 (the MFMA waves finished reading it -- stage k - 2 --
    // before barrier k - 1) -> the set takes stage k + NF -> barrier k hands bf16 stage k over
    auto step = [&](int k, f32x4 (&lo)[NCV], f32x4 (&hi)[NCV]) {
      const long long tw0 = TIMING ? clock64() : 0;
      cvw(k & 1, lo, hi);
      const long long tc1 = TIMING ? clock64() : 0;
      fetch(k + NF < last ? k + NF : last, lo, hi);
      const long long tc2 = TIMING ? clock64() : 0;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (TIMING) { const long long tc3 = clock64(); tl_conv += tc1 - tw0; tl_issue += tc2 - tc1; tl_bar += tc3 - tc2; }
    };
    for (int k = 0; k < nround; k += NF) {
      step(k, rl[0], rh[0]);
      step(k + 1, rl[1], rh[1]);
      if constexpr (NF >= 3) step(k + 2, rl[2], rh[2]);
    }
    if (TIMING && wave == 4 && lane == 0) { a.dbg[(size_t)blockIdx.x * 16 + 8] = tl_conv; a.dbg[(size_t)blockIdx.x * 16 + 9] = tl_issue; a.dbg[(size_t)blockIdx.x * 16 + 10] = tl_bar; }
    if (TIMING && wave == 4 && lane == 0) { a.dbg[(size_t)blockIdx.x * 16 + 4] = tl_wait; a.dbg[(size_t)blockIdx.x * 16 + 5] = clock64() - tl_t0; }
  } else {
    // ---- MFMA waves: lane (i16, kg) reads k-group kg of row i16 of each of its 16-row blocks
    const int sw = g16_swh(i16);
    const int sh0 = (kg ^ sw) * 16, sh1 = ((4 + kg) ^ sw) * 16;      // the two MFMA k-steps of a stage
    const int arow = (wr * 64 + i16) * 128, brow = (G16_BT + wc * 16 * NJ + i16) * 128;
    long long tm_bar = 0, tm_rd = 0, tm_t0 = TIMING ? clock64() : 0;
    int rb = 0;                                        // stage k lives in buffer k % NSB
    for (int k = 0; k < nstage; k++) {
      const long long tb0 = TIMING ? clock64() : 0;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // barrier k (own reads of stage k - 1 are complete)
      if (TIMING) tm_bar += clock64() - tb0;
      const char *sb = hbuf + rb * STGH;
      rb = rb + 1 == NSB ? 0 : rb + 1;
      bf16x8 af[2][4], bfr[2][NJ];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        af[0][i] = *reinterpret_cast<const bf16x8 *>(sb + arow + i * 2048 + sh0);
        af[1][i] = *reinterpret_cast<const bf16x8 *>(sb + arow + i * 2048 + sh1);
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        bfr[0][j] = *reinterpret_cast<const bf16x8 *>(sb + brow + j * 2048 + sh0);
        bfr[1][j] = *reinterpret_cast<const bf16x8 *>(sb + brow + j * 2048 + sh1);
      }
      if (TIMING) { const long long tr0 = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tm_rd += clock64() - tr0; }
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[h][i], bfr[h][j], acc[i][j], 0, 0, 0);
    }
    if (!H) for (int k = nstage; k < nround; k++) asm volatile("s_barrier" ::: "memory");   // (the loaders' padding steps)
    if (TIMING && wave == 0 && lane == 0) {
      a.dbg[(size_t)blockIdx.x * 16 + 0] = tm_bar; a.dbg[(size_t)blockIdx.x * 16 + 1] = tm_rd; a.dbg[(size_t)blockIdx.x * 16 + 2] = clock64() - tm_t0;
      a.dbg[(size_t)blockIdx.x * 16 + 3] = nstage;
    }
  }
  const long long te0 = TIMING ? clock64() : 0;
  __syncthreads();                                   // (every LDS read of the K loop is complete: the staging area becomes the result tile)

  // ---- epilogue.  Block (i, j) of MFMA wave (wr, wc), lane (i16, kg): rows 64 wr + 16 i + 4 kg + (0..3) at column 16 NJ wc + 16 j + i16.
  const int etid = tid & 255;                        // (thread index among the MFMA waves; loader wave w + 4 mirrors MFMA wave w)
  float *const T = reinterpret_cast<float *>(smem);  // [128][BTN + 4]
  constexpr int TLD = BTN + 4;
  auto to_tile = [&](int i, int j, const f32x4 &v4) {
    float *tp = T + (wr * 64 + i * 16 + 4 * kg) * TLD + wc * 16 * NJ + j * 16 + i16;
    tp[0] = v4.x; tp[TLD] = v4.y; tp[2 * TLD] = v4.z; tp[3 * TLD] = v4.w;
  };
  // every thread: 16 bytes of a row per pass, rows in 4 NJ... contiguous runs of BTN floats
  auto store_tile = [&]() {
    const bool vec = ((reinterpret_cast<uintptr_t>(g.C) | (g.add ? reinterpret_cast<uintptr_t>(g.add) : 0)) & 15) == 0 && g.ldc % 4 == 0 &&
                     (!g.add || g.add_ld % 4 == 0) && g.N % 4 == 0;
    constexpr int Q = BTN / 4, NP = G16_BT * Q / 512;   // float4s per row; passes of the 512 threads (tile fully covered: 128 Q % 512 == 0)
    if (vec) {
      // all loads of all passes first, then the stores: one memory round trip for the tile instead of one per pass
      float4 v[NP], bq[NP], dq[NP];
      bool on[NP];
#pragma unroll
      for (int p_ = 0; p_ < NP; p_++) {
        const int u = tid + 512 * p_, rl = u / Q, c4 = (u % Q) * 4, m = m0 + rl, n = n0 + c4;
        on[p_] = m < g.M && n < g.N;
        v[p_] = *reinterpret_cast<const float4 *>(T + rl * TLD + c4);
        bq[p_] = (g.bias && on[p_]) ? *reinterpret_cast<const float4 *>(g.bias + n) : float4{0.f, 0.f, 0.f, 0.f};
        dq[p_] = (g.add && on[p_]) ? *reinterpret_cast<const float4 *>(g.add + (size_t)m * g.add_ld + n) : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int p_ = 0; p_ < NP; p_++) {
        if (!on[p_]) continue;
        const int u = tid + 512 * p_, rl = u / Q, c4 = (u % Q) * 4;
        float4 w = v[p_];
        w.x += bq[p_].x; w.y += bq[p_].y; w.z += bq[p_].z; w.w += bq[p_].w;
        if (g.add) { w.x = dq[p_].x + w.x; w.y = dq[p_].y + w.y; w.z = dq[p_].z + w.z; w.w = dq[p_].w + w.w; }   // (d_r = out_diff + the product, :391)
        *reinterpret_cast<float4 *>(g.C + (size_t)(m0 + rl) * g.ldc + n0 + c4) = w;
      }
      return;
    }
    for (int u = tid; u < G16_BT * Q; u += 512) {
      const int rl = u / Q, c4 = (u % Q) * 4, m = m0 + rl, n = n0 + c4;
      if (m >= g.M || n >= g.N) continue;
      const float4 v = *reinterpret_cast<const float4 *>(T + rl * TLD + c4);
      const float e[4] = {v.x, v.y, v.z, v.w};
      for (int q = 0; q < 4 && n + q < g.N; q++) {
        float w = e[q] + (g.bias ? g.bias[n + q] : 0.f);
        if (g.add) w = g.add[(size_t)m * g.add_ld + n + q] + w;
        g.C[(size_t)m * g.ldc + n + q] = w;
      }
    }
  };
  if (a.ks == 1) {
    if (!loader) {
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) to_tile(i, j, acc[i][j]);
    }
    __syncthreads();
    store_tile();
    if (TIMING && wave == 0 && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.dbg[(size_t)blockIdx.x * 16 + 6] = clock64() - te0; }
    return;
  }
  // ---- split K: partial tile -> slab z (accumulator order, 16-byte write-through stores), drain, ticket; the last arrival adds the
  // slabs in slice order -- all eight waves take part in that (loader wave w + 4 takes blocks i = 2, 3 of MFMA wave w's layout)
  const __amdgpu_buffer_rsrc_t rs = buf_rsrc(a.ws, (int)a.ws_bytes);
  const unsigned tile_bytes = (unsigned)(G16_BT * BTN * 4);
  const unsigned mine = ((unsigned)z * (unsigned)a.nt_all + (unsigned)tile) * tile_bytes;
  if (!loader) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, (int)(mine + (unsigned)(((i * NJ + j) * 256 + etid) * 16)), 0, 16);   // sc1
    // ... and into LDS in the same order (the staging area is free): should this workgroup be the last to arrive, it adds its OWN partial
    // tile from here instead of reading it back from memory -- one slab of ks less through the one compute unit that does the sum
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) *reinterpret_cast<f32x4 *>(smem + ((i * NJ + j) * 256 + etid) * 16) = acc[i][j];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // every wave: its slab stores have been performed at device scope
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = old == (unsigned)(a.ks - 1) ? 1u : 0u;
    if (last) {
      __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (ready for the next launch on this stream)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    last_s = last;
  }
  __syncthreads();
  if (TIMING && wave == 0 && lane == 0) a.dbg[(size_t)blockIdx.x * 16 + 6] = clock64() - te0;       // (slab stores + drain + ticket)
  if (!last_s) return;
  constexpr int ZC = 4;                                             // slabs whose loads are in flight together (2 NJ ZC x 16 bytes per thread)
  const int ibase = loader ? 2 : 0;
  f32x4 sum[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) sum[i][j] = (f32x4){0, 0, 0, 0};
  // slice order, whoever came last: the other ks - 1 slabs from memory (up to ZC of them in flight together), the own one from LDS
  auto add_own = [&]() {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) sum[i][j] += *reinterpret_cast<const f32x4 *>(smem + (((ibase + i) * NJ + j) * 256 + etid) * 16);
  };
  bool own_done = false;
  for (int x0 = 0; x0 < a.ks - 1; x0 += ZC) {                       // x: index among the OTHER slices; slice = x < z ? x : x + 1
    u32x4 q[ZC][2][NJ];
#pragma unroll
    for (int zc = 0; zc < ZC; zc++) {
      if (x0 + zc >= a.ks - 1) continue;
      const unsigned zz = (unsigned)(x0 + zc < z ? x0 + zc : x0 + zc + 1);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
          q[zc][i][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((zz * (unsigned)a.nt_all + (unsigned)tile) * tile_bytes +
                                                                        (unsigned)((((ibase + i) * NJ + j) * 256 + etid) * 16)), 0, 16);   // sc1
    }
#pragma unroll
    for (int zc = 0; zc < ZC; zc++)
      if (x0 + zc < a.ks - 1) {
        if (!own_done && x0 + zc >= z) { add_own(); own_done = true; }   // (the next one from memory is slice x + 1 > z)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < NJ; j++) sum[i][j] += __builtin_bit_cast(f32x4, q[zc][i][j]);
      }
  }
  if (!own_done) add_own();
  __syncthreads();                                                  // (every read of the own partial tile is done: the area becomes the result tile)
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int i = 0; i < 2; i++) to_tile(ibase + i, j, sum[i][j]);
  __syncthreads();
  store_tile();
  if (TIMING && wave == 0 && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.dbg[(size_t)blockIdx.x * 16 + 7] = clock64() - te0; }   // (the last arrival: + the slab sums)
}

// ---------------------------------------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------------------------------------
static inline int g16_cdiv(int a, int b) { return (a + b - 1) / b; }
static long long *g16_dbg = nullptr;                 // set by gemm_bf16_nt2_debug_buffer (tools/gemm16_probe.py): the TIMING instantiations run
void gemm_bf16_nt2_debug_buffer(long long *dev) { g16_dbg = dev; }

bool gemm_bf16_nt2_supported(const Nt2Job &g) {
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return g.M >= 256 && g.N >= 32 && g.K >= 64 && g.K % 64 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && al16(g.A) && al16(g.B) && g.C;
}
// ... and may it read the bf16 copies (both given, rows 16-byte aligned)?
bool gemm_bf16_nt2_copies_usable(const Nt2Job &g) {
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return g.Ah && g.Bh && al16(g.Ah) && al16(g.Bh) && g.lda % 8 == 0 && g.ldb % 8 == 0;
}

// Tile width (16-column blocks per wave) and K slices for `njobs` products launched together.  Empirical (tools/gemm16_probe.py on the
// shapes of BASELINE.json configs[4], 640 rows; profiles/r05_gemm16_probe.txt): the kernel is bound by the compute unit's vector-memory
// path (fp32 operands: 4 bytes per element through 64 bytes per clock), so the widest tile that still gives every compute unit a
// workgroup wins, and K is split only when the output has few tiles:
//   long K (>= 2048), two products or >= 32 tiles of 128 x 128: 128 x 128 tiles, 4 slices   (d_r + in_diff: 25.5-27.6 us; 49 in round 4)
//   long K, one product with fewer tiles:                       128 x 64 tiles, 4 slices    (d_r alone: 19.3-20.1 us; 24.5)
//   short K, wide output (N >= 2048):                            128 x 128 tiles, no split   (x-projection: 12.7-13.4 us; 16.2)
//   short K, narrow output:                                      128 x 32 tiles, no split    (P: 7.3-7.5 us; 11.0)
// From bf16 copies (LDS-DMA form; tools/gemm16_probe.py --copies, profiles/r05_gemm16_copies.txt): the stage is half the bytes and the
// loaders issue 1 KB per instruction, so narrower tiles with fewer slices win -- the in-launch reduction (slab round trips through
// memory, summed by ONE compute unit per tile) is then the larger part: d_r + in_diff 128 x 64 tiles in 2 slices.
Nt2Plan gemm_bf16_nt2_plan(const Nt2Job *jobs, int njobs, int force_nj, int force_ks) {
  int kmin = jobs[0].K, nmax = jobs[0].N;
  for (int q = 1; q < njobs; q++) { kmin = jobs[q].K < kmin ? jobs[q].K : kmin; nmax = jobs[q].N > nmax ? jobs[q].N : nmax; }
  auto tiles = [&](int nj) { int nt = 0; for (int q = 0; q < njobs; q++) nt += g16_cdiv(jobs[q].M, G16_BT) * g16_cdiv(jobs[q].N, 32 * nj); return nt; };
  bool h = true;                                       // the LDS-DMA form (bf16 copies in memory) will run
  for (int q = 0; q < njobs; q++) h = h && gemm_bf16_nt2_copies_usable(jobs[q]);
  int nj, ks;
  if (kmin >= 2048 && h && njobs == 2) { nj = 2; ks = 2; }            // (copies: d_r + in_diff 18.5-20 us in two slices of 128 x 64 tiles, 21-22 as below; 27-30 from fp32 operands)
  else if (kmin >= 2048) { nj = (njobs == 2 || tiles(4) >= 32) ? 4 : 2; ks = 4; }
  else if (nmax >= 2048) { nj = 4; ks = 1; }
  else { nj = tiles(1) > 512 ? 2 : 1; ks = 1; }
  if (force_nj) nj = force_nj;
  if (force_ks) ks = force_ks;
  while (ks > 1 && (kmin / ks < 2 * G16_BK || g16_cdiv(g16_cdiv(kmin, ks), G16_BK) * G16_BK * (ks - 1) >= kmin)) ks >>= 1;   // (no short, no empty slice)
  Nt2Plan pl{nj, ks, tiles(nj), 0};
  pl.ws_floats = ks > 1 ? (size_t)ks * pl.nt * G16_BT * 32 * nj : 0;
  return pl;
}

hipError_t launch_gemm_bf16_nt2(const Nt2Job *jobs, int njobs, const Nt2Plan &pl, float *ws, size_t ws_floats, unsigned *tickets, int ntickets,
                                hipStream_t st, LaunchProbe pr) {
  if (njobs < 1 || njobs > 2) return hipErrorInvalidValue;
  Nt2Args a;
  int kmax = 0;
  for (int q = 0; q < 2; q++) {
    a.j[q] = jobs[q < njobs ? q : 0];
    a.ntm[q] = g16_cdiv(a.j[q].M, G16_BT);
    if (q < njobs && !gemm_bf16_nt2_supported(jobs[q])) return hipErrorInvalidValue;
    if (q < njobs && jobs[q].K > kmax) kmax = jobs[q].K;
  }
  a.nt0 = a.ntm[0] * g16_cdiv(jobs[0].N, 32 * pl.nj);
  a.nt_all = a.nt0 + (njobs > 1 ? a.ntm[1] * g16_cdiv(jobs[1].N, 32 * pl.nj) : 0);
  a.ks = pl.ks;
  a.kslice = g16_cdiv(g16_cdiv(kmax, pl.ks), G16_BK) * G16_BK;
  a.cpg = g16_cdiv(a.nt_all, 8 / pl.ks);
  a.ws = ws; a.tickets = tickets;
  const size_t need = pl.ks > 1 ? (size_t)pl.ks * a.nt_all * G16_BT * 32 * pl.nj : 0;
  if (pl.ks > 1 && (!ws || !tickets || ws_floats < need || ntickets < a.nt_all || need * 4 >= (1ull << 31))) return hipErrorInvalidValue;
  if (pl.ks != 1 && pl.ks != 2 && pl.ks != 4 && pl.ks != 8) return hipErrorInvalidValue;
  for (int q = 0; q < njobs; q++)                     // every K slice of every product has at least one stage (the kernel's loaders load unconditionally)
    if ((long)a.kslice * (pl.ks - 1) >= jobs[q].K) return hipErrorInvalidValue;
  a.ws_bytes = (unsigned)(need * 4);
  a.dbg = g16_dbg;
  const dim3 grid(8 * a.cpg), block(512);
  auto go = [&](auto kern, int lds, int hslot = -1) -> hipError_t {
    // (per launch, like every other kernel here: the attribute belongs to the current DEVICE -- engines may sit on several -- and a
    //  process-wide "already raised" cache would also be a data race between host threads)
    (void)hslot;
    {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
    }
    if (pr.start) hipExtLaunchKernelGGL(kern, grid, block, lds, st, pr.start, pr.stop, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    return hipGetLastError();
  };
  bool h = true;                                       // every product of the launch has both bf16 copies: the LDS-DMA form
  for (int q = 0; q < njobs; q++) h = h && gemm_bf16_nt2_copies_usable(jobs[q]);
  if (h && g16_dbg) {
    if (pl.nj == 4) return go(k_gemm_bf16_nt2<4, 4, true, true>, G16Geo<4, 4>::LDSH, 11);
    if (pl.nj == 2) return go(k_gemm_bf16_nt2<2, 6, true, true>, G16Geo<2, 6>::LDSH, 10);
    return go(k_gemm_bf16_nt2<1, 6, true, true>, G16Geo<1, 6>::LDSH, 9);
  }
  if (h) {
    // stage buffers: what fits the LDS -- the DMA's issue-to-landed time (~1 us) is hidden by the stages in flight, nothing else
    if (pl.nj == 4) return go(k_gemm_bf16_nt2<4, 4, false, true>, G16Geo<4, 4>::LDSH, 8);
    if (pl.nj == 2) return go(k_gemm_bf16_nt2<2, 6, false, true>, G16Geo<2, 6>::LDSH, 7);
    return go(k_gemm_bf16_nt2<1, 6, false, true>, G16Geo<1, 6>::LDSH, 6);
  }
  if (g16_dbg) {
    if (pl.nj == 4) return go(k_gemm_bf16_nt2<4, 2, true>, G16Geo<4, 2>::LDS);
    if (pl.nj == 2) return go(k_gemm_bf16_nt2<2, 2, true>, G16Geo<2, 2>::LDS);
    return go(k_gemm_bf16_nt2<1, 2, true>, G16Geo<1, 2>::LDS);
  }
  if (pl.nj == 4) return go(k_gemm_bf16_nt2<4, 2>, G16Geo<4, 2>::LDS);
  if (pl.nj == 2) return go(k_gemm_bf16_nt2<2, 2>, G16Geo<2, 2>::LDS);
  return go(k_gemm_bf16_nt2<1, 2>, G16Geo<1, 2>::LDS);
}

}  // namespace klstm
