// kaldi-lstm_amd/csrc/klstm_kernels.h -- host-callable launchers of the gfx950 kernels.
// Every launcher enqueues on `st` and returns the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>

namespace klstm {

// Activation planes are time-major (row = tb*S + s, tb = time block 0..T+1 like the reference
// slab rows, ...streams.h:229-243) but split per column group so every kernel streams
// unit-stride rows:
//   gifo [(T+2)S x 4C]  activated g|i|f|o           cc, hh, mm [(T+2)S x C]   rr [(T+2)S x R]
//   dgifo[(T+2)S x 4C]  d(pre-activation) g|i|f|o   dc [(T+2)S x C]           dr [(T+2)S x R]
struct Dims { int I, C, R, S, T; };

struct FwdPtrs {
  const float *wx, *wr, *bias, *pi, *pf, *po, *wm;   // canonical [N x K] row-major weights
  float *gifo, *cc, *hh, *mm, *rr;
  const float *prev_c, *prev_r;                      // carried state [S x C], [S x R] the minibatch starts from (:231) ...
  float *next_c, *next_r;                            // ... and where c(T), r(T) go (:331): double-buffered by the engine, so that a
                                                     // minibatch whose persistent launch gave up can be run again from the same state
  const float4 *pk_gates, *pk_proj;                  // packed (MFMA-operand-ordered) weight copies, or null
  const float4 *pk_fold;                             // packed [W_rm | W_x] of the folded recurrence, or null
  bool fat;                                          // allow the 64-row x 32-stream kernels when S > 16
  bool bf16;                                         // pk_* hold bf16 operands; activations are rounded to bf16 when staged
  const unsigned short *wr_bf16 = nullptr;           // (or null) W_gifo_r rounded to bf16, natural [4C x R] layout: the fold product's operand plane
                                                     // (klstm_fold3.hip split mode 3), valid whenever the bf16 W_rm is
};

struct BwdPtrs {
  const float *wrT, *wmT, *wxT;                      // transposed copies [R x 4C], [C x R], [I x 4C]
  const float *wr_nat = nullptr, *wx_nat = nullptr;  // the natural W_gifo_r [4C x R], W_gifo_x [4C x I] (the persistent BPTT launch's tail workgroups read these)
  const float *pi, *pf, *po;
  const float *gifo, *cc, *hh;
  float *dgifo, *dc, *dr;
  unsigned short *dgifo_h = nullptr;                 // (or null) bf16 copy (RNE) of the dgifo rows, written by the per-XCD BPTT chain next to them (Nt2Job::Ah)
  float *dr_part;                                    // split-K slabs [KS][S][R]
  float *dx_part;                                    // split-K slabs [KS][S][I] (in_diff of one frame)
  int ks;                                            // number of slabs
  const float4 *pk_dr, *pk_dm;                       // packed weight copies, or null
  const float4 *pk_fold;                             // packed W_rm^T (4-row geometry) of the folded recurrence, or null
  const float4 *pk_fold_gates; int nch_gates;        // the gates-order operand [W_rm | W_x] (FwdPtrs::pk_fold) and its chunks per row tile: the persistent
                                                     // backward launch gathers its columns of W_rm from it (the 4-row array is then not needed at all)
  bool fat;
  bool bf16;
};

// optional per-launch timing through hipExtLaunchKernelGGL start/stop events
struct LaunchProbe { hipEvent_t start = nullptr, stop = nullptr; };

// Range guard of the fp16-plane products (klstm_math.h nonfinite_probe): a wave whose accumulators came out non-finite -- an operand
// beyond the fp16 range -- recomputes its outputs in plain fp32 and counts the event in a host-mapped word (pinned, portable; readable
// on the host without a synchronisation).  The words and what follows from them are state of a RangeGuard: every engine has its own
// (the launches an engine makes run with the engine's guard current on the calling thread), the stateless klstm_affine_* calls use
// the calling device's default guard -- one engine's overflow does not change another engine's kernels.  A product family whose
// word moved runs on its fp32-range kernel (the fold product: three bf16 planes) for a cool-down of `cool_len` looks, then the
// fp16 planes are tried again; a trigger that follows a re-arm closely doubles the cool-down (up to 2^20), a clean run as long
// as the last cool-down resets it -- like the persistent chain's give-up cool-down.  klstm_set_option "fp16_products" = 0 switches a
// guard off (all families on their fp32-range kernels), = 1 clears it.
enum { REDO_FOLD = 0, REDO_NT = 1, REDO_OUTER = 2, REDO_SKINNY = 3, REDO_WORDS = 8 };
struct RangeGuard;
RangeGuard *range_guard_create();
void range_guard_destroy(RangeGuard *g);
RangeGuard *range_guard_exchange(RangeGuard *g);     // make g the calling thread's current guard (null: the device default); returns the previous one
struct RangeGuardScope {                            // RAII form of the above
  RangeGuard *prev;
  explicit RangeGuardScope(RangeGuard *g) : prev(range_guard_exchange(g)) {}
  ~RangeGuardScope() { range_guard_exchange(prev); }
  RangeGuardScope(const RangeGuardScope &) = delete;
  RangeGuardScope &operator=(const RangeGuardScope &) = delete;
};
void range_guard_reset(RangeGuard *g, bool enabled);  // g null: the current device's default guard.  Counters, latches and cool-downs cleared
long range_guard_events(RangeGuard *g, int which);    // events counted so far (g null: the current device's default guard)
void range_guard_set_note(void (*fn)(const char *));  // where "family X met an operand beyond the fp16 range" / "re-armed" remarks go
unsigned *redo_counters();          // device-visible address of the current guard's REDO_WORDS words, or nullptr (then nothing is counted)
unsigned redo_count(int which);     // one LOOK at a family of the current guard: non-zero = keep to the fp32-range kernel (new events, a
                                    // cool-down in progress -- this look counts towards it --, or the guard is switched off)

// forward step t (1..T).  fuse_x: the x_t * W_gifo_x^T + bias term (...streams.h:246,:259) is
// contracted inside the step kernel (x = in rows of frame t); otherwise gifo already holds it.
// The state bridge (:231, :331) is folded in: t==1 reads prev_c/prev_r and mirrors them into
// time block 0, t==T writes c back to prev_c.
hipError_t launch_gates_step(const Dims &d, const FwdPtrs &p, int t, bool fuse_x, const float *in,
                             int in_stride, hipStream_t st, LaunchProbe pr = {}, bool fold = false);

// Folded recurrence (NumStream <= small_max; see the FOLDED RECURRENCE section of klstm_kernels.hip):
//   launch_fold     W_rm = W_gifo_r * W_r_m once per Update, written straight into the two packed operand arrays
//   launch_gates_step(..., fold = true)   a(t) = W_x x(t) + b + W_rm m(t-1), t >= 2
//   launch_rbatch   r(1..T) = m(1..T) W_r_m^T -> rr rows, out rows, prev_r
//   launch_dmf_step d_m(t) = P(t) + dgifo(t+1) W_rm with P = out_diff W_r_m, then the elementwise BPTT (:411-440)
// the fold product on wave-owned 32 x 80 (or 64) tiles without LDS staging (klstm_fold.hip); launch_fold uses it when supported
bool fold_direct_supported(const Dims &d);
void set_fold_direct(int v);
hipError_t launch_fold_direct(const Dims &d, const float *wr, const float *wmT, float *pk_fold[2], int nch1, int nch2,
                              hipStream_t st, LaunchProbe pr = {});
// Cm = beta*Cm + A^T B and P -= lr*Cm in one pass (N, ldc % 4 == 0, 16-byte aligned Cm and P)
hipError_t launch_gemm_tn_update(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float beta, float *Cm,
                                 float *P, int ldc, float lr, hipStream_t st, LaunchProbe pr = {});
// klstm_outer.hip: G = A^T B for few frames (K <= 96) and a wide result on the f16 matrix cores at fp32 accuracy, + column sums of A
bool outer_f16_supported(int M, int N, int K, const float *diff, int ldd, const float *x, int ldx, const float *Cm, int ldc,
                         const float *P, const float *bias);
hipError_t launch_outer_f16(int M, int N, int K, const float *diff, int ldd, const float *x, int ldx, float beta, float *Cm, int ldc,
                            float *P, float lr, float beta_b, float *bias, float *bias_p, float lr_b, hipStream_t st, LaunchProbe pr = {});
void set_outer_f16(int on);
int skinny16_pair_groups(int M, int N1, int N2, int K, int max_groups);       // klstm_fold.hip: two skinny products in one launch
hipError_t launch_skinny16_pair(int M, int K, const float *A1, const float *A2, int lda, const float *B1, int N1, const float *B2, int N2,
                                float *ws1, float *ws2, int G, hipStream_t st, LaunchProbe pr = {});
void set_skinny_f16_pair(int on);
void set_skinny_f16(int on);           // klstm_fold.hip: 0 = in_diff of a wide layer on the fp32 MFMA kernel (k_skinny_nn), A-B
hipError_t launch_gemm_tn_coal(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float beta, float *Cm,
                               int ldc, hipStream_t st, LaunchProbe pr = {});
// C = A B^T + bias for up to 80 rows and many columns, operands straight into MFMA registers (klstm_fold.hip)
void set_direct_nt_shape(int ni, int waves);
bool direct_nt_supported(int M, int N, int K, const float *A, int lda, const float *B, int ldb);
hipError_t launch_direct_nt(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *Cm, int ldc,
                            const float *bias, hipStream_t st, LaunchProbe pr = {});
// the same product as six bf16 MFMA products of three-way split operands (fp32 accuracy; klstm_fold3.hip); scratch holds the planes
// mode (per engine, option "fold_bf16x3"): 0: fp32 MFMA kernel, 1: bf16 x 3 planes (six products), 2: fp16 x 2 planes (three products)
bool fold_bf16x3_supported(const Dims &d, int mode);
void set_fold_bf16x3(int v);       // the default mode of engines created from now on
int fold_default_mode();
size_t fold_bf16x3_scratch_bytes(const Dims &d);
hipError_t launch_fold_bf16x3(const Dims &d, int mode, const float *wr, const float *wmT, void *scratch, float *pk_fold[2], int nch1,
                              int nch2, hipStream_t st, LaunchProbe pr_split = {}, LaunchProbe pr = {}, bool planes_fresh = false);
                              // planes_fresh: the split pass is skipped (the fused Update wrote the planes: GradsUpdate::a3 / b3)
// the fold product of the bf16 operand mode (klstm_persist_ms.hip): one bf16 plane per operand (split mode 3), result as bf16 in
// logical-row order [4C x C]
hipError_t launch_fold_ms(const Dims &d, const float *wr, const float *wmT, void *scratch, unsigned short *wl, hipStream_t st,
                          LaunchProbe pr_split = {}, LaunchProbe pr = {}, bool planes_fresh = false,
                          unsigned short *wlT = nullptr);      // wlT: also the transpose, [C][4C logical rows] bf16 (klstm_persist_xl.hip backward)
void fold_bf16x3_planes(const Dims &d, void *scratch, unsigned short **a3, long *a_plane, unsigned short **b3, long *b_plane);
hipError_t launch_fold(const Dims &d, const float *param_blob, const float *wmT, float *pk_fold[2], bool pack_x,
                       hipStream_t st, LaunchProbe pr = {}, LaunchProbe pr2 = {}, void *scratch3 = nullptr, LaunchProbe pr3 = {},
                       bool planes_fresh = false, int mode3 = 2);
                       // scratch3 (fold_bf16x3_scratch_bytes) selects the bf16x3 kernel when it supports the shape; pr3 = its split pass
                       // pk_fold zero-filled once by the caller; pack_x = false: launch_pack(.., foldx) already wrote the W_x chunks
hipError_t launch_rbatch(const Dims &d, const FwdPtrs &p, float *out, int out_stride, float *ws, hipStream_t st,
                         LaunchProbe pr = {}, LaunchProbe pr2 = {}, const unsigned *guard = nullptr);
                         // ws: split-K workspace (gemm_splitk_plan(T*S, R, C) slices); guard: the engine's control words -- behind a
                         // persistent launch that gave up ([2] | [6] != 0) the kernel that writes out / the carried r does nothing
hipError_t launch_dmf_step(const Dims &d, const BwdPtrs &p, int t, const float *P, hipStream_t st, LaunchProbe pr = {});
//   launch_bwd_tail d_r(1..T) = out_diff + dgifo(2..T+1) W_gifo_r and in_diff = dgifo(1..T) W_gifo_x, one split-K launch pair
size_t bwd_tail_ws_floats(const Dims &d);
hipError_t launch_bwd_tail(const Dims &d, const float *dgifo, const float *wr, const float *wx, const float *out_diff,
                           int od_stride, float *dr, float *in_diff, int id_stride, float *ws, hipStream_t st,
                           LaunchProbe pr = {}, LaunchProbe pr2 = {});
void pack_sizes_fold(const Dims &d, long n4[2]);   // float4 counts of the two folded operands
// r(t) = m(t) W_r_m^T (:312) -> rr, out rows (:328); t==T also prev_r (:331)
hipError_t launch_proj_step(const Dims &d, const FwdPtrs &p, int t, float *out, int out_stride,
                            hipStream_t st, LaunchProbe pr = {});
// backward step: partial d_r(t) = DGIFO(t+1) W_gifo_r (:391) as KS slabs; if in_diff != nullptr
// also the partial in_diff of frame t+1 = DGIFO(t+1) W_gifo_x (:457).  t may be 0 (x part only,
// written straight to in_diff rows of frame 1, single slice).
hipError_t launch_dr_step(const Dims &d, const BwdPtrs &p, int t, float *in_diff, int id_stride,
                          hipStream_t st, LaunchProbe pr = {});
// d_r(t) = out_diff(t) + slabs; d_m = d_r W_r_m (:408); elementwise BPTT (:411-440); also reduces
// the in_diff slabs of frame t+1 when in_diff != nullptr and t < T.
hipError_t launch_dm_step(const Dims &d, const BwdPtrs &p, int t, const float *out_diff, int od_stride,
                          float *in_diff, int id_stride, hipStream_t st, LaunchProbe pr = {});

// Generic batched GEMM  C[MxN] = beta*C + op(A)*op(B) (+ bias[n]).
//   transA: A stored [K x M] (lda)   else [M x K]
//   transB: B stored [N x K] (ldb)   else [K x N]
hipError_t launch_gemm(bool transA, bool transB, int M, int N, int K, const float *A, int lda,
                       const float *B, int ldb, float beta, float *Cm, int ldc, const float *bias,
                       hipStream_t st, LaunchProbe pr = {});

// Split-K form of the same product for few-tile / long-K shapes: ks = gemm_splitk_plan(...) slices of klen, partial
// tiles in ws (ks*M*N floats), summed in fixed order by a second kernel.  ks == 1: use launch_gemm.
int gemm_splitk_plan(int M, int N, int K, int *klen);
hipError_t launch_gemm_splitk(bool transA, bool transB, int M, int N, int K, const float *A, int lda, const float *B,
                              int ldb, float beta, float *Cm, int ldc, const float *bias, float *ws, int ks, int klen,
                              hipStream_t st, const float *add = nullptr, int add_ld = 0, LaunchProbe pr = {},
                              LaunchProbe pr2 = {}, float *C2 = nullptr, int ldc2 = 0, float *C3 = nullptr, int tail0 = 0,
                              const unsigned *guard = nullptr);
                              // add: C = beta*C + add + sum of slices;  C2 / C3: mirrors of the result (second copy; rows >= tail0)

// All seven gradient accumulations (...streams.h:468-487) in ONE launch: three A^T*B products
// (w_gifo_x, w_gifo_r, w_r_m) plus the bias / peephole column sums.  dst = beta*dst + grad, dst is a
// blob in GetParams order.
// upd: fold the Update (:504-512) into the same pass -- dst_blob must then be the momentum blob: dst = beta*dst + grad,
// clipped if clip > 0, params -= lr*dst, and the three transposed copies are written from the updated parameters (the C % 4,
// R % 4 row quads of a tile: shapes with C, R multiples of 4).  fp32 tiles only (launch fails for the bf16 tile path).
struct GradsUpdate {
  float *params; float lr, clip; float *wrT, *wmT, *wxT;
  // optional: the bf16 planes of the fold operands (fold_bf16x3_planes) are written from the updated W_gifo_r / W_r_m too
  unsigned short *a3 = nullptr, *b3 = nullptr; long a_plane = 0, b_plane = 0;
  int split_mode = 1;      // the engine's fold mode: 1 = three bf16 planes, 2 = two fp16 planes
  unsigned short *wrTh = nullptr, *wxTh = nullptr;   // (or null) bf16 copies (RNE) of the refreshed wrT / wxT, same layouts (Nt2Job::Bh)
  bool no_wT32 = false;    // bf16 tiles only: full tiles leave the fp32 wrT / wxT out (only the bf16 copies have a reader); the caller treats them as stale
};
// C = A B for few rows, a narrow result and a long contraction (klstm_fold.hip: the output layer's in_diff); ws holds one partial per K slice
bool skinny_nn_supported(int M, int N, int K, const float *A, int lda, const float *B, int ldb, const float *Cm, int ldc);
size_t skinny_nn_workspace_floats(int M, int N, int K);
hipError_t launch_skinny_nn(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *Cm, int ldc, float *ws,
                            hipStream_t st);
// C = A B^T + bias on the bf16 pipe (both operands rounded, fp32 accumulate): the batched x-projection of the bf16 operand mode
bool gemm_bf16_nt_supported(int M, int K, const float *A, int lda, const float *B, int ldb);
hipError_t launch_gemm_bf16_nt(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *Cm, int ldc,
                               const float *bias, hipStream_t st, LaunchProbe pr = {}, float *C2 = nullptr, int ldc2 = 0,
                               float *C3 = nullptr, int tail0 = 0, float beta = 0.f);   // C2: a second copy of the result; C3: rows >= tail0, dense (ld = N);
                                                                                        // beta: Cm = beta Cm + A B^T (+ bias)
hipError_t launch_gemm_bf16_nt_splitk(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float beta, float *Cm, int ldc,
                                      const float *add, int add_ld, float *ws, int ks, int klen, hipStream_t st, LaunchProbe pr = {},
                                      LaunchProbe pr2 = {});   // K in ks slices of klen (multiple of 64) through ws [ks][M x N]; C = beta C + add + A B^T
// klstm_gemm16.hip: the same product, pipelined (two K tiles in flight behind the one being contracted), split K inside the launch
// (last-arriving slice adds the slabs in slice order), K slice z on XCD z, one or two products that share their K per launch.
//   C = A B^T (+ bias[n]) (+ add[m][n])
struct Nt2Job {
  int M, N, K;
  const float *A; int lda;            // [M x K], k-contiguous rows, 16-byte aligned, lda % 4 == 0
  const float *B; int ldb;            // [N x K]
  float *C; int ldc;
  const float *bias;                  // [N] or null
  const float *add; int add_ld;       // [M x N] or null
  const unsigned short *Ah = nullptr, *Bh = nullptr;   // (or null) bf16 copies of A and B, RNE, same shapes and leading dimensions (in elements; % 8 == 0):
                                                       // given for every product of a launch, the kernel reads THEM (LDS-DMA, half the bytes) -- same bits out
};
struct Nt2Plan { int nj, ks, nt; size_t ws_floats; };       // 16-column blocks per wave (tile = 128 x 32 nj), K slices, output tiles, workspace
bool gemm_bf16_nt2_supported(const Nt2Job &g);
bool gemm_bf16_nt2_copies_usable(const Nt2Job &g);
Nt2Plan gemm_bf16_nt2_plan(const Nt2Job *jobs, int njobs, int force_nj = 0, int force_ks = 0);
hipError_t launch_gemm_bf16_nt2(const Nt2Job *jobs, int njobs, const Nt2Plan &pl, float *ws, size_t ws_floats, unsigned *tickets, int ntickets,
                                hipStream_t st, LaunchProbe pr = {});   // tickets: >= pl.nt words, zero between launches (the kernel leaves them zero)
void gemm_bf16_nt2_debug_buffer(long long *dev);      // probe support: 8 shader-clock sums per workgroup (null: off)
bool grads_bf16_tiles(const Dims &d, bool bf16);      // would launch_grads take the bf16 tile path?
// The reduction of the tail workgroups' partial d_r / in_diff rows (klstm_persist_bwd.hip: one partial row set per 32-cell slot, added in
// slot order) as a job: run by k_tail_reduce behind the BPTT launch, or -- "tail_merge" -- by the FIRST workgroups of the gradient launch
// that follows on the same stream (launch_grads `tr`): the launch of its own (4.3 us of dispatch + one round trip at 40/800/512) is gone,
// the W_r_m gradient tiles (the only readers of d_r) wait for an arrival counter, everybody else starts at once.
struct TailReduceJob {
  const float *tws = nullptr; int nslots = 0, T = 0, S = 0, R = 0, ncols = 0;   // partial rows [nslots][T*S][ncols]; ncols = R (+ I)
  const float *od = nullptr; int od_stride = 0;                                // out_diff: added to the d_r columns (:391), d_r(T) = out_diff(T) (:351)
  float *dr = nullptr; float *in_diff = nullptr; int id_stride = 0;
  unsigned *ctr = nullptr;        // merged form: [0] arrivals of the reduce workgroups (never reset: launch n waits for n * nred), [1] waits that expired
  unsigned seq = 0;               // merged form: ordinal of this launch among the engine's merged launches (1, 2, ...)
};
int tail_reduce_blocks(const TailReduceJob &j);   // reduce workgroups of the merged form (a multiple of 8)
hipError_t launch_tail_reduce(const TailReduceJob &j, const unsigned *guard, hipStream_t st, LaunchProbe pr = {});   // the launch of its own
hipError_t launch_grads(const Dims &d, const float *dgifo, const float *dr, const float *in, int in_stride,
                        const float *rr, const float *mm, const float *cc, float beta, float *dst_blob,
                        hipStream_t st, LaunchProbe pr = {}, bool bf16 = false, const GradsUpdate *upd = nullptr,
                        const unsigned *guard = nullptr, float *mark = nullptr,
                        const TailReduceJob *tr = nullptr);   // bf16: operands rounded to bf16, 16x16x32 MFMA, fp32 accumulate; tr: fp32 tiles only
// mark (data-parallel runs): one float behind the gradient blob that travels through the all-reduce with it -- launch_grads writes 0
// (this rank's gradient is real) or 1 (the guard stopped it); launch_update_repack / launch_apply_momentum given the same address AFTER
// the all-reduce leave everything alone when the sum is non-zero (and count it in *peer_skip): a minibatch that was invalid on ANY rank
// is left out by EVERY rank, without a host wait in front of the collective.

// Update (:504-512) + refresh of the transposed weight copies in ONE launch.
//   grad != nullptr : corr = mmt*corr + grad first (DP mode, after the all-reduce)
//   clip > 0        : corr clipped in place to +-clip (standard/ variant)
//   lr == 0 && !grad: pure repack (after klstm_set_params)
hipError_t launch_update_repack(const Dims &d, float *param_blob, float *corr_blob, const float *grad_blob,
                                float mmt, float lr, float clip, float *wrT, float *wmT, float *wxT,
                                hipStream_t st, LaunchProbe pr = {}, const unsigned *guard = nullptr,
                                const GradsUpdate *planes = nullptr, const float *mark = nullptr, unsigned *peer_skip = nullptr);   // planes->a3 / b3: also write the fold operands' bf16 planes (vector kernel only)
bool update_repack_vectorised(const Dims &d, const float *param_blob, const float *corr_blob, const float *grad_blob, const float *wrT,
                              const float *wmT, const float *wxT);
// guard (both): the two status words of the persistent chain (ctrl[2], ctrl[6]); when either is non-zero the kernels return
// without touching anything -- a minibatch whose chain gave up must not reach the momentum buffers or the parameters

// Packed weight copies for the vector kernels (shapes with R, I, C multiples of 8):
//   [0] gates [W_gifo_r | W_gifo_x]   [1] proj W_r_m   [2] dr [W_gifo_r^T ; W_gifo_x^T]   [3] dm W_r_m^T
bool pack_supported(const Dims &d);
void pack_sizes(const Dims &d, long n4[4]);         // float4 counts
// mask: bit i selects array i (forward operands = 3, BPTT operands = 12);  bf16: pack as bf16 (same tile/chunk/lane
// order, one 16-byte vector of 8 bf16 per lane and chunk -> half the bytes of the fp32 copies)
hipError_t launch_pack(const Dims &d, const float *param_blob, const float *wrT, const float *wmT, const float *wxT,
                       float *pk[4], int mask, bool bf16, hipStream_t st, LaunchProbe pr = {}, float *foldx = nullptr);
                       // foldx (with mask bit 0, fp32): also write the W_x chunks into the folded gates array

// out[dst] = in[clamp(dst + shift)] row gather (TimeShift; shift 0 = Transmit copy)
hipError_t launch_time_shift(const float *in, int rows, int cols, int in_stride, float *out, int out_stride, int shift,
                             hipStream_t st, LaunchProbe pr = {});

// output tail (Softmax, Xent::EvalMasked) and AffineTransform::Update helpers
hipError_t launch_softmax(const float *in, int rows, int cols, int in_stride, float *out, int out_stride, hipStream_t st);
hipError_t launch_xent(const float *y, int rows, int cols, int stride, const int *target, const float *mask, float *diff,
                       int diff_stride, float *row_xent, float *row_correct, hipStream_t st);
hipError_t launch_softmax_xent(const float *in, int rows, int cols, int in_stride, float *post, int post_stride, const int *target,
                               const float *mask, float *diff, int diff_stride, float *row_xent, float *row_correct, double *totals,
                               unsigned *ticket, hipStream_t st);
hipError_t launch_xent_accumulate(const float *row_xent, const float *row_correct, const float *mask, int rows, double *totals, hipStream_t st);
hipError_t launch_xent_post(const float *y, int rows, int cols, int stride, const int *post_off, const int *post_pdf, const float *post_w,
                            const float *mask, float *diff, int diff_stride, float *row_xent, float *row_ent, float *row_correct,
                            hipStream_t st);      // general (sparse) posteriors, nnet-loss.cc:76-142
hipError_t launch_col_sum(const float *src, int rows, int cols, int stride, float beta, float *dst, hipStream_t st);
hipError_t launch_axpy(float *y, const float *x, float a, long n, hipStream_t st);

hipError_t launch_sgd_momentum(float *param, float *corr, const float *grad, float mmt, float lr, long n, hipStream_t st);
hipError_t launch_apply_momentum(float *corr, const float *grad, float mmt, long n, hipStream_t st, LaunchProbe pr = {},
                                 const unsigned *guard = nullptr, const float *mark = nullptr);

// Weights-resident persistent chain (klstm_persist.hip forward, klstm_persist_bwd.hip backward; NumStream <= 8, folded
// recurrence, x term fused): ONE launch per direction runs all T steps with the packed fold operands held in registers and
// the per-step all-to-all done inside the launch through data-tagged granules.
//   gran: persist_gran_bytes(d) of device memory per direction, zero-filled once;  ctrl: 4 words per direction {epoch,
//   finished workgroups, status, pad}, zero-filled once.  status != 0 after the launch: a bounded wait expired
//   (0x80000000 | step) and the results of that launch are invalid.
// Per-engine knobs (A-B experiments and tests; 0 / -1 = defaults):
struct PersistOpts {
  int waves = 0, tpw = 0;         // forward: waves per workgroup (8, 12, 16), tiles of 4 cells per workgroup (1, 2)
  int nap0 = -1, nap = -1;        // sweepers sleep nap0 x 256 clocks before the first pass of a step, nap x 64 between passes
  int nap0_bwd = -1;              // the same for the backward launch
  int bwd_waves = 0;              // backward: 12 or 16 waves per workgroup
  int ncu = 0;                    // compute units of the device (0: unknown): the backward launch puts d_r / in_diff on workgroups of their own when
                                  // there are enough of them next to the chain's C / 4
  int tail_mode = -1;             // option "persist_tail": -1 / 1 d_r / in_diff inside the backward launch, on tail workgroups where they fit, else on
                                  // the chain's; 2 always on the chain's workgroups (rounds 3-5); 0 batched products after the launch
  int fwd_interleave = -1;        // forward, 5..8 streams: the two groups of 4 as interleaved chains (-1 / 1) or in lock-step (0: rounds 2-5)
  int bwd_interleave = -1;        // backward, 5..8 streams: the two groups of 4 as interleaved chains (-1 / 1) or one after the other (0)
  int xl = -1;                    // many streams, bf16, C = 1024: one chain per XCD (klstm_persist_xl.hip; -1 / 1) or klstm_persist_ms.hip (0)
  int xl_bwd = -1;                // ... and the BPTT chain the same way (-1 / 1) or one launch per step (0)
  long long spin_limit = 0;       // wall-clock ticks (100 MHz) a single in-kernel wait may take (0 = 50 ms)
  int test_stall_fwd = 0, test_stall_bwd = 0;   // test hook: workgroup 0 withholds its publish of this step -> timeout path
  unsigned *hstat = nullptr;      // host-mapped status word: set by a launch that gives up (the engine polls it without a sync)
  unsigned *guard = nullptr;      // the engine's control words: a launch that finds a status word ([2], [6]) set by an EARLIER launch does
                                  // nothing (whatever was queued behind a give-up leaves state, planes and parameters alone);
                                  // [8] counts the persistent launches of the engine that have run (both directions, in stream order):
                                  // the first launch that gives up records its ordinal ([8] + 1) in its ctrl[3]
  long long *dbg = nullptr;       // tools/persist_anatomy (KLSTM_PERSIST_TIMING builds only)
};
bool persist_supported(const Dims &d, const PersistOpts &o);       // forward
bool persist_x_batched(const Dims &d);   // wide input: the caller runs x W_gifo_x^T + bias as one batched product into the gifo plane first
bool persist_bwd_supported(const Dims &d, const PersistOpts &o);   // backward
int persist_fwd_grid(const Dims &d, const PersistOpts &o);         // workgroups that must be co-resident (one per CU)
int persist_bwd_grid(const Dims &d);
size_t persist_gran_bytes(const Dims &d);
bool persist_r_in_kernel(const Dims &d, const PersistOpts &o);   // r(t) = W_r_m m(t), the output rows and prev_r written by the forward launch (pass out)
hipError_t launch_fwd_persist(const Dims &d, const FwdPtrs &p, const float *in, int in_stride, float *out, int out_stride,
                              unsigned long long *gran, unsigned *ctrl, const PersistOpts &o, hipStream_t st, LaunchProbe pr = {});
bool persist_p_in_kernel(const Dims &d, const PersistOpts &o);   // P = out_diff W_r_m computed inside the backward launch (then P may be null)
bool persist_tail_in_kernel(const Dims &d, bool want_in_diff, const PersistOpts &o);   // d_r / in_diff contracted inside the backward launch
bool persist_tail_in_chain(const Dims &d, bool want_in_diff, const PersistOpts &o);    // ... on the chain's own workgroups
int persist_bwd_tail_wgs(const Dims &d, bool want_in_diff, const PersistOpts &o);      // ... on this many workgroups of their own (0: on the chain's)
// tws / tws_floats: workspace for the tail workgroups' partial rows (persist_bwd_tail_ws_floats; null: the chain's workgroups carry d_r / in_diff);
// pr_reduce: the k_tail_reduce launch behind the chain launch when tail workgroups ran
size_t persist_bwd_tail_ws_floats(const Dims &d, bool want_in_diff);
hipError_t launch_bwd_persist(const Dims &d, const BwdPtrs &p, const float *P, const float *out_diff, int od_stride,
                              float *in_diff, int id_stride, bool tail_inside, unsigned long long *gran, unsigned *ctrl,
                              const PersistOpts &o, hipStream_t st, LaunchProbe pr = {}, float *tws = nullptr, size_t tws_floats = 0,
                              LaunchProbe pr_reduce = {}, TailReduceJob *defer = nullptr);
// defer (or null): when tail workgroups ran, the reduction is NOT launched but described in *defer (tws != null says so): the caller hands
// it to the gradient launch that follows (launch_grads `tr`) or to launch_tail_reduce

// Many-stream (9..32) weights-resident forward chain of the bf16 operand mode (klstm_persist_ms.hip): one launch runs all T steps of
// the folded recurrence; wrm = W_gifo_r W_r_m as bf16, logical rows (4 cell + gate) x C (launch_fold_ms, once per Update); the x term must be in
// the gifo plane (batched product); r(1..T) -> rr plane, output rows and carried r come out of the same launch.
bool persist_xl_supported(const Dims &d, const PersistOpts &o);      // klstm_persist_xl.hip takes this launch (same arguments, same buffers)
size_t persist_xl_gran_bytes();
hipError_t launch_fwd_persist_xl(const Dims &d, const FwdPtrs &p, const unsigned short *wrm, float *out, int out_stride, void *gran, unsigned *ctrl,
                                 const PersistOpts &o, hipStream_t st, LaunchProbe pr = {});
size_t persist_xl_bwd_gran_bytes();
hipError_t launch_bwd_persist_xl(const Dims &d, const BwdPtrs &p, const unsigned short *wrmT, const float *P, void *gran, unsigned *ctrl,
                                 const PersistOpts &o, hipStream_t st, LaunchProbe pr = {});   // wrmT: launch_fold_ms wlT; P = out_diff W_r_m [T*S x C]
bool persist_ms_supported(const Dims &d);
int persist_ms_grid(const Dims &d);
size_t persist_ms_gran_bytes(const Dims &d);
hipError_t launch_fwd_persist_ms(const Dims &d, const FwdPtrs &p, const unsigned short *wrm, float *out, int out_stride, uint4 *gran, unsigned *ctrl,
                                 const PersistOpts &o, hipStream_t st, LaunchProbe pr = {});

int get_small_max();
void set_fat_fine(int v);       // A-B knob: half-size row tiles in the many-stream kernels (-1 auto, 0, 1)
void set_small_nt2(int v);     // A-B knob: two stream groups per workgroup at 5..small_max streams
void set_small_max(int s);        // tuning knob: largest NumStream that uses the 4x4x1_16b geometry
int dr_split_k(const Dims &d);   // number of split-K slabs launch_dr_step writes

}  // namespace klstm
