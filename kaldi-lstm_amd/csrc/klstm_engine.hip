// kaldi-lstm_amd/csrc/klstm_engine.hip -- engine object behind the C-ABI of include/klstm.h.
// Owns parameters / gradient / momentum blobs, carried stream state and activation planes in
// HBM, sequences the step kernels for one BPTT minibatch and replays that sequence from a
// hipGraph (the reference issues ~665 tiny launches per minibatch; here a
// minibatch is 2 graph launches of ~2T+8 kernels each).
//
// There is deliberately NO CPU fallback: without a usable gfx950 device klstm_create fails with
// KLSTM_ERR_NOGPU.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>          // types only: the library is resolved at run time (see the RCCL section below)

#include "../../include/klstm.h"
#include "klstm_kernels.h"

using namespace klstm;
void klstm_oneshot_set_abort_words(klstm_oneshot *h, unsigned *guard_word_dev, unsigned *host_mapped_word);   // klstm_oneshot.hip
long klstm_oneshot_floats(klstm_oneshot *h);                                                                   // klstm_oneshot.hip

static thread_local std::string g_err;
static klstm_status fail(klstm_status st, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return st;
}
// a remark for klstm_last_error() that is not a failure (why a faster path was not taken)
static void note(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(KLSTM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

struct ProbeRec { std::string name; hipEvent_t start, stop; };

// The calls of the current minibatch -- since the last klstm_propagate / klstm_reset BEGAN; from then on the caller may reuse the
// previous minibatch's buffers -- with the arguments they came with: what the engine re-runs on the launch-per-step chain when
// a persistent launch of this minibatch gave up (recover()).
struct MbRec {
  bool have_fwd = false, have_bwd = false, have_upd = false;
  bool have_ar = false;                       // the gradient all-reduce of this minibatch has been enqueued (with the validity word: klstm_allreduce_grads)
  unsigned fwd_seq = 0, bwd_seq = 0;          // ordinal of the call's persistent launch (0: it did not use one)
  int sp_before = 0;                          // state buffer the forward started from
  const float *in = nullptr; int rows = 0, in_stride = 0; float *out = nullptr; int out_stride = 0;
  const float *bin = nullptr; int bin_stride = 0; const float *od = nullptr; int od_stride = 0; float *idf = nullptr; int id_stride = 0;
  float mmt = 0.f; int flags = 0;
  float lr = 0.f, clip = 0.f;
};
static const klstm_status KLSTM_RECOVERED = (klstm_status)100;   // internal: a give-up was found and answered (never leaves the library)

struct klstm_engine {
  int I = 0, C = 0, R = 0, S = 0, device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  long nparams = 0;
  long ar_len = 0;              // floats of the engine's OWN gradient blob: nparams rounded up to 4, + 4 of which the first is the validity
                                // word of data-parallel runs (klstm_kernels.h launch_grads `mark`)
  bool ar_marked = false;       // the blob went through an all-reduce WITH its validity word since the last gradient products
  float *params = nullptr, *grads = nullptr, *corr = nullptr;
  float *grads_own = nullptr;   // the engine's own gradient blob (grads points elsewhere after klstm_bind_grad_blob)
  float *wrT = nullptr, *wmT = nullptr, *wxT = nullptr;   // transposed copies for the BPTT kernels
  unsigned short *wrTh = nullptr, *wxTh = nullptr;        // bf16 copies (RNE) of wrT / wxT, written by the Update kernels of the many-stream bf16 mode
  bool wth_fresh = false;                                 // ... and whether they are the roundings of the CURRENT W_gifo_r^T / W_gifo_x^T
  bool wT32_stale = false;                                // the fp32 wrT / wxT were left out by the last Update (only their bf16 copies have a reader while the
                                                          // per-XCD chains run): whoever needs them -- a launch-per-step chain, k_pack, the fp32-operand product --
                                                          // refreshes them first (ensure_wT32)
  unsigned short *dgifo_h = nullptr;                      // bf16 copy of the dgifo rows, written by the per-XCD BPTT chain ((T_alloc + 2) S x 4C)
  long tail_wgs = 0;                                      // tail workgroups of the last persistent backward launch (klstm_profile_query "persist_tail_wgs")
  long n_copies = 0;                                      // launches of the batched products that read the bf16 copies (klstm_profile_query "gemm_copies_launches")
  int skip_wT32 = 1;                                      // (part of "gemm_copies": 2 = copies without leaving the fp32 ones out; A-B runs)
  int copies_plan = 0;                                    // option "gemm_copies_plan" (A-B runs): 16 nj + ks forced on the launches that read the copies (0: the planner)
  int use_copies = 1;                                     // option "gemm_copies": d_r + in_diff read the bf16 copies (klstm_gemm16.hip, LDS-DMA form)
  float *pk[4] = {nullptr, nullptr, nullptr, nullptr};    // packed MFMA-operand-ordered copies (vector kernels)
  // carried state, double-buffered: a forward pass reads [sp] and writes [sp ^ 1], then the engine flips sp -- a persistent
  // launch that gives up leaves the state it started from intact, and so does everything queued behind it (device-side guard)
  float *prev_c[2] = {nullptr, nullptr}, *prev_r[2] = {nullptr, nullptr};
  int sp = 0;
  int *flags_dev = nullptr;
  float *stage[4] = {nullptr, nullptr, nullptr, nullptr};   // host-matrix staging: in, out, out_diff, in_diff (dense rows)
  int stage_rows = 0;
  // activation planes, (T_alloc+2) time blocks each
  int T_alloc = 0;
  float *gifo = nullptr, *cc = nullptr, *hh = nullptr, *mm = nullptr, *rr = nullptr;
  float *dgifo = nullptr, *dc = nullptr, *dr = nullptr, *dr_part = nullptr, *dx_part = nullptr;
  int ks = 1;
  int T_fwd = -1;     // T of the last propagate (-1: none yet)
  int T_bwd = -1;
  int use_graph = 0;        // option "graph": 0 plain launches (default), 1 a hipGraph per call unless the call is one or two launches, 2 always
  bool mmt_pending = false;   // DP: corr = mmt*corr + grads is folded into the next Update
  // KLSTM_BPTT_FUSE_UPDATE: the gradient products of the last backpropagate wait for klstm_update (or for anything that looks)
  bool grads_pending = false;
  const float *gp_in = nullptr; int gp_in_stride = 0, gp_T = 0; float gp_mmt = 0.f; bool gp_bf16 = false;
  float mmt_value = 0.f;
  bool use_vector = true;
  bool use_fat = true;
  int use_fold = -1;       // folded recurrence (W_rm = W_gifo_r W_r_m): -1 auto, 0 off, 1 on (whenever the shape allows)
  bool foldx_fresh = false; // the W_x chunks of pk_fold[0] were written by the last k_pack
  bool fold_dirty = true;  // W_rm / its packed copies are older than the parameters
  bool pk2_fresh = false;  // the last fold product also wrote the 4-row array W_rm^T (pk_fold[1]: only the launch-per-step folded BPTT reads it)
  int pk_stale = 0;        // unfolded operand arrays (bits 1..3) not refreshed by the last Update because the folded path is in use
  bool fwd_folded = false; // the last propagate ran the folded chain (its backpropagate follows suit)
  int use_persist = -1;    // weights-resident persistent chain (klstm_persist.hip): -1 auto (both directions, from 8 frames per
                           // stream), 0 off, 1 forward only, 2 forward and backward whenever the shape allows
  bool fwd_persist = false; // the last propagate ran inside one persistent launch
  bool fwd_ms = false;      // ... the many-stream bf16 one (klstm_persist_ms.hip): batched x term, launch, batched projection
  unsigned short *wrm_l = nullptr;   // W_rm = W_gifo_r W_r_m as bf16, logical rows x C (a bf16 product, refreshed after every Update) for that launch
  uint4 *gran_ms = nullptr; // its granule slots
  bool bwd_xl = false;      // ... and its BPTT as one chain per XCD (klstm_persist_xl.hip k_bwd_persist_xl)
  unsigned short *wrmT_l = nullptr;   // W_rm^T as bf16 [C][4C logical rows] (written by the fold product next to wrm_l)
  void *gran_xb = nullptr;  // the backward chain's granule slots
  PersistOpts popt;         // per-engine knobs of the persistent kernels (options persist_waves, persist_tpw, persist_nap*, ...)
  int ncu = 0;              // compute units of the device: every workgroup of a persistent launch needs one of its own
  int persist_tail = 1;     // option "persist_tail": d_r / in_diff inside the persistent backward launch (0: batched products after it)
  // option "tail_merge" (default 0: measured slower): the reduction of the tail workgroups' partial rows runs on the first workgroups of the
  // gradient launch that follows the BPTT launch (k_grads_tm) instead of in a launch of its own.  With KLSTM_BPTT_FUSE_UPDATE that launch
  // is klstm_update's: the job waits with the gradient products (tail_pending; whoever flushes them, or synchronises, runs it).
  // tools/ab_step.py, 40/800/512: 4 streams k_tail_reduce 4.3 + k_grads 14.4 = 18.7 us apart, 20.0 us merged (146.0 -> 148.1 us per
  // minibatch); 8 streams 4.4 + 18.2 -> 27.3 (189.9 -> 195.0).  Every gradient tile is resident from the start and latency-bound (operands,
  // old corr / parameters, stores: three dependent memory round trips); the W_r_m tiles start one cross-XCD hand-over later (partial rows ->
  // write-through d_r -> acknowledgement -> counter -> poll -> sc1 loads, ~6 us) and the launch ends that much later.  Bit-identical (test).
  int tail_merge = 0;
  unsigned *tr_ctr = nullptr;   // device: [0] arrivals of reduce workgroups over the engine's life, [1] expired waits
  unsigned tr_seq = 0;          // merged launches enqueued so far
  bool tail_pending = false;
  TailReduceJob tr_job;
  bool bwd_persist = false; // ... and its backpropagate runs steps T..1 inside one persistent launch
  bool persist_dirty = false;   // a persistent launch ran since the status words were last read back
  unsigned long long *gran[2] = {nullptr, nullptr};   // granule slots of the forward / backward chain
  unsigned *pctrl = nullptr;    // 2 x 4 words: {epoch, finished workgroups, status, pad} per direction
  unsigned *pstat_host = nullptr;   // pinned, device-mapped word the persistent kernels set when they give up (polled without a sync)
  // ---- give-up handling of the persistent chain (recover()) ----
  unsigned pseq = 0;            // persistent launches enqueued so far, both directions (the device counts the same in pctrl[8])
  int persist_verify = 0;       // option: wait for every persistent launch and answer a give-up before the call returns
  int verify_spin = 1;          // option "persist_verify_spin": that wait spins on the host-mapped done word (0: hipStreamSynchronize)
  bool verify_later = false;    // "persist_verify": the wait for this minibatch's BPTT launch has been moved behind the Update's launches
  bool verify_later_reports = false;   // (KLSTM_BPTT_FUSE_UPDATE: klstm_update follows immediately -- verify_deferred())
  int cooldown = 0, cooldown_len = 64;   // minibatches on the launch-per-step chain after a give-up, then the persistent chain again
  int cooldown_cur = 0;                  // this give-up's cool-down: doubles with every give-up that follows a re-arm closely (a co-tenant that
                                         // stays would cost a spin limit + a re-run every cooldown_len minibatches), back to cooldown_len
                                         // after as many clean persistent minibatches as the last cool-down was long
  long clean_run = 0;                    // persistent minibatches since the last give-up
  long n_giveups = 0, n_replayed = 0, n_dropped = 0;
  bool replaying = false;
  // every persistent forward launch nobody has looked at yet: which state buffer it started from and the Reset flags the caller issued
  // between the forward pass before it and this one (empty: none).  Pruned when a host synchronisation finds the status words clean;
  // a caller that never looks is made to every MARKS_MAX minibatches (klstm_propagate).
  struct FwdMark { unsigned seq; int sp_before; std::vector<int> resets; };
  std::deque<FwdMark> marks;
  static constexpr size_t MARKS_MAX = 4096;
  MbRec rec;
  std::vector<int> resets;      // Reset flags since the last forward pass was enqueued (applied again when the state goes back a buffer)
  int fold_mode = 2;            // fold product, as asked for: 0 fp32 MFMA, 1 three bf16 planes, 2 two fp16 planes (option "fold_bf16x3")
  int fold_eff = 2;             // ... as it runs: 1 instead of 2 while this engine's range guard keeps the fold product off the fp16 planes
  RangeGuard *rg = nullptr;     // this engine's range guard (klstm_kernels.h): its fp16-plane products, their event words and cool-downs
  unsigned knob_gen = 0;        // the process-wide A-B knobs' generation this engine's cached graphs were captured under
  void *fold_scratch = nullptr;             // bf16 planes of the two fold operands (klstm_fold3.hip)
  bool planes_fresh = false;                // ... and they were written from the current parameters (by the fused Update)
  float *pk_fold[2] = {nullptr, nullptr};   // packed [W_rm | W_x] (gates order) and W_rm^T (4-row geometry)
  float *Pm = nullptr;     // out_diff * W_r_m for all frames [(T_alloc) S x C]
  float *ws = nullptr;     // split-K workspace of the batched d_r / in_diff products
  size_t ws_floats = 0;
  unsigned *tickets = nullptr;   // klstm_gemm16.hip: one word per output tile of a split-K launch (NT2_TICKETS words, zero between launches)
  int fuse_update_ok = 1;  // option "fuse_update": 0 = KLSTM_BPTT_FUSE_UPDATE is ignored (gradient products and Update as separate passes; A-B runs)
  int use_nt2 = 1;         // option "gemm_nt2": the batched bf16 products on the pipelined kernel (0: round 4's kernel + reduction launches)
  bool use_bf16 = false;   // bf16 operands in the step kernels (option "bf16"); masters/planes/gradients stay fp32
  int fuse_x = -1;    // -1 auto (small NumStream), 0 batched x-projection GEMM, 1 fused into the step kernel
  bool profile = false;
  std::vector<ProbeRec> probes;
  std::vector<hipEvent_t> event_pool;   // recycled by klstm_profile_query: no hipEventCreate on the launch path after the first pass
  std::map<std::string, std::pair<double, long>> prof;   // name -> (total us, launches)
  typedef std::tuple<int, const void *, int, const void *, int, const void *, int, float, int> Key;
  std::map<Key, hipGraphExec_t> graphs;

  // offsets into a blob, GetParams order
  long o_wx() const { return 0; }
  long o_wr() const { return (long)4 * C * I; }
  long o_b() const { return o_wr() + (long)4 * C * R; }
  long o_pi() const { return o_b() + 4 * C; }
  long o_pf() const { return o_pi() + C; }
  long o_po() const { return o_pf() + C; }
  long o_wm() const { return o_po() + C; }
};

static LaunchProbe probe(klstm_engine *e, const char *name) {
  LaunchProbe pr;
  if (!e->profile) return pr;
  ProbeRec r;
  r.name = name;
  if (e->event_pool.size() >= 2) {
    r.start = e->event_pool.back(); e->event_pool.pop_back();
    r.stop = e->event_pool.back(); e->event_pool.pop_back();
  } else if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return pr;
  e->probes.push_back(r);
  pr.start = r.start; pr.stop = r.stop;
  return pr;
}

static void free_planes(klstm_engine *e) {
  float **ps[] = {&e->gifo, &e->cc, &e->hh, &e->mm, &e->rr, &e->dgifo, &e->dc, &e->dr, &e->dr_part, &e->dx_part, &e->Pm, &e->ws};
  for (float **p : ps) { if (*p) (void)hipFree(*p); *p = nullptr; }
  if (e->dgifo_h) { (void)hipFree(e->dgifo_h); e->dgifo_h = nullptr; }
}
static void drop_graphs(klstm_engine *e) {
  for (auto &kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
  e->graphs.clear();
}

// The process-wide A-B knobs (which kernel a product runs on: "direct_nt_shape", "skinny_f16", "outer_f16", ...) must reach the cached
// graphs of every live engine.  The caller's thread does not touch foreign engines (another thread may be driving them): it moves a
// generation counter on, and every engine compares at the head of its own launching calls and drops its own graphs.
static std::mutex g_engines_mu;
static std::set<klstm_engine *> g_engines;
static std::atomic<unsigned> g_knob_gen{0};
static void knob_changed_everywhere() { g_knob_gen.fetch_add(1, std::memory_order_relaxed); }
static klstm_status knobs_seen(klstm_engine *e) {
  const unsigned gen = g_knob_gen.load(std::memory_order_relaxed);
  if (gen == e->knob_gen) return KLSTM_OK;
  e->knob_gen = gen;
  if (!e->graphs.empty()) {
    if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(KLSTM_ERR_HIP, "hipStreamSynchronize failed");
    drop_graphs(e);
  }
  e->fold_dirty = true;            // ("fold_direct": the fold product may run on another kernel)
  return KLSTM_OK;
}

static klstm_status ensure_planes(klstm_engine *e, int T) {
  if (T <= e->T_alloc) return KLSTM_OK;
  HIPCHK(hipStreamSynchronize(e->stream));
  drop_graphs(e);                     // graphs bake plane addresses
  free_planes(e);
  const size_t nb = (size_t)(T + 2) * e->S;
  const Dims d{e->I, e->C, e->R, e->S, T};
  e->ks = dr_split_k(d);
  HIPCHK(hipMalloc(&e->gifo, nb * 4 * e->C * sizeof(float)));
  HIPCHK(hipMalloc(&e->cc, nb * e->C * sizeof(float)));
  HIPCHK(hipMalloc(&e->hh, nb * e->C * sizeof(float)));
  HIPCHK(hipMalloc(&e->mm, nb * e->C * sizeof(float)));
  HIPCHK(hipMalloc(&e->rr, nb * e->R * sizeof(float)));
  HIPCHK(hipMalloc(&e->dgifo, nb * 4 * e->C * sizeof(float)));
  HIPCHK(hipMalloc(&e->dgifo_h, nb * 4 * e->C * sizeof(unsigned short)));
  HIPCHK(hipMemsetAsync(e->dgifo_h, 0, nb * 4 * e->C * sizeof(unsigned short), e->stream));
  HIPCHK(hipMalloc(&e->dc, nb * e->C * sizeof(float)));
  HIPCHK(hipMalloc(&e->dr, nb * e->R * sizeof(float)));
  HIPCHK(hipMalloc(&e->dr_part, (size_t)e->ks * e->S * e->R * sizeof(float)));
  HIPCHK(hipMalloc(&e->dx_part, (size_t)e->ks * e->S * e->I * sizeof(float)));
  HIPCHK(hipMalloc(&e->Pm, (size_t)T * e->S * e->C * sizeof(float)));      // folded path: P = out_diff W_r_m plane
  e->ws_floats = 0;                                                        // split-K workspace: sized per call (ensure_ws)
  // kSetZero semantics of the reference slabs (...streams.h:230, :352)
  HIPCHK(hipMemsetAsync(e->gifo, 0, nb * 4 * e->C * sizeof(float), e->stream));
  HIPCHK(hipMemsetAsync(e->cc, 0, nb * e->C * sizeof(float), e->stream));
  HIPCHK(hipMemsetAsync(e->hh, 0, nb * e->C * sizeof(float), e->stream));
  HIPCHK(hipMemsetAsync(e->mm, 0, nb * e->C * sizeof(float), e->stream));
  HIPCHK(hipMemsetAsync(e->rr, 0, nb * e->R * sizeof(float), e->stream));
  HIPCHK(hipMemsetAsync(e->dgifo, 0, nb * 4 * e->C * sizeof(float), e->stream));
  HIPCHK(hipMemsetAsync(e->dc, 0, nb * e->C * sizeof(float), e->stream));
  HIPCHK(hipMemsetAsync(e->dr, 0, nb * e->R * sizeof(float), e->stream));
  e->T_alloc = T;
  return KLSTM_OK;
}

// Split-K workspace of the folded path's batched products (r, P, d_r + in_diff).  gemm_splitk_plan() takes MORE K slices
// when M = T*S is smaller, so ks*M*N is not monotonic in T: the need is computed for the T of THIS call, never inferred
// from T_alloc.
static size_t ws_need(const klstm_engine *e, int T) {
  const Dims d{e->I, e->C, e->R, e->S, T};
  int kl = 0;
  const size_t M = (size_t)T * e->S;
  size_t need = bwd_tail_ws_floats(d);                                                               // d_r + in_diff
  const size_t nr = (size_t)gemm_splitk_plan((int)M, e->R, e->C, &kl) * M * e->R;                    // r
  const size_t np = (size_t)gemm_splitk_plan((int)M, e->C, e->R, &kl) * M * e->C;                    // P
  if (nr > need) need = nr;
  if (np > need) need = np;
  const size_t nx = (size_t)8 * M * (e->R > e->I ? e->R : e->I);                                     // the per-XCD BPTT chain's d_r / in_diff in 8 K slices
  if (nx > need) need = nx;
  const size_t ntl = persist_bwd_tail_ws_floats(d, true);                                             // the persistent BPTT launch's tail workgroups: partial d_r / in_diff rows per 32-cell slot
  if (ntl > need) need = ntl;
  if (e->use_bf16 && M >= 256) {                                                                      // ... and on the pipelined kernel (klstm_gemm16.hip): padded tiles
    const Nt2Job j[2] = {Nt2Job{(int)M, e->R, 4 * e->C, nullptr, 4 * e->C, nullptr, 4 * e->C, nullptr, e->R, nullptr, nullptr, 0},
                         Nt2Job{(int)M, e->I, 4 * e->C, nullptr, 4 * e->C, nullptr, 4 * e->C, nullptr, e->I, nullptr, nullptr, 0}};
    const Nt2Job jp{(int)M, e->C, e->R, nullptr, e->R, nullptr, e->R, nullptr, e->C, nullptr, nullptr, 0};
    const Nt2Job jx{(int)M, 4 * e->C, e->I, nullptr, e->I, nullptr, e->I, nullptr, 4 * e->C, nullptr, nullptr, 0};
    const size_t cand[4] = {gemm_bf16_nt2_plan(j, 2).ws_floats, gemm_bf16_nt2_plan(j, 1).ws_floats, gemm_bf16_nt2_plan(&jp, 1).ws_floats,
                            gemm_bf16_nt2_plan(&jx, 1).ws_floats};
    for (size_t n : cand) if (n > need) need = n;
  }
  return need;
}
constexpr int NT2_TICKETS = 8192;
static klstm_status ensure_tickets(klstm_engine *e) {
  if (e->tickets) return KLSTM_OK;
  HIPCHK(hipMalloc(&e->tickets, NT2_TICKETS * sizeof(unsigned)));
  HIPCHK(hipMemsetAsync(e->tickets, 0, NT2_TICKETS * sizeof(unsigned), e->stream));
  return KLSTM_OK;
}
// one or two batched bf16 products on the pipelined kernel; false: not taken (shape / alignment / workspace), the caller falls back
static bool nt2_products(klstm_engine *e, const Nt2Job *jobs, int njobs, LaunchProbe pr, hipError_t *err) {
  if (!e->use_nt2 || !e->tickets) return false;
  for (int q = 0; q < njobs; q++) if (!gemm_bf16_nt2_supported(jobs[q])) return false;
  const bool copies = jobs[0].Ah != nullptr;
  const Nt2Plan pl = gemm_bf16_nt2_plan(jobs, njobs, copies ? e->copies_plan >> 4 : 0, copies ? e->copies_plan & 15 : 0);
  if (pl.ks > 1 && (pl.ws_floats > e->ws_floats || !e->ws || pl.nt > NT2_TICKETS)) return false;
  *err = launch_gemm_bf16_nt2(jobs, njobs, pl, e->ws, e->ws_floats, e->tickets, NT2_TICKETS, e->stream, pr);
  if (*err == hipErrorInvalidValue) {               // the launcher refused the plan before anything was enqueued (a forced split the shape
    (void)hipGetLastError();                        // cannot take, an empty slice): not taken, the caller's round-4 kernels run
    *err = hipSuccess;
    return false;
  }
  if (copies && *err == hipSuccess) e->n_copies++;
  return true;
}
static klstm_status ensure_ws(klstm_engine *e, int T) {
  const size_t need = ws_need(e, T);
  if (need <= e->ws_floats && e->ws) return KLSTM_OK;
  HIPCHK(hipStreamSynchronize(e->stream));
  drop_graphs(e);                     // captured launches hold the old address
  if (e->ws) (void)hipFree(e->ws);
  e->ws = nullptr; e->ws_floats = 0;
  HIPCHK(hipMalloc(&e->ws, need * sizeof(float)));
  e->ws_floats = need;
  return KLSTM_OK;
}

// The validity word of data-parallel runs: behind the engine's OWN gradient blob (a blob bound by the caller has no room for it).
static float *ar_mark(const klstm_engine *e) {
  float *own = e->grads_own ? e->grads_own : e->grads;
  return e->grads == own && e->ar_len ? own + e->ar_len - 4 : nullptr;
}
static const float *ar_mark_if_reduced(const klstm_engine *e) { return e->ar_marked ? ar_mark(e) : nullptr; }

// "tail_merge": the waiting reduction as an argument of the gradient launch that is about to be enqueued (null: none waits).  The launch
// ordinal is taken here: call it once per launch_grads, right in front of it.
static const TailReduceJob *take_tail_job(klstm_engine *e) {
  if (!e->tail_pending) return nullptr;
  e->tail_pending = false;
  e->tr_job.ctr = e->tr_ctr; e->tr_job.seq = ++e->tr_seq;
  return &e->tr_job;
}
// ... or in a launch of its own (somebody wants in_diff / d_r before any gradient launch: klstm_synchronize, a getter, set_corr)
static klstm_status flush_tail(klstm_engine *e) {
  if (!e->tail_pending) return KLSTM_OK;
  e->tail_pending = false;
  HIPCHK(launch_tail_reduce(e->tr_job, e->pctrl, e->stream, probe(e, "k_tail_reduce")));
  return KLSTM_OK;
}

// gradient products deferred by KLSTM_BPTT_FUSE_UPDATE, the ordinary way (somebody looks before the Update arrives)
static klstm_status flush_grads(klstm_engine *e) {
  if (!e->grads_pending) return flush_tail(e);
  e->grads_pending = false;
  const RangeGuardScope rgs(e->rg);
  const Dims d{e->I, e->C, e->R, e->S, e->gp_T};
  HIPCHK(launch_grads(d, e->dgifo, e->dr, e->gp_in, e->gp_in_stride, e->rr, e->mm, e->cc, e->gp_mmt, e->corr, e->stream,
                      probe(e, "k_grads"), e->gp_bf16, nullptr, e->pctrl, nullptr, take_tail_job(e)));
  if (e->verify_later) e->persist_dirty = true;      // (the promised klstm_update did not come: whoever synchronises next looks, as for any unverified launch)
  return KLSTM_OK;
}
static klstm_status flush_momentum(klstm_engine *e) {
  { klstm_status gs = flush_grads(e); if (gs != KLSTM_OK) return gs; }
  if (!e->mmt_pending) return KLSTM_OK;
  e->mmt_pending = false;
  HIPCHK(launch_apply_momentum(e->corr, e->grads, e->mmt_value, e->nparams, e->stream, probe(e, "k_apply_momentum"), e->pctrl,
                               ar_mark_if_reduced(e)));
  return KLSTM_OK;
}

// the fp32 transposed copies of W_gifo_r / W_gifo_x, if the last Update left them out (a pure transposition of the current parameters)
static klstm_status ensure_wT32(klstm_engine *e) {
  if (!e->wT32_stale) return KLSTM_OK;
  const Dims d{e->I, e->C, e->R, e->S, 0};
  HIPCHK(launch_update_repack(d, e->params, e->corr, nullptr, 0.f, 0.f, 0.f, e->wrT, e->wmT, e->wxT, e->stream, probe(e, "k_update_repack")));
  e->wT32_stale = false;
  return KLSTM_OK;
}
// The bf16 copies of W_gifo_r^T / W_gifo_x^T (the LDS-DMA operands of the per-XCD BPTT chain's d_r + in_diff) as roundings of the
// CURRENT parameters, by an UNGUARDED pure transposition.  The Update kernels write them too -- but an Update can decide ON THE
// DEVICE to do nothing (a give-up in front of it, a data-parallel peer's validity word), and the host cannot know: an Update that
// is the first to be asked for the copies (after klstm_create, set_params / sync_params, a "gemm_copies" toggle, a cool-down) is
// therefore preceded by this pass, so that the copies are valid whether or not it runs.  One extra launch, only then.
static klstm_status refresh_wth(klstm_engine *e) {
  const Dims d{e->I, e->C, e->R, e->S, 0};
  if (!e->wrTh || !e->wxTh || !update_repack_vectorised(d, e->params, e->corr, nullptr, e->wrT, e->wmT, e->wxT)) { e->wth_fresh = false; return KLSTM_OK; }
  GradsUpdate u{e->params, 0.f, 0.f, e->wrT, e->wmT, e->wxT};
  u.wrTh = e->wrTh; u.wxTh = e->wxTh;
  HIPCHK(launch_update_repack(d, e->params, e->corr, nullptr, 0.f, 0.f, 0.f, e->wrT, e->wmT, e->wxT, e->stream, probe(e, "k_refresh_wth"),
                              nullptr, &u, nullptr, nullptr));
  e->wT32_stale = false;
  e->wth_fresh = true;
  return KLSTM_OK;
}
static klstm_status repack(klstm_engine *e) {
  const Dims d{e->I, e->C, e->R, e->S, 0};
  e->wT32_stale = false;
  if (e->wth_fresh && !e->graphs.empty()) {           // (the bf16 copies of wrT / wxT come out of the Update kernels only; a cached graph may read them)
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
  }
  e->wth_fresh = false;
  HIPCHK(launch_update_repack(d, e->params, e->corr, nullptr, 0.f, 0.f, 0.f, e->wrT, e->wmT, e->wxT, e->stream,
                              probe(e, "k_update_repack")));
  if (e->pk[0]) HIPCHK(launch_pack(d, e->params, e->wrT, e->wmT, e->wxT, e->pk, 15, e->use_bf16, e->stream, probe(e, "k_pack")));
  e->fold_dirty = true;
  e->planes_fresh = false;
  e->foldx_fresh = false;
  e->pk_stale = 0;
  return KLSTM_OK;
}

static int g_d2h_small = 1;           // process-wide A-B knob ("d2h_small"): klstm_memcpy_d2h's small-copy kernel (below)

// ---- folded recurrence: policy, buffers, refresh of W_rm and its packed copies ----
static bool fold_wanted(const klstm_engine *e, int T) {
  if (e->use_fold == 0 || !e->use_vector || e->use_bf16 || !e->pk[0] || e->S > get_small_max()) return false;
  // auto: one fold product (2*4C*C*R flop, ~41 us at 800/512) per Update against T-1 saved projection launches and T-1
  // saved d_r launches; every group of 4 streams re-reads the (larger) folded operands, so the gain is gone by 12 streams
  // (measured at 40/800/512, T = 20, fwd+bwd+update: 1 stream 361 -> 296 us, 4: 377 -> 311, 8: 406 -> 396, 12: 472 -> 522)
  return e->use_fold == 1 ? T >= 2 : (T >= 12 && e->S <= 8);
}
static bool use_fused_x(const klstm_engine *e);
// persistent chain: needs the folded operands, up to 8 streams, the x term inside the step or -- wide inputs --
// from the batched product; one launch per direction covers all T steps; auto = on from 8 frames per stream; one compute
// unit per workgroup (an engine on a smaller device or partition keeps to one launch per step)
static bool persist_bwd_wanted(const klstm_engine *e, int T) {
  const Dims d{e->I, e->C, e->R, e->S, T};
  return e->use_persist != 1 && persist_bwd_supported(d, e->popt) && persist_bwd_grid(d) <= e->ncu;
}
static bool persist_wanted(const klstm_engine *e, int T) {
  if (e->replaying || e->cooldown > 0) return false;       // (a minibatch being re-run after a give-up, and the minibatches after it)
  if (e->use_persist == 0 || e->use_fold == 0 || !e->use_vector || e->use_bf16 || !e->pk[0]) return false;
  const Dims d{e->I, e->C, e->R, e->S, T};
  if (!use_fused_x(e) && !persist_x_batched(d)) return false;   // (x inside the step unless the input is wide: then the batched product feeds the launch)
  if (T < 3 || !persist_supported(d, e->popt)) return false;
  if (persist_fwd_grid(d, e->popt) > e->ncu) {                   // (its workgroups could not all be resident at once)
    if (e->use_persist >= 1)
      note("persistent chain not used: it needs %d co-resident workgroups, the device (partition) has %d compute units; "
           "falling back to one launch per step", persist_fwd_grid(d, e->popt), e->ncu);
    return false;
  }
  // auto: from 8 frames on, both directions, 1..8 streams (5..8: two groups of 4 against the same resident rows -- together in
  // the forward launch, one after the other in the backward launch): 272 vs 293 us per minibatch at 8 streams, 166 vs 229 at 4
  // (tools/persist_timing.py, profiles/r03_persist_timing.txt)
  return e->use_persist >= 1 ? true : T >= 8;
}
static klstm_status ensure_persist(klstm_engine *e) {
  if (e->pctrl) return KLSTM_OK;
  const size_t gb = persist_gran_bytes(Dims{e->I, e->C, e->R, e->S, 0});
  for (int i = 0; i < 2; i++) {
    HIPCHK(hipMalloc(&e->gran[i], gb));
    HIPCHK(hipMemsetAsync(e->gran[i], 0, gb, e->stream));
  }
  HIPCHK(hipMalloc(&e->pctrl, 16 * sizeof(unsigned)));   // (+ 8 diagnostic words: which cells' granules never arrived)
  HIPCHK(hipMemsetAsync(e->pctrl, 0, 16 * sizeof(unsigned), e->stream));
  HIPCHK(hipMalloc(&e->tr_ctr, 4 * sizeof(unsigned)));   // "tail_merge": arrivals / expired waits (words of their own: recover() rewrites pctrl)
  HIPCHK(hipMemsetAsync(e->tr_ctr, 0, 4 * sizeof(unsigned), e->stream));
  e->tr_seq = 0;
  if (hipHostMalloc(reinterpret_cast<void **>(&e->pstat_host), 64, hipHostMallocMapped) == hipSuccess) {
    *e->pstat_host = 0u;
    e->pstat_host[1] = 0u;                         // (the done word of finish(): launch count | give-up bit)
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, e->pstat_host, 0) == hipSuccess) e->popt.hstat = static_cast<unsigned *>(dp);
  } else {
    (void)hipGetLastError();
    e->pstat_host = nullptr;                       // (no early notice then: the status words are still read at every sync)
  }
  e->popt.guard = e->pctrl;
  return KLSTM_OK;
}
// ---- a persistent launch gave up (a bounded in-kernel wait expired: its workgroups were not all resident, e.g. another process
// holds part of the chip) ----
// What the device has done by itself: the launch recorded its ordinal and a status word; every persistent launch and every
// gradient / momentum / Update kernel queued behind it found the word and did nothing; the carried state the failed minibatch
// started from is intact (double buffer).  What happens here, at the first call that looks (a poll of the host-mapped word at
// the head of propagate / backpropagate / update / reset, or any synchronising call): the status is cleared, the state index goes
// back to the buffer the first affected forward pass started from, and the calls of the CURRENT minibatch whose launches sit at or
// behind the failure are run again on the launch-per-step chain with the arguments they came with (their buffers are still the
// engine's to read: klstm.h "persist") -- results bit-identical to an engine that never used the persistent chain.  Anything
// older cannot be run again (the caller has its buffers back): that minibatch is dropped -- no Update, no state advance -- and
// counted.  The engine stays on the launch-per-step chain for `cooldown_len` minibatches, then tries the persistent one again.
// Option "persist_verify" = 1 makes every persistent call wait for its launch, so that a give-up is always answered inside the
// call that caused it, before the caller has seen `out` / `in_diff`.
static klstm_status do_propagate(klstm_engine *e, const float *in, int rows, int in_stride, float *out, int out_stride);
static klstm_status do_backpropagate(klstm_engine *e, const float *in, int in_stride, const float *out_diff, int out_diff_stride,
                                     float *in_diff, int in_diff_stride, int rows, float momentum, int flags);
static klstm_status do_update(klstm_engine *e, float learn_rate, float clip_grad);
static klstm_status apply_reset(klstm_engine *e, const std::vector<int> &flags);

static klstm_status recover(klstm_engine *e, const unsigned (&w)[16]) {
  unsigned fseq = 0;                                  // ordinal of the first launch that gave up
  for (unsigned v : {w[3], w[7]}) if (v && (!fseq || v < fseq)) fseq = v;
  unsigned z[16] = {0};
  z[0] = w[0]; z[4] = w[4]; z[8] = w[8]; z[10] = w[10];   // epochs, the launch counter and the count of Updates left out for a peer stay
  HIPCHK(hipMemcpy(e->pctrl, z, sizeof(z), hipMemcpyHostToDevice));
  if (e->pstat_host) e->pstat_host[1] &= 0x7fffffffu;     // (the done word's give-up bit: answered here)
  e->n_giveups++;
  if (e->cooldown_cur < e->cooldown_len || e->clean_run >= (long)e->cooldown_cur) e->cooldown_cur = e->cooldown_len;
  else e->cooldown_cur = e->cooldown_cur >= (1 << 15) ? (1 << 16) : 2 * e->cooldown_cur;
  e->cooldown = e->cooldown_cur;
  e->clean_run = 0;
  // host-side bookkeeping of work the device skipped
  e->grads_pending = false; e->mmt_pending = false; e->verify_later = false; e->tail_pending = false;
  e->planes_fresh = false; e->fold_dirty = true; e->foldx_fresh = false;
  e->wth_fresh = false;                               // (an Update behind the give-up did nothing: the next one that wants the bf16 copies makes them first)
  if (e->pk[0]) e->pk_stale = 15;
  e->bwd_persist = false; e->bwd_xl = false;
  drop_graphs(e);
  // The carried state goes back to the buffer the FIRST affected forward launch started from (intact: every launch behind the failure
  // did nothing).  Decided by the launch ordinal, not by which of the two buffers is current: the host may be any number of
  // minibatches ahead.  Resets the caller issued after that launch was enqueued -- recorded with the later marks, or still pending
  // since the last forward pass -- went to buffers that are abandoned now: they are applied again to the restored one.  (Resets
  // issued BEFORE the first affected launch are in its buffer already.)
  int i0 = -1;
  for (size_t i = 0; i < e->marks.size(); i++)
    if (!fseq || e->marks[i].seq >= fseq) { i0 = (int)i; break; }
  std::vector<int> again;
  if (i0 >= 0) {
    e->sp = e->marks[i0].sp_before;
    auto merge = [&](const std::vector<int> &f) {
      if (f.empty()) return;
      if (again.empty()) again.assign(f.size(), 0);
      for (size_t s = 0; s < f.size() && s < again.size(); s++) if (f[s] == 1) again[s] = 1;
    };
    for (size_t j = (size_t)i0 + 1; j < e->marks.size(); j++) merge(e->marks[j].resets);
    merge(e->resets);
  }
  e->marks.clear();
  const MbRec r = e->rec;
  e->rec = MbRec();
  // A minibatch whose gradient all-reduce is already enqueued cannot be run again by ONE rank (the collective is everybody's): its
  // validity word went out as 1, so every rank -- this one too: the Update kernels look at the reduced word -- leaves that Update
  // out; here it counts as dropped.
  const bool re_fwd = !r.have_ar && r.have_fwd && r.fwd_seq && fseq && r.fwd_seq >= fseq;
  const bool re_bwd = !r.have_ar && r.have_bwd && (re_fwd || (r.bwd_seq && fseq && r.bwd_seq >= fseq));
  const bool re_upd = r.have_upd && re_bwd;
  if (!again.empty()) {
    const klstm_status rs = apply_reset(e, again);
    if (rs != KLSTM_OK) return rs;
  }
  const unsigned affected = fseq && e->pseq >= fseq ? e->pseq - fseq + 1 : 1;
  const unsigned covered = (re_fwd ? 1u : 0u) + (re_bwd && r.bwd_seq ? 1u : 0u);
  if (re_fwd || re_bwd) {
    e->replaying = true;
    klstm_status st = KLSTM_OK;
    if (re_fwd) st = do_propagate(e, r.in, r.rows, r.in_stride, r.out, r.out_stride);
    if (st == KLSTM_OK && re_bwd) st = do_backpropagate(e, r.bin, r.bin_stride, r.od, r.od_stride, r.idf, r.id_stride, r.rows, r.mmt, r.flags);
    if (st == KLSTM_OK && re_upd) st = do_update(e, r.lr, r.clip);
    e->replaying = false;
    if (st != KLSTM_OK) return st;
    e->n_replayed++;
  }
  if (affected > covered) e->n_dropped++;
  note("persistent recurrence chain gave up (forward status %x, backward status %x, launch %u of %u): %s; %d minibatches on the "
       "launch-per-step chain follow", w[2], w[6], fseq, e->pseq,
       affected > covered ? (covered ? "this minibatch was run again on the launch-per-step chain, an earlier one was dropped (no Update, no state advance)"
                                     : "that minibatch was dropped (no Update, no state advance: its buffers were the caller's again)")
                          : "the minibatch was run again on the launch-per-step chain",
       e->cooldown);
  return KLSTM_RECOVERED;
}

// After a host synchronisation of the engine's stream: did a persistent launch give up?  KLSTM_RECOVERED: yes, and it has been
// answered (work may have been enqueued: a caller that has already read results reads them again).
static klstm_status check_persist(klstm_engine *e) {
  if (!e->persist_dirty || !e->pctrl) return KLSTM_OK;
  e->persist_dirty = false;
  if (e->pstat_host) *reinterpret_cast<volatile unsigned *>(e->pstat_host) = 0u;
  unsigned w[16];
  HIPCHK(hipMemcpy(w, e->pctrl, sizeof(w), hipMemcpyDeviceToHost));
  if (w[9] != 0) {
    // the one-shot all-reduce (experimental, klstm_oneshot.hip) did not see every peer in time: the blob was not reduced, the
    // Update kernels of this rank saw the word and did nothing -- but other ranks may have stepped: the run cannot continue
    const unsigned z9 = 0u;
    HIPCHK(hipMemcpy(e->pctrl + 9, &z9, sizeof(z9), hipMemcpyHostToDevice));
    e->grads_pending = false; e->mmt_pending = false;
    { klstm_status ts = flush_tail(e); if (ts != KLSTM_OK) return ts; }
    return fail(KLSTM_ERR_HIP, "one-shot all-reduce timed out (phase %x): a peer did not arrive; the gradient blob was NOT reduced and this "
                "rank's Update was NOT applied -- replicas may have diverged, stop the run (or use klstm_allreduce_grads)", w[9]);
  }
  if (w[2] == 0 && w[6] == 0) { e->marks.clear(); return KLSTM_OK; }     // (everything enqueued so far has run and is good)
  return recover(e, w);
}
// synchronise + look; KLSTM_OK also when a give-up was found and answered
static klstm_status settle(klstm_engine *e) {
  if (!e->persist_dirty) return KLSTM_OK;
  HIPCHK(hipStreamSynchronize(e->stream));
  const klstm_status st = check_persist(e);
  return st == KLSTM_RECOVERED ? KLSTM_OK : st;
}

// Early notice without a synchronisation: the kernels set a host-mapped word when they give up.  Called at the head of
// propagate / backpropagate / update / reset; what it finds belongs to an EARLIER call (launches are asynchronous).
static klstm_status poll_persist(klstm_engine *e) {
  if (!e->pstat_host || !*reinterpret_cast<volatile unsigned *>(e->pstat_host)) return KLSTM_OK;
  e->persist_dirty = true;
  return settle(e);
}

static klstm_status ensure_packs(klstm_engine *e, int want = 15) {       // want: bit i = operand array i is about to be read
  if (!e->pk[0]) { e->pk_stale = 0; return KLSTM_OK; }
  const int todo = e->pk_stale & want;
  if (!todo) return KLSTM_OK;
  const Dims d{e->I, e->C, e->R, e->S, 0};
  { klstm_status ws = ensure_wT32(e); if (ws != KLSTM_OK) return ws; }
  HIPCHK(launch_pack(d, e->params, e->wrT, e->wmT, e->wxT, e->pk, todo, e->use_bf16, e->stream, probe(e, "k_pack")));
  e->pk_stale &= ~todo;
  return KLSTM_OK;
}
// the many-stream bf16 forward launch: policy, buffers, refresh of W_rm
static bool persist_ms_wanted(const klstm_engine *e, int T) {
  if (e->replaying || e->cooldown > 0 || e->use_persist == 0 || !e->use_bf16 || !e->use_vector || !e->pk[0]) return false;
  const Dims d{e->I, e->C, e->R, e->S, T};
  return persist_ms_supported(d) && persist_ms_grid(d) <= e->ncu;
}
static klstm_status ensure_ms(klstm_engine *e) {
  const Dims d{e->I, e->C, e->R, e->S, 0};
  if (!e->wrm_l) {
    HIPCHK(hipMalloc(&e->wrm_l, (size_t)4 * e->C * e->C * sizeof(unsigned short)));
    const size_t gb = persist_ms_gran_bytes(d);
    HIPCHK(hipMalloc(&e->gran_ms, gb));
    HIPCHK(hipMemsetAsync(e->gran_ms, 0, gb, e->stream));
    if (!e->fold_scratch) HIPCHK(hipMalloc(&e->fold_scratch, fold_bf16x3_scratch_bytes(d)));
    e->fold_dirty = true; e->planes_fresh = false;
  }
  const Dims dx{e->I, e->C, e->R, e->S, 32};          // (does the per-XCD BPTT chain apply to this layer?  T is checked per call)
  if (!e->wrmT_l && e->popt.xl_bwd != 0 && persist_xl_supported(dx, e->popt)) {
    HIPCHK(hipMalloc(&e->wrmT_l, (size_t)4 * e->C * e->C * sizeof(unsigned short)));
    const size_t gb = persist_xl_bwd_gran_bytes();
    HIPCHK(hipMalloc(&e->gran_xb, gb));
    HIPCHK(hipMemsetAsync(e->gran_xb, 0, gb, e->stream));
    e->fold_dirty = true;
  }
  if (!e->fold_dirty) return KLSTM_OK;
  // W_rm [4C x C] = W_gifo_r [4C x R] W_r_m [R x C], both operands rounded to bf16 (one plane each: written by the Update, or by
  // a split pass when the parameters changed some other way), fp32 accumulate, stored as bf16 (klstm_fold3.hip)
  HIPCHK(launch_fold_ms(d, e->params + e->o_wr(), e->wmT, e->fold_scratch, e->wrm_l, e->stream, probe(e, "k_split3"), probe(e, "k_fold_ms"),
                        e->planes_fresh, e->wrmT_l));
  e->fold_dirty = false;
  return KLSTM_OK;
}
// need_x: the x chunks of the packed gates operand are read too (launch-per-step folded chain; the persistent kernel
// takes W_gifo_x from the natural matrix)
static klstm_status ensure_fold(klstm_engine *e, bool need_x, bool need_pk2 = true) {
  const Dims d{e->I, e->C, e->R, e->S, 0};
  if (!e->pk_fold[0]) {
    long nf[2];
    pack_sizes_fold(d, nf);
    for (int i = 0; i < 2; i++) {
      HIPCHK(hipMalloc(&e->pk_fold[i], (size_t)nf[i] * 16));
      HIPCHK(hipMemsetAsync(e->pk_fold[i], 0, (size_t)nf[i] * 16, e->stream));   // padding rows / k tails stay zero
    }
    e->fold_dirty = true;
  }
  const bool pack_x = need_x && !e->foldx_fresh;
  if (!e->fold_dirty && !pack_x && !(need_pk2 && !e->pk2_fresh)) return KLSTM_OK;
  // range guard (klstm_math.h): fp16 planes asked for -- one look at this engine's guard per fold product.  A parameter that passed the
  // fp16 range was noticed by the product itself (recomputed in fp32 where it mattered); three bf16 planes (fp32 range, six products
  // instead of three) for the guard's cool-down, then the fp16 planes again.  This engine only.
  {
    const int eff = e->fold_mode == 2 && redo_count(REDO_FOLD) != 0 ? 1 : e->fold_mode;
    if (eff != e->fold_eff) {
      e->fold_eff = eff; e->planes_fresh = false;
      if (!e->graphs.empty()) { HIPCHK(hipStreamSynchronize(e->stream)); drop_graphs(e); }
    }
  }
  if (!e->fold_scratch && fold_bf16x3_supported(d, e->fold_eff)) HIPCHK(hipMalloc(&e->fold_scratch, fold_bf16x3_scratch_bytes(d)));
  const bool f3 = e->fold_scratch && fold_bf16x3_supported(d, e->fold_eff);
  float *pkw[2] = {e->pk_fold[0], need_pk2 ? e->pk_fold[1] : nullptr};
  HIPCHK(launch_fold(d, e->params, e->wmT, pkw, pack_x, e->stream, probe(e, "k_fold"),
                     pack_x ? probe(e, "k_pack_foldx") : LaunchProbe(), f3 ? e->fold_scratch : nullptr,
                     f3 ? probe(e, "k_split3") : LaunchProbe(), f3 && e->planes_fresh, e->fold_eff));
  if (pack_x) e->foldx_fresh = true;
  e->fold_dirty = false;
  e->pk2_fresh = need_pk2;
  return KLSTM_OK;
}

extern "C" {

const char *klstm_last_error(void) { return g_err.c_str(); }
const char *klstm_version(void) { return "klstm 0.4 gfx950 (f32 MFMA, weights-resident chains for 1..8 streams and per-XCD for 9..32 in bf16, 16-bit split-operand products with range guard)"; }

klstm_status klstm_create(int input_dim, int cell_dim, int recur_dim, int num_stream, int device,
                          void *hip_stream, klstm_engine **out) {
  if (!out) return fail(KLSTM_ERR_ARG, "klstm_create: out is null");
  *out = nullptr;
  if (input_dim <= 0 || cell_dim <= 0 || recur_dim <= 0 || num_stream <= 0)
    return fail(KLSTM_ERR_ARG, "klstm_create: dims must be positive (I=%d C=%d R=%d S=%d)", input_dim,
                cell_dim, recur_dim, num_stream);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(KLSTM_ERR_NOGPU, "klstm_create: no HIP device visible (this engine has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(KLSTM_ERR_ARG, "klstm_create: device %d out of range [0,%d)", device, ndev);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(KLSTM_ERR_NOGPU, "klstm_create: device %d is %s, kernels are built for gfx950 only", device, prop.gcnArchName);
  HIPCHK(hipSetDevice(device));
  klstm_engine *e = new klstm_engine();
  e->ncu = prop.multiProcessorCount;
  e->popt.ncu = e->ncu;
  e->fold_mode = e->fold_eff = fold_default_mode();
  e->rg = range_guard_create();
  range_guard_set_note([](const char *m) { note("%s", m); });
  e->I = input_dim; e->C = cell_dim; e->R = recur_dim; e->S = num_stream; e->device = device;
  e->nparams = e->o_wm() + (long)e->R * e->C;
  if (hip_stream) { e->stream = (hipStream_t)hip_stream; e->own_stream = false; }
  else {
    // One process-wide stream per device, shared by every engine created with hip_stream = NULL, and BLOCKING:
    // it synchronises implicitly with the legacy default (NULL) stream, which is where a Kaldi build (and the
    // stateless klstm_* helpers called with hip_stream = NULL) put everything else.  All engines of a stacked net
    // are thereby ordered with each other and with the caller's default-stream work exactly as the reference's
    // single-stream CuMatrix calls were.  (The NULL stream itself cannot be used: hipGraph capture is not allowed on it.)
    static std::mutex mu;
    static std::map<int, hipStream_t> shared;
    std::lock_guard<std::mutex> lock(mu);
    auto it = shared.find(device);
    if (it == shared.end()) {
      hipStream_t s = nullptr;
      hipError_t er = hipStreamCreateWithFlags(&s, hipStreamDefault);
      if (er != hipSuccess) { delete e; return fail(KLSTM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(er)); }
      it = shared.emplace(device, s).first;
    }
    e->stream = it->second;
    e->own_stream = false;
  }
  const size_t pb = (size_t)e->nparams * sizeof(float);
  klstm_status st = KLSTM_OK;
  auto alloc0 = [&](float **p, size_t bytes) {
    if (st != KLSTM_OK) return;
    hipError_t er = hipMalloc(p, bytes);
    if (er == hipSuccess) er = hipMemsetAsync(*p, 0, bytes, e->stream);
    if (er != hipSuccess) st = fail(KLSTM_ERR_HIP, "hipMalloc/Memset(%zu): %s", bytes, hipGetErrorString(er));
  };
  e->ar_len = ((e->nparams + 3) & ~3L) + 4;
  alloc0(&e->params, pb); alloc0(&e->grads, (size_t)e->ar_len * sizeof(float)); alloc0(&e->corr, pb);
  alloc0(&e->wrT, (size_t)4 * e->C * e->R * sizeof(float));
  alloc0(&e->wmT, (size_t)e->R * e->C * sizeof(float));
  alloc0(&e->wxT, (size_t)4 * e->C * e->I * sizeof(float));
  alloc0(reinterpret_cast<float **>(&e->wrTh), (size_t)4 * e->C * e->R * sizeof(unsigned short));
  alloc0(reinterpret_cast<float **>(&e->wxTh), (size_t)4 * e->C * e->I * sizeof(unsigned short));
  for (int b = 0; b < 2; b++) {
    alloc0(&e->prev_c[b], (size_t)e->S * e->C * sizeof(float));
    alloc0(&e->prev_r[b], (size_t)e->S * e->R * sizeof(float));
  }
  {
    const Dims d{e->I, e->C, e->R, e->S, 0};
    if (pack_supported(d)) {
      long n4[4];
      pack_sizes(d, n4);
      for (int i = 0; i < 4; i++) alloc0(&e->pk[i], (size_t)n4[i] * 16);
    }
  }
  if (st == KLSTM_OK && hipMalloc(&e->flags_dev, (size_t)e->S * sizeof(int)) != hipSuccess)
    st = fail(KLSTM_ERR_HIP, "hipMalloc(flags) failed");
  if (st != KLSTM_OK) { klstm_destroy(e); return st; }
  { std::lock_guard<std::mutex> lk(g_engines_mu); g_engines.insert(e); }
  *out = e;
  return KLSTM_OK;
}

void klstm_destroy(klstm_engine *e) {
  if (!e) return;
  { std::lock_guard<std::mutex> lk(g_engines_mu); g_engines.erase(e); }
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  drop_graphs(e);
  for (auto &r : e->probes) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  for (hipEvent_t ev : e->event_pool) (void)hipEventDestroy(ev);
  free_planes(e);
  float *ps[] = {e->params, e->grads_own ? e->grads_own : e->grads, e->corr, e->wrT, e->wmT, e->wxT, e->prev_c[0], e->prev_c[1], e->prev_r[0], e->prev_r[1], e->pk[0], e->pk[1], e->pk[2], e->pk[3],
                 e->pk_fold[0], e->pk_fold[1]};
  for (float *p : ps) if (p) (void)hipFree(p);
  if (e->flags_dev) (void)hipFree(e->flags_dev);
  for (auto *g : e->gran) if (g) (void)hipFree(g);
  if (e->pctrl) (void)hipFree(e->pctrl);
  if (e->tr_ctr) (void)hipFree(e->tr_ctr);
  if (e->fold_scratch) (void)hipFree(e->fold_scratch);
  if (e->wrm_l) (void)hipFree(e->wrm_l);
  if (e->gran_ms) (void)hipFree(e->gran_ms);
  if (e->wrmT_l) (void)hipFree(e->wrmT_l);
  if (e->gran_xb) (void)hipFree(e->gran_xb);
  if (e->tickets) (void)hipFree(e->tickets);
  if (e->wrTh) (void)hipFree(e->wrTh);
  if (e->wxTh) (void)hipFree(e->wxTh);
  range_guard_destroy(e->rg);
  if (e->pstat_host) (void)hipHostFree(e->pstat_host);
  for (float *p : e->stage) if (p) (void)hipFree(p);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int klstm_input_dim(const klstm_engine *e) { return e ? e->I : -1; }
int klstm_cell_dim(const klstm_engine *e) { return e ? e->C : -1; }
int klstm_recur_dim(const klstm_engine *e) { return e ? e->R : -1; }
int klstm_num_stream(const klstm_engine *e) { return e ? e->S : -1; }
long klstm_num_params(const klstm_engine *e) { return e ? e->nparams : -1; }
float *klstm_grad_blob(klstm_engine *e) { return e ? e->grads : nullptr; }
long klstm_grad_blob_len(const klstm_engine *e) { return !e ? -1 : ar_mark(e) ? e->ar_len : e->nparams; }
klstm_status klstm_bind_grad_blob(klstm_engine *e, float *grad_dev) {
  if (!e) return fail(KLSTM_ERR_ARG, "null argument");
  if (grad_dev && (reinterpret_cast<uintptr_t>(grad_dev) & 15)) return fail(KLSTM_ERR_ARG, "klstm_bind_grad_blob: blob must be 16-byte aligned");
  HIPCHK(hipSetDevice(e->device));
  { klstm_status fs = flush_momentum(e); if (fs != KLSTM_OK) return fs; }
  HIPCHK(hipStreamSynchronize(e->stream));
  drop_graphs(e);                                   // captured launches hold the old address
  if (!e->grads_own) e->grads_own = e->grads;
  e->grads = grad_dev ? grad_dev : e->grads_own;
  e->ar_marked = false;
  return KLSTM_OK;
}
float *klstm_param_blob(klstm_engine *e) { return e ? e->params : nullptr; }

static klstm_status blob_h2d(klstm_engine *e, float *dst, const float *src) {
  if (!e || !src) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)e->nparams * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return KLSTM_OK;
}
static klstm_status blob_d2h(klstm_engine *e, float *dst, const float *src) {
  if (!e || !dst) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  for (;;) {
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)e->nparams * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const klstm_status st = check_persist(e);
    if (st != KLSTM_RECOVERED) return st;              // (a give-up was answered just now: what was copied is older than that)
  }
}

klstm_status klstm_set_params_host(klstm_engine *e, const float *flat) {
  klstm_status st = blob_h2d(e, e ? e->params : nullptr, flat);
  if (st != KLSTM_OK) return st;
  return repack(e);
}
klstm_status klstm_set_params_device(klstm_engine *e, const float *flat_dev) {
  if (!e || !flat_dev) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(e->params, flat_dev, (size_t)e->nparams * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
  return repack(e);
}
klstm_status klstm_get_params_host(klstm_engine *e, float *flat) { return blob_d2h(e, flat, e ? e->params : nullptr); }
klstm_status klstm_get_corr_host(klstm_engine *e, float *flat) {
  if (e) { klstm_status st = flush_momentum(e); if (st != KLSTM_OK) return st; }
  return blob_d2h(e, flat, e ? e->corr : nullptr);
}
klstm_status klstm_set_corr_host(klstm_engine *e, const float *flat) {
  if (e) { e->mmt_pending = false; e->grads_pending = false; }    // (whatever was on its way into corr is replaced)
  if (e) { klstm_status ts = flush_tail(e); if (ts != KLSTM_OK) return ts; }   // (in_diff of that minibatch still wants its reduction)
  return blob_h2d(e, e ? e->corr : nullptr, flat);
}
klstm_status klstm_get_grads_host(klstm_engine *e, float *flat) { return blob_d2h(e, flat, e ? e->grads : nullptr); }
/* (klstm_set_params_* leave deferred gradient products pending: they do not read the parameters) */

klstm_status klstm_reset(klstm_engine *e, const int *flags, int n) {
  if (!e || !flags) return fail(KLSTM_ERR_ARG, "klstm_reset: null argument");
  if (n != e->S) return fail(KLSTM_ERR_SHAPE, "klstm_reset: %d flags for %d streams", n, e->S);
  HIPCHK(hipSetDevice(e->device));
  e->rec = MbRec();                                   // a new minibatch begins: the previous one's buffers are the caller's again
  { klstm_status ps = poll_persist(e); if (ps != KLSTM_OK) return ps; }
  const std::vector<int> f(flags, flags + n);
  if (e->resets.empty()) e->resets = f;
  else for (int s = 0; s < n; s++) if (f[s] == 1) e->resets[s] = 1;
  return apply_reset(e, f);
}

klstm_status klstm_get_state_host(klstm_engine *e, float *c, float *r) {
  if (!e || !c || !r) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  { klstm_status ss = settle(e); if (ss != KLSTM_OK) return ss; }       // (which buffer is current depends on it)
  HIPCHK(hipMemcpyAsync(c, e->prev_c[e->sp], (size_t)e->S * e->C * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(r, e->prev_r[e->sp], (size_t)e->S * e->R * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return KLSTM_OK;
}
klstm_status klstm_set_state_host(klstm_engine *e, const float *c, const float *r) {
  if (!e || !c || !r) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  { klstm_status ss = settle(e); if (ss != KLSTM_OK) return ss; }
  HIPCHK(hipMemcpyAsync(e->prev_c[e->sp], c, (size_t)e->S * e->C * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->prev_r[e->sp], r, (size_t)e->S * e->R * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->resets.clear();                                  // (the state is what the caller says, whatever was reset before)
  return KLSTM_OK;
}

}  // extern "C"

// Reset (...streams.h:212-220) on the CURRENT state buffer: zero contiguous runs of flagged streams (the reference issues one
// SetZero per stream, :215-219)
static klstm_status apply_reset(klstm_engine *e, const std::vector<int> &flags) {
  const int n = (int)flags.size();
  int s = 0;
  while (s < n) {
    if (flags[s] != 1) { s++; continue; }
    int s1 = s;
    while (s1 < n && flags[s1] == 1) s1++;
    HIPCHK(hipMemsetAsync(e->prev_c[e->sp] + (size_t)s * e->C, 0, (size_t)(s1 - s) * e->C * sizeof(float), e->stream));
    HIPCHK(hipMemsetAsync(e->prev_r[e->sp] + (size_t)s * e->R, 0, (size_t)(s1 - s) * e->R * sizeof(float), e->stream));
    s = s1;
  }
  return KLSTM_OK;
}

// ------------------------------------------------------------------------------------------------
// launch sequences
// ------------------------------------------------------------------------------------------------
static FwdPtrs fwd_ptrs(klstm_engine *e) {
  FwdPtrs p;
  p.wx = e->params + e->o_wx(); p.wr = e->params + e->o_wr(); p.bias = e->params + e->o_b();
  p.pi = e->params + e->o_pi(); p.pf = e->params + e->o_pf(); p.po = e->params + e->o_po();
  p.wm = e->params + e->o_wm();
  p.gifo = e->gifo; p.cc = e->cc; p.hh = e->hh; p.mm = e->mm; p.rr = e->rr;
  p.prev_c = e->prev_c[e->sp]; p.prev_r = e->prev_r[e->sp];
  p.next_c = e->prev_c[e->sp ^ 1]; p.next_r = e->prev_r[e->sp ^ 1];
  p.pk_gates = e->use_vector ? reinterpret_cast<const float4 *>(e->pk[0]) : nullptr;
  p.pk_proj = e->use_vector ? reinterpret_cast<const float4 *>(e->pk[1]) : nullptr;
  p.pk_fold = reinterpret_cast<const float4 *>(e->pk_fold[0]);
  p.fat = e->use_fat;
  p.bf16 = e->use_bf16;
  p.wr_bf16 = (e->fwd_ms && e->fold_scratch) ? static_cast<const unsigned short *>(e->fold_scratch) : nullptr;   // (fold_bf16x3_planes: plane 0 of a3)
  return p;
}
static BwdPtrs bwd_ptrs(klstm_engine *e) {
  BwdPtrs p;
  p.wrT = e->wrT; p.wmT = e->wmT; p.wxT = e->wxT;
  p.wr_nat = e->params + e->o_wr(); p.wx_nat = e->params + e->o_wx();
  p.pi = e->params + e->o_pi(); p.pf = e->params + e->o_pf(); p.po = e->params + e->o_po();
  p.gifo = e->gifo; p.cc = e->cc; p.hh = e->hh;
  p.dgifo = e->dgifo; p.dc = e->dc; p.dr = e->dr; p.dr_part = e->dr_part; p.dx_part = e->dx_part; p.ks = e->ks;
  p.dgifo_h = e->use_copies ? e->dgifo_h : nullptr;
  p.pk_dr = e->use_vector ? reinterpret_cast<const float4 *>(e->pk[2]) : nullptr;
  p.pk_dm = e->use_vector ? reinterpret_cast<const float4 *>(e->pk[3]) : nullptr;
  p.pk_fold = reinterpret_cast<const float4 *>(e->pk_fold[1]);
  p.pk_fold_gates = reinterpret_cast<const float4 *>(e->pk_fold[0]);
  p.nch_gates = (e->C + 31) / 32 + (e->I + 31) / 32;
  p.fat = e->use_fat;
  p.bf16 = e->use_bf16;
  return p;
}

static bool use_fused_x(const klstm_engine *e) { return e->fuse_x < 0 ? e->S <= 16 : e->fuse_x != 0; }

static klstm_status seq_forward(klstm_engine *e, const float *in, int in_stride, float *out, int out_stride, int T) {
  const Dims d{e->I, e->C, e->R, e->S, T};
  const FwdPtrs p = fwd_ptrs(e);
  hipStream_t st = e->stream;
  const bool fx = use_fused_x(e) && !(e->fwd_persist && persist_x_batched(d)) && !e->fwd_ms;
  if (!fx) {  // x -> g,i,f,o for all frames at once + bias (...streams.h:246, :259)
    const Nt2Job jx{T * d.S, 4 * d.C, d.I, in, in_stride, p.wx, d.I, e->gifo + (size_t)d.S * 4 * d.C, 4 * d.C, p.bias, nullptr, 0};
    hipError_t xerr = hipSuccess;
    if (e->use_bf16 && gemm_bf16_nt_supported(T * d.S, d.I, in, in_stride, p.wx, d.I) && nt2_products(e, &jx, 1, probe(e, "k_gemm_xproj"), &xerr))
      HIPCHK(xerr);                                                                      // (two K tiles in flight: klstm_gemm16.hip)
    else if (e->use_bf16 && gemm_bf16_nt_supported(T * d.S, d.I, in, in_stride, p.wx, d.I))   // bf16 mode: operands rounded like the fused form's
      HIPCHK(launch_gemm_bf16_nt(T * d.S, 4 * d.C, d.I, in, in_stride, p.wx, d.I, e->gifo + (size_t)d.S * 4 * d.C, 4 * d.C, p.bias,
                                 st, probe(e, "k_gemm_xproj")));
    else if (direct_nt_supported(T * d.S, 4 * d.C, d.I, in, in_stride, p.wx, d.I))       // few frames, wide input (klstm_fold.hip)
      HIPCHK(launch_direct_nt(T * d.S, 4 * d.C, d.I, in, in_stride, p.wx, d.I, e->gifo + (size_t)d.S * 4 * d.C, 4 * d.C, p.bias, st,
                              probe(e, "k_gemm_xproj")));
    else
      HIPCHK(launch_gemm(false, true, T * d.S, 4 * d.C, d.I, in, in_stride, p.wx, d.I, 0.f,
                         e->gifo + (size_t)d.S * 4 * d.C, 4 * d.C, p.bias, st, probe(e, "k_gemm_xproj")));
  }
  if (e->fwd_ms) {
    // many streams, bf16 operands: all T steps of the folded recurrence in one launch, r(1..T) = m(1..T) W_r_m^T (:312) -- rr rows,
    // the output rows (:328), the carried r (:331) -- riding along
    HIPCHK(launch_fwd_persist_ms(d, p, e->wrm_l, out, out_stride, e->gran_ms, e->pctrl, e->popt, st,
                                 probe(e, persist_xl_supported(d, e->popt) ? "k_fwd_persist_xl" : "k_fwd_persist_ms")));   // (the kernel's own name)
    return KLSTM_OK;
  }
  if (e->fwd_folded) {
    // step 1 closes over the CARRIED r (set by Reset / the previous minibatch, possibly under older weights): unfolded
    // gates kernel, r(0) mirrored into time block 0; steps 2..T close over m(t-1) through W_rm; r(1..T) in one GEMM
    if (e->fwd_persist) {                         // (all T steps in the one launch, step 1 on the natural matrices)
      HIPCHK(launch_fwd_persist(d, p, in, in_stride, out, out_stride, e->gran[0], e->pctrl, e->popt, st, probe(e, "k_fwd_persist")));
      e->persist_dirty = true;
      if (persist_r_in_kernel(d, e->popt)) return KLSTM_OK;  // (r(1..T), the output rows and the carried r come out of the same launch)
    } else {
      HIPCHK(launch_gates_step(d, p, 1, fx, in, in_stride, st, probe(e, "k_gates_step")));
      for (int t = 2; t <= T; t++)
        HIPCHK(launch_gates_step(d, p, t, fx, in, in_stride, st, probe(e, "k_gates_fold"), true));
    }
    // (behind a persistent launch the product that writes `out` and the carried r(T) -- the OTHER state buffer, the one a failed
    //  minibatch's successor is run again from -- looks at the status words like every other kernel that must not outrun a give-up)
    HIPCHK(launch_rbatch(d, p, out, out_stride, e->ws, st, probe(e, "k_gemm_rbatch"), probe(e, "k_reduce_rbatch"),
                         e->fwd_persist ? e->pctrl : nullptr));
    return KLSTM_OK;
  }
  for (int t = 1; t <= T; t++) {
    HIPCHK(launch_gates_step(d, p, t, fx, in, in_stride, st, probe(e, "k_gates_step")));
    HIPCHK(launch_proj_step(d, p, t, out, out_stride, st, probe(e, "k_proj_step")));
  }
  return KLSTM_OK;
}

// May klstm_backpropagate leave the gradient products to the following klstm_update?
static bool grads_fusable(const klstm_engine *e, int T, int flags, bool bf16_path) {
  (void)T; (void)bf16_path;                           // (round 5: the bf16 tiles carry the fused epilogue too)
  return (flags & KLSTM_BPTT_FUSE_UPDATE) && !(flags & KLSTM_BPTT_DEFER_MOMENTUM) && e->R % 4 == 0 && e->C % 4 == 0 && e->I % 4 == 0 &&
         e->fuse_update_ok;
}
static klstm_status seq_backward(klstm_engine *e, const float *in, int in_stride, const float *out_diff,
                                 int od_stride, float *in_diff, int id_stride, int T, float mmt, int flags) {
  const Dims d{e->I, e->C, e->R, e->S, T};
  const BwdPtrs p = bwd_ptrs(e);
  hipStream_t st = e->stream;
  const bool copies_form = e->bwd_xl && e->use_copies && e->wth_fresh && p.dgifo_h && !e->use_graph;   // d_r + in_diff from the bf16 copies
  // The persistent fp32 launch: d_r / in_diff inside it -- on tail workgroups (the launcher's own conditions: 16-byte rows at the boundary,
  // the workspace, compute units next to the chain's; they read the NATURAL W_gifo_r / W_gifo_x) or on the chain's workgroups
  bool tail_inside = false;
  if (e->fwd_folded && e->bwd_persist && !e->bwd_xl) {
    const bool al = (reinterpret_cast<uintptr_t>(out_diff) & 15) == 0 && od_stride % 4 == 0 &&
                    (!in_diff || ((reinterpret_cast<uintptr_t>(in_diff) & 15) == 0 && id_stride % 4 == 0));
    e->tail_wgs = e->persist_tail != 0 && al && e->ws && e->ws_floats >= persist_bwd_tail_ws_floats(d, in_diff != nullptr)
                      ? persist_bwd_tail_wgs(d, in_diff != nullptr, e->popt) : 0;
    tail_inside = e->persist_tail != 0 && (e->tail_wgs > 0 || persist_tail_in_chain(d, in_diff != nullptr, e->popt));
  }
  const bool natural_only = e->fwd_folded && e->bwd_persist && !e->bwd_xl && (e->tail_wgs > 0 || !tail_inside);   // (nobody in this call reads wrT / wxT)
  if (!copies_form && !natural_only) { klstm_status ws = ensure_wT32(e); if (ws != KLSTM_OK) return ws; }   // (everybody else reads the fp32 wrT / wxT)
  if (e->bwd_xl) {
    // many streams, bf16, one BPTT chain per XCD: P = out_diff W_r_m for all frames, the chain d_m(t) = P(t) + dgifo(t+1) W_rm with the
    // elementwise pass of the own cells, then d_r(1..T) = out_diff + dgifo(2..T+1) W_gifo_r (:391) and in_diff = dgifo W_gifo_x (:457)
    // as batched products -- every product on the bf16 tiles with fp32 accumulation, operands rounded like the step kernels'
    const int M = T * d.S, KS = 8, KL = 4 * d.C / KS;          // (K = 4C in 8 slices of 512: 40 output tiles -> 320 workgroups)
    const Nt2Job jp{M, d.C, d.R, out_diff, od_stride, p.wmT, d.R, e->Pm, d.C, nullptr, nullptr, 0};
    hipError_t perr = hipSuccess;
    if (gemm_bf16_nt_supported(M, d.R, out_diff, od_stride, p.wmT, d.R) && nt2_products(e, &jp, 1, probe(e, "k_gemm_P"), &perr))
      HIPCHK(perr);
    else if (gemm_bf16_nt_supported(M, d.R, out_diff, od_stride, p.wmT, d.R))
      HIPCHK(launch_gemm_bf16_nt(M, d.C, d.R, out_diff, od_stride, p.wmT, d.R, e->Pm, d.C, nullptr, st, probe(e, "k_gemm_P")));
    else                                              // (a caller's view that is not 16-byte aligned: the fp32 tiles take any layout)
      HIPCHK(launch_gemm(false, true, M, d.C, d.R, out_diff, od_stride, p.wmT, d.R, 0.f, e->Pm, d.C, nullptr, st, probe(e, "k_gemm_P")));
    HIPCHK(launch_bwd_persist_xl(d, p, e->wrmT_l, e->Pm, e->gran_xb, e->pctrl + 4, e->popt, st, probe(e, "k_bwd_persist_xl")));
    // d_r and in_diff contract the same dgifo rows (one time block apart) over K = 4C: ONE launch of the pipelined kernel, the K
    // slices' partial tiles added by the last-arriving slice (klstm_gemm16.hip) -- no reduction launches
    const Nt2Job jt[2] = {Nt2Job{M, d.R, 4 * d.C, e->dgifo + (size_t)2 * d.S * 4 * d.C, 4 * d.C, p.wrT, 4 * d.C, e->dr + (size_t)d.S * d.R, d.R, nullptr,
                                 out_diff, od_stride},
                          Nt2Job{M, d.I, 4 * d.C, e->dgifo + (size_t)d.S * 4 * d.C, 4 * d.C, p.wxT, 4 * d.C, in_diff, id_stride, nullptr, nullptr, 0}};
    hipError_t terr = hipSuccess;
    Nt2Job jh[2] = {jt[0], jt[1]};
    if (copies_form) {   // (not under a cached graph: it would bake the choice in) both operands exist as bf16 copies (the chain above wrote dgifo's, the last Update the weights'): LDS-DMA form, same bits
      jh[0].Ah = p.dgifo_h + (size_t)2 * d.S * 4 * d.C; jh[0].Bh = e->wrTh;
      jh[1].Ah = p.dgifo_h + (size_t)d.S * 4 * d.C;     jh[1].Bh = e->wxTh;
    }
    if (nt2_products(e, jh, in_diff ? 2 : 1, probe(e, "k_gemm_dr"), &terr)) HIPCHK(terr);
    else {
      { klstm_status ws = ensure_wT32(e); if (ws != KLSTM_OK) return ws; }
      HIPCHK(launch_gemm_bf16_nt_splitk(M, d.R, 4 * d.C, e->dgifo + (size_t)2 * d.S * 4 * d.C, 4 * d.C, p.wrT, 4 * d.C, 0.f,
                                        e->dr + (size_t)d.S * d.R, d.R, out_diff, od_stride, e->ws, KS, KL, st, probe(e, "k_gemm_dr"),
                                        probe(e, "k_reduce_dr")));
      if (in_diff)
        HIPCHK(launch_gemm_bf16_nt_splitk(M, d.I, 4 * d.C, e->dgifo + (size_t)d.S * 4 * d.C, 4 * d.C, p.wxT, 4 * d.C, 0.f, in_diff, id_stride,
                                          nullptr, 0, e->ws, KS, KL, st, probe(e, "k_gemm_indiff"), probe(e, "k_reduce_indiff")));
    }
    const bool defer = (flags & KLSTM_BPTT_DEFER_MOMENTUM) != 0;
    if (grads_fusable(e, T, flags, e->use_bf16)) return KLSTM_OK;
    HIPCHK(launch_grads(d, e->dgifo, e->dr, in, in_stride, e->rr, e->mm, e->cc, defer ? 0.f : mmt, defer ? e->grads : e->corr, st,
                        probe(e, "k_grads"), e->use_bf16, nullptr, e->pctrl, defer ? ar_mark(e) : nullptr));
    return KLSTM_OK;
  }
  if (e->fwd_folded) {
    const float *wx = e->params + e->o_wx(), *wr = e->params + e->o_wr(), *wm = e->params + e->o_wm();
    const int M = T * d.S;
    // P = out_diff W_r_m for all frames; chain of T folded steps; then d_r(1..T) = out_diff + dgifo(2..T+1) W_gifo_r
    // (:391, feeds the W_r_m gradient :486) and in_diff = dgifo W_gifo_x (:457) as split-K products
    const bool p_inside = e->bwd_persist && persist_p_in_kernel(d, e->popt) && (reinterpret_cast<uintptr_t>(out_diff) & 15) == 0 && od_stride % 4 == 0;
    int kl = 0;
    int ks = gemm_splitk_plan(M, d.C, d.R, &kl);
    if (p_inside) {}                              // (the persistent kernel contracts its own columns of P while its weights load)
    else if (ks > 1) HIPCHK(launch_gemm_splitk(false, false, M, d.C, d.R, out_diff, od_stride, wm, d.C, 0.f, e->Pm, d.C, nullptr, e->ws, ks, kl,
                                          st, nullptr, 0, probe(e, "k_gemm_P"), probe(e, "k_reduce_P")));
    else HIPCHK(launch_gemm(false, false, M, d.C, d.R, out_diff, od_stride, wm, d.C, 0.f, e->Pm, d.C, nullptr, st, probe(e, "k_gemm_P")));
    // "tail_merge": the reduction of the tail workgroups' partial rows rides on the gradient launch that follows (here, or in klstm_update)
    TailReduceJob tj;
    const bool merge = e->bwd_persist && e->tail_merge && e->tail_wgs > 0 && e->tr_ctr && !e->use_graph && d.R % 8 == 0 && d.C % 8 == 0;
    if (e->bwd_persist) {
      HIPCHK(launch_bwd_persist(d, p, e->Pm, out_diff, od_stride, in_diff, id_stride, tail_inside, e->gran[1], e->pctrl + 4, e->popt, st,
                                probe(e, "k_bwd_persist"), e->ws, e->ws_floats, probe(e, "k_tail_reduce"), merge ? &tj : nullptr));
      e->persist_dirty = true;
      if (merge && tj.tws) { e->tr_job = tj; e->tail_pending = true; }
    } else {
      for (int t = T; t >= 1; t--) HIPCHK(launch_dmf_step(d, p, t, e->Pm, st, probe(e, "k_dmf_step")));
    }
    if (!tail_inside)
      HIPCHK(launch_bwd_tail(d, e->dgifo, wr, wx, out_diff, od_stride, e->dr, in_diff, id_stride, e->ws, st,
                             probe(e, "k_gemm_tail"), probe(e, "k_reduce_tail")));
    const bool defer = (flags & KLSTM_BPTT_DEFER_MOMENTUM) != 0;
    if (grads_fusable(e, T, flags, false)) return KLSTM_OK;      // (klstm_update runs them together with the Update)
    HIPCHK(launch_grads(d, e->dgifo, e->dr, in, in_stride, e->rr, e->mm, e->cc, defer ? 0.f : mmt, defer ? e->grads : e->corr, st,
                        probe(e, "k_grads"), false, nullptr, e->pctrl, defer ? ar_mark(e) : nullptr, take_tail_job(e)));
    return KLSTM_OK;
  }
  for (int t = T; t >= 1; t--) {
    if (t < T) HIPCHK(launch_dr_step(d, p, t, in_diff, id_stride, st, probe(e, "k_dr_step")));
    HIPCHK(launch_dm_step(d, p, t, out_diff, od_stride, in_diff, id_stride, st, probe(e, "k_dm_step")));
  }
  // in_diff of frame 1 (:457); frames 2..T were reduced inside the loop.  (Measured: running this as a
  // parallel graph branch next to k_grads makes the whole replay ~120 us SLOWER on ROCm 7.2 -- cross-queue
  // dependencies inside a hipGraph are far more expensive than the 6 us this kernel costs in line.)
  if (in_diff) HIPCHK(launch_dr_step(d, p, 0, in_diff, id_stride, st, probe(e, "k_dr_step0")));
  const bool defer = (flags & KLSTM_BPTT_DEFER_MOMENTUM) != 0;
  float *dst = defer ? e->grads : e->corr;
  const float beta = defer ? 0.f : mmt;
  if (grads_fusable(e, T, flags, e->use_bf16)) return KLSTM_OK;
  HIPCHK(launch_grads(d, e->dgifo, e->dr, in, in_stride, e->rr, e->mm, e->cc, beta, dst, st,
                      probe(e, "k_grads"), e->use_bf16, nullptr, e->pctrl, defer ? ar_mark(e) : nullptr));                  // :468-487
  return KLSTM_OK;
}

template <class F>
static klstm_status run_graphed(klstm_engine *e, const klstm_engine::Key &key, F &&seq, bool short_seq = false) {
  // short_seq: the call is one or two launches (persistent chain): a graph launch costs more than they do (bench A-B at
  // 40/800/512, 4 streams: 196 us per minibatch with a graph per call, 188 with plain launches); option "graph" = 2 forces it
  if (!e->use_graph || e->profile || (short_seq && e->use_graph != 2)) return seq();
  auto it = e->graphs.find(key);
  if (it == e->graphs.end()) {
    // graphs bake the caller's pointers; a trainer that cycles through a pool of minibatch buffers needs one
    // graph per buffer (bench.py: 50 feature chunks x {fwd, bwd}).  Bounded so a pathological caller cannot leak.
    if (e->graphs.size() >= 4096) { HIPCHK(hipStreamSynchronize(e->stream)); drop_graphs(e); }   // launched execs may still run
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    klstm_status st = seq();
    hipError_t er = hipStreamEndCapture(e->stream, &graph);
    if (st != KLSTM_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
    if (er != hipSuccess) return fail(KLSTM_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(er));
    hipGraphExec_t exec = nullptr;
    er = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (er != hipSuccess) return fail(KLSTM_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(er));
    it = e->graphs.emplace(key, exec).first;
  }
  HIPCHK(hipGraphLaunch(it->second, e->stream));
  return KLSTM_OK;
}

// The body of klstm_propagate (arguments checked by the caller): also what recover() runs a minibatch again with.
static klstm_status do_propagate(klstm_engine *e, const float *in, int rows, int in_stride, float *out, int out_stride) {
  const int T = rows / e->S;
  const RangeGuardScope rgs(e->rg);                   // (this engine's fp16-plane products look at this engine's range guard)
  { const klstm_status ks = knobs_seen(e); if (ks != KLSTM_OK) return ks; }
  klstm_status st = ensure_planes(e, T);
  if (st != KLSTM_OK) return st;
  e->fwd_persist = persist_wanted(e, T);
  e->fwd_ms = !e->fwd_persist && persist_ms_wanted(e, T);
  e->bwd_xl = false;
  if (!e->replaying && e->cooldown > 0) e->cooldown--;       // (counted in minibatches that ran on the launch-per-step chain)
  else if (!e->replaying && (e->fwd_persist || e->fwd_ms)) e->clean_run++;
  // launch-per-step kernels are not guarded on the device: nothing of them may be queued behind a persistent launch that
  // nobody has looked at yet (the batched products in front of the many-stream launch only write planes it would have overwritten)
  if (!e->fwd_persist && !e->fwd_ms && (st = settle(e)) != KLSTM_OK) return st;
  e->bwd_persist = e->fwd_persist && persist_bwd_wanted(e, T);
  e->fwd_folded = e->fwd_persist || (!e->fwd_ms && fold_wanted(e, T));
  if ((e->fwd_persist || e->fwd_ms) && (st = ensure_persist(e)) != KLSTM_OK) return st;
  if (e->fwd_ms && (st = ensure_ms(e)) != KLSTM_OK) return st;
  e->bwd_xl = e->fwd_ms && e->wrmT_l && e->popt.xl_bwd != 0 && persist_xl_supported(Dims{e->I, e->C, e->R, e->S, T}, e->popt);
  if ((e->fwd_folded || e->bwd_xl || e->use_bf16) && (st = ensure_ws(e, T)) != KLSTM_OK) return st;
  if (e->use_bf16 && (st = ensure_tickets(e)) != KLSTM_OK) return st;
  // (the persistent backward launch takes its columns of W_rm from the gates-order operand: the second layout is not written then)
  if (e->fwd_folded && (st = ensure_fold(e, !e->fwd_persist, !e->bwd_persist)) != KLSTM_OK) return st;     // outside the graph: only after an Update
  if (!e->fwd_folded && (st = ensure_packs(e, e->fwd_ms ? (e->bwd_xl ? 0 : 12) : 15)) != KLSTM_OK) return st;     // (the many-stream launch reads the natural matrices; BPTT its packed operands)
  if (e->fwd_folded && !e->fwd_persist && (e->pk_stale & 1) && e->pk[0]) {   // step 1 of the launch-per-step folded chain
    const Dims d0{e->I, e->C, e->R, e->S, 0};
    if ((st = ensure_wT32(e)) != KLSTM_OK) return st;
    HIPCHK(launch_pack(d0, e->params, e->wrT, e->wmT, e->wxT, e->pk, 1, e->use_bf16, e->stream, probe(e, "k_pack")));
    e->pk_stale &= ~1;
  }
  klstm_engine::Key key(T, in, in_stride, out, out_stride, nullptr, 0, 0.f, (e->fwd_ms ? -4 : e->fwd_persist ? -3 : e->fwd_folded ? -2 : -1) * 2 - e->sp);
  st = run_graphed(e, key, [&]() { return seq_forward(e, in, in_stride, out, out_stride, T); },
                   e->fwd_persist && persist_r_in_kernel(Dims{e->I, e->C, e->R, e->S, T}, e->popt));
  if (st != KLSTM_OK) return st;
  if (e->fwd_persist || e->fwd_ms) {                  // (counted here, not inside the launch sequence: a graph replay is a launch too)
    e->pseq++;
    e->persist_dirty = true;
    e->marks.push_back(klstm_engine::FwdMark{e->pseq, e->sp, e->resets});   // (the Resets since the forward pass before this one)
  }
  e->sp ^= 1;                                         // c(T), r(T) were written to the other buffer: it is the carried state now
  e->resets.clear();
  e->T_fwd = T;
  e->T_bwd = -1;
  return KLSTM_OK;
}

static klstm_status do_backpropagate(klstm_engine *e, const float *in, int in_stride, const float *out_diff, int out_diff_stride,
                                     float *in_diff, int in_diff_stride, int rows, float momentum, int flags) {
  (void)rows;
  const int T = e->T_fwd;
  const RangeGuardScope rgs(e->rg);
  klstm_status st;
  { const klstm_status ks = knobs_seen(e); if (ks != KLSTM_OK) return ks; }
  e->ar_marked = false;          // (new gradient products: whatever all-reduce the blob went through belongs to an earlier minibatch)
  // (Launch-per-step BPTT kernels behind a persistent forward launch nobody has looked at yet are NOT waited for: if that launch gave
  //  up they compute on invalid planes, but what they write -- derivative planes, in_diff -- is rewritten when the minibatch is run
  //  again, and the gradient / Update kernels behind them are guarded.)
  if (e->fwd_folded) { klstm_status fs = ensure_fold(e, !e->fwd_persist, !e->bwd_persist); if (fs != KLSTM_OK) return fs; }   // no-op unless parameters changed in between
  { klstm_status ts = flush_tail(e); if (ts != KLSTM_OK) return ts; }   // (a reduction still waiting here has lost its gradient launch)
  klstm_engine::Key key(-T, in, in_stride, out_diff, out_diff_stride, in_diff, in_diff_stride, momentum,
                        flags | (e->fwd_folded ? 256 : 0) | (e->bwd_persist ? 512 : 0) | (e->bwd_xl ? 1024 : 0));
  // (one or two launches: the persistent kernel with P and the tail inside, plus at most the gradient products)
  const bool bwd_short = e->bwd_persist && persist_p_in_kernel(Dims{e->I, e->C, e->R, e->S, T}, e->popt) &&
                         persist_tail_in_kernel(Dims{e->I, e->C, e->R, e->S, T}, in_diff != nullptr, e->popt) && e->persist_tail != 0;
  st = run_graphed(e, key, [&]() {
    return seq_backward(e, in, in_stride, out_diff, out_diff_stride, in_diff, in_diff_stride, T, momentum, flags);
  }, bwd_short);
  if (st != KLSTM_OK) return st;
  if (e->bwd_persist || e->bwd_xl) { e->pseq++; e->persist_dirty = true; }
  e->T_bwd = T;
  if (grads_fusable(e, T, flags, e->fwd_folded ? false : e->use_bf16)) {
    e->grads_pending = true;
    e->gp_in = in; e->gp_in_stride = in_stride; e->gp_T = T; e->gp_mmt = momentum; e->gp_bf16 = e->fwd_folded ? false : e->use_bf16;
  }
  return KLSTM_OK;
}

// persist_verify: the call waits for its persistent launch and answers a give-up before it returns
// `reports`: the persistent launch of this call writes the host-mapped done word when it is over (klstm_persist_dev.h finish(): launch
// count + "a status word is up" in one word).  The host then spins on that word -- it hears of the end of the launch one PCIe write
// after the last workgroup left and enqueues what follows at once -- instead of a stream synchronisation (a barrier packet, its
// completion signal, the runtime's wake-up: measured 24 us of idle GPU per wait at 40/800/512 x 4 streams, twice per minibatch).
// Anything unusual (the bit is up, the word does not move for a second, no mapped word) takes the synchronising path.
static klstm_status verify_now(klstm_engine *e, bool reports) {
  if (!e->persist_verify || !e->persist_dirty) return KLSTM_OK;
  if (reports && e->pstat_host && e->verify_spin) {
    const volatile unsigned *done = reinterpret_cast<volatile unsigned *>(e->pstat_host) + 1;
    const unsigned want = e->pseq & 0x7fffffffu;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; spins++) {
      const unsigned v = *done;
      if (v >> 31) break;                                           // a give-up: synchronise, look, answer
      if ((int)((v & 0x7fffffffu) - want) >= 0) {                   // this launch (and everything in front of it) is over and clean
        e->marks.clear();
        e->persist_dirty = false;
        return KLSTM_OK;
      }
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#else
      std::this_thread::yield();
#endif
      if ((spins & 4095) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(1)) break;
    }
  }
  return settle(e);
}
// the wait klstm_backpropagate left to the call that follows it (see there)
static klstm_status verify_deferred(klstm_engine *e) {
  if (!e->verify_later) return KLSTM_OK;
  e->verify_later = false;
  return verify_now(e, e->verify_later_reports);
}

extern "C" {

klstm_status klstm_propagate(klstm_engine *e, const float *in, int rows, int in_stride, float *out, int out_stride) {
  if (!e || ((!in || !out) && rows != 0)) return fail(KLSTM_ERR_ARG, "klstm_propagate: null argument");
  if (rows < 0 || rows % e->S != 0)
    return fail(KLSTM_ERR_SHAPE, "klstm_propagate: rows (%d) %% num_stream (%d) != 0", rows, e->S);
  if (rows == 0) {          // T = 0: the reference's loops simply do not run (:261, :328 with zero rows); state is unchanged
    { klstm_status vs = verify_deferred(e); if (vs != KLSTM_OK) return vs; }   // (a pending wait of the previous minibatch needs its record: as for rows > 0)
    e->rec = MbRec();       // (a new minibatch all the same)
    e->T_fwd = 0;
    e->T_bwd = -1;
    return KLSTM_OK;
  }
  HIPCHK(hipSetDevice(e->device));
  if (in_stride < e->I || out_stride < e->R) return fail(KLSTM_ERR_ARG, "klstm_propagate: stride smaller than row width");
  // (arguments are checked before anything is touched: a rejected call leaves the record of the current minibatch alone)
  { klstm_status vs = verify_deferred(e); if (vs != KLSTM_OK) return vs; }   // (a backpropagate whose promised klstm_update never came: its record is still whole)
  // A new minibatch begins: the previous one's buffers are the caller's again -- in Kaldi `in` IS the previous minibatch's buffer,
  // refilled -- so a give-up of the previous minibatch that the host only hears of here cannot be run again with the arguments it
  // came with: it is dropped and counted (klstm.h "persist"; "persist_verify" = 1 answers inside the call that launched instead).
  e->rec = MbRec();
  { klstm_status ps = poll_persist(e); if (ps != KLSTM_OK) return ps; }
  // (a caller that has not looked for MARKS_MAX persistent minibatches: one host wait, so that the list of unverified launches stays bounded)
  if (e->marks.size() >= klstm_engine::MARKS_MAX) { e->persist_dirty = true; klstm_status ss = settle(e); if (ss != KLSTM_OK) return ss; }
  { klstm_status gs = flush_grads(e); if (gs != KLSTM_OK) return gs; }   // (deferred gradient products read the planes of the last minibatch)
  const int sp0 = e->sp;
  klstm_status st = do_propagate(e, in, rows, in_stride, out, out_stride);
  if (st != KLSTM_OK) return st;
  MbRec &r = e->rec;
  r.have_fwd = true; r.fwd_seq = (e->fwd_persist || e->fwd_ms) ? e->pseq : 0; r.sp_before = sp0;
  r.in = in; r.rows = rows; r.in_stride = in_stride; r.out = out; r.out_stride = out_stride;
  return verify_now(e, e->fwd_persist || (e->fwd_ms && !persist_xl_supported(Dims{e->I, e->C, e->R, e->S, rows / e->S}, e->popt)));
}

klstm_status klstm_backpropagate(klstm_engine *e, const float *in, int in_stride, const float *out_diff,
                                 int out_diff_stride, float *in_diff, int in_diff_stride, int rows,
                                 float momentum, int flags) {
  if (!e || ((!in || !out_diff) && rows != 0)) return fail(KLSTM_ERR_ARG, "klstm_backpropagate: null argument");
  if (e->T_fwd < 0) return fail(KLSTM_ERR_STATE, "klstm_backpropagate: no preceding klstm_propagate");
  if (rows != e->T_fwd * e->S)
    return fail(KLSTM_ERR_SHAPE, "klstm_backpropagate: rows (%d) differ from the preceding propagate (%d)", rows, e->T_fwd * e->S);
  if (rows != 0 && (in_stride < e->I || out_diff_stride < e->R || (in_diff && in_diff_stride < e->I)))
    return fail(KLSTM_ERR_ARG, "klstm_backpropagate: stride smaller than row width");
  HIPCHK(hipSetDevice(e->device));
  { klstm_status ps = poll_persist(e); if (ps != KLSTM_OK) return ps; }
  { klstm_status fs = flush_momentum(e); if (fs != KLSTM_OK) return fs; }   // grads is about to be overwritten
  if (rows == 0) {          // T = 0: every gradient GEMM has K = 0 -> corr = momentum*corr (:468-487), grads = 0
    const bool defer0 = (flags & KLSTM_BPTT_DEFER_MOMENTUM) != 0;
    if (defer0) HIPCHK(hipMemsetAsync(e->grads, 0, (size_t)e->nparams * sizeof(float), e->stream));
    else {
      HIPCHK(hipMemsetAsync(e->grads, 0, (size_t)e->nparams * sizeof(float), e->stream));
      HIPCHK(launch_apply_momentum(e->corr, e->grads, momentum, e->nparams, e->stream, probe(e, "k_apply_momentum")));
    }
    e->T_bwd = 0;
    return KLSTM_OK;
  }
  klstm_status st = do_backpropagate(e, in, in_stride, out_diff, out_diff_stride, in_diff, in_diff_stride, rows, momentum, flags);
  if (st != KLSTM_OK) return st;
  MbRec &r = e->rec;
  r.have_bwd = true; r.bwd_seq = (e->bwd_persist || e->bwd_xl) ? e->pseq : 0;
  r.bin = in; r.bin_stride = in_stride; r.od = out_diff; r.od_stride = out_diff_stride; r.idf = in_diff; r.id_stride = in_diff_stride;
  r.mmt = momentum; r.flags = flags;
  // "klstm_update follows immediately" (KLSTM_BPTT_FUSE_UPDATE, taken: the gradient products wait for it): nobody reads in_diff
  // before that call has returned (Kaldi's Component::Backpropagate runs BackpropagateFnc and Update back to back), so the wait for
  // the BPTT launch moves behind the Update's launches -- they are guarded, a give-up makes them do nothing and is answered there, both
  // calls run again.  The host then enqueues gradient products + Update while the chain runs instead of after it: one idle gap
  // (~5 us at 40/800/512) less per minibatch.  A caller that breaks the promise is looked after by the next call into the engine.
  if (e->persist_verify && e->persist_dirty && e->grads_pending) {
    e->verify_later = true; e->verify_later_reports = e->bwd_persist;
    return KLSTM_OK;
  }
  return verify_now(e, e->bwd_persist);
}

// ---- host-matrix variants: stage through dense device buffers, run the device path, copy back ----
static klstm_status ensure_stage(klstm_engine *e, int rows) {
  if (rows <= e->stage_rows) return KLSTM_OK;
  HIPCHK(hipStreamSynchronize(e->stream));
  for (float *&p : e->stage) { if (p) (void)hipFree(p); p = nullptr; }
  e->stage_rows = 0;
  const int w[4] = {e->I, e->R, e->R, e->I};
  for (int i = 0; i < 4; i++) HIPCHK(hipMalloc(&e->stage[i], (size_t)rows * w[i] * sizeof(float)));
  e->stage_rows = rows;
  return KLSTM_OK;
}
static klstm_status copy_rows(klstm_engine *e, float *dst, int dst_ld, const float *src, int src_ld, int cols, int rows,
                              hipMemcpyKind kind) {
  HIPCHK(hipMemcpy2DAsync(dst, (size_t)dst_ld * sizeof(float), src, (size_t)src_ld * sizeof(float),
                          (size_t)cols * sizeof(float), rows, kind, e->stream));
  return KLSTM_OK;
}

int klstm_pointer_on_device(const klstm_engine *e, const void *p) {
  if (!e || !p) return -1;
  if (hipSetDevice(e->device) != hipSuccess) return -1;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return 0; }   // unknown to the runtime: pageable host
  return (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged || at.devicePointer != nullptr) ? 1 : 0;
}

klstm_status klstm_propagate_host(klstm_engine *e, const float *in, int rows, int in_stride, float *out, int out_stride) {
  if (!e || ((!in || !out) && rows != 0)) return fail(KLSTM_ERR_ARG, "klstm_propagate_host: null argument");
  if (rows <= 0 || rows % e->S != 0) return klstm_propagate(e, in, rows, in_stride, out, out_stride);   // status / no-op as above
  if (in_stride < e->I || out_stride < e->R) return fail(KLSTM_ERR_ARG, "klstm_propagate_host: stride smaller than row width");
  HIPCHK(hipSetDevice(e->device));
  klstm_status st = ensure_stage(e, rows);
  if (st != KLSTM_OK) return st;
  if ((st = copy_rows(e, e->stage[0], e->I, in, in_stride, e->I, rows, hipMemcpyHostToDevice)) != KLSTM_OK) return st;
  if ((st = klstm_propagate(e, e->stage[0], rows, e->I, e->stage[1], e->R)) != KLSTM_OK) return st;
  if ((st = settle(e)) != KLSTM_OK) return st;        // (the staged rows are still in place: a give-up is answered before anything is copied back)
  if ((st = copy_rows(e, out, out_stride, e->stage[1], e->R, e->R, rows, hipMemcpyDeviceToHost)) != KLSTM_OK) return st;
  HIPCHK(hipStreamSynchronize(e->stream));
  return KLSTM_OK;
}

klstm_status klstm_backpropagate_host(klstm_engine *e, const float *in, int in_stride, const float *out_diff,
                                      int out_diff_stride, float *in_diff, int in_diff_stride, int rows,
                                      float momentum, int flags) {
  if (!e || ((!in || !out_diff) && rows != 0)) return fail(KLSTM_ERR_ARG, "klstm_backpropagate_host: null argument");
  flags &= ~KLSTM_BPTT_FUSE_UPDATE;                 // (the staged copy of `in` does not outlive this call's contract)
  if (rows <= 0 || e->T_fwd < 0 || rows != e->T_fwd * e->S)
    return klstm_backpropagate(e, in, in_stride, out_diff, out_diff_stride, in_diff, in_diff_stride, rows, momentum, flags);
  if (in_stride < e->I || out_diff_stride < e->R || (in_diff && in_diff_stride < e->I))
    return fail(KLSTM_ERR_ARG, "klstm_backpropagate_host: stride smaller than row width");
  HIPCHK(hipSetDevice(e->device));
  klstm_status st = ensure_stage(e, rows);
  if (st != KLSTM_OK) return st;
  if ((st = copy_rows(e, e->stage[0], e->I, in, in_stride, e->I, rows, hipMemcpyHostToDevice)) != KLSTM_OK) return st;
  if ((st = copy_rows(e, e->stage[2], e->R, out_diff, out_diff_stride, e->R, rows, hipMemcpyHostToDevice)) != KLSTM_OK) return st;
  st = klstm_backpropagate(e, e->stage[0], e->I, e->stage[2], e->R, in_diff ? e->stage[3] : nullptr, e->I, rows, momentum, flags);
  if (st != KLSTM_OK) return st;
  if ((st = settle(e)) != KLSTM_OK) return st;
  if (in_diff && (st = copy_rows(e, in_diff, in_diff_stride, e->stage[3], e->I, e->I, rows, hipMemcpyDeviceToHost)) != KLSTM_OK) return st;
  HIPCHK(hipStreamSynchronize(e->stream));
  return KLSTM_OK;
}

klstm_status klstm_apply_momentum(klstm_engine *e, float momentum) {
  if (!e) return fail(KLSTM_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->device));
  // Deferred: the following klstm_update performs corr = momentum*corr + grads in its own pass over the blob
  // (one launch instead of two).  Anything else that observes corr or grads flushes it first.
  klstm_status st = flush_momentum(e);
  if (st != KLSTM_OK) return st;
  e->mmt_pending = true;
  e->mmt_value = momentum;
  return KLSTM_OK;
}

}  // extern "C"
// the fp32 wrT / wxT have no reader while this holds: the last backward pass ran the persistent fp32 launch with tail workgroups (or with the
// batched tail behind it: natural matrices too) and nothing suggests the next one will not
static bool natural_readers_only(const klstm_engine *e) {
  return e->skip_wT32 && e->fwd_folded && e->fwd_persist && e->bwd_persist && !e->bwd_xl && !e->use_bf16 && e->tail_wgs > 0 && e->persist_tail == 1 &&
         !e->replaying && !e->use_graph && e->cooldown == 0;
}
static klstm_status do_update(klstm_engine *e, float learn_rate, float clip_grad) {
  const Dims d{e->I, e->C, e->R, e->S, 0};
  const RangeGuardScope rgs(e->rg);
  // theta -= lr * corr, and the transposed copies the BPTT kernels read are refreshed in the same pass
  if (e->grads_pending) {
    // KLSTM_BPTT_FUSE_UPDATE: gradient products, momentum, Update and the transposed copies in one pass
    e->grads_pending = false;
    const Dims dg{e->I, e->C, e->R, e->S, e->gp_T};
    GradsUpdate u{e->params, learn_rate, clip_grad, e->wrT, e->wmT, e->wxT};
    // The persistent fp32 chain with tail workgroups reads the NATURAL W_gifo_r / W_gifo_x: the transposed copies have no reader and are
    // left out (7 MB of the pass's 57 at 40/800/512); ensure_wT32() is there for whoever needs them (in-chain tail, launch-per-step chain, k_pack)
    if (natural_readers_only(e) && !e->gp_bf16) { u.wrT = nullptr; u.wxT = nullptr; e->wT32_stale = true; }
    e->planes_fresh = e->fold_scratch && (e->fwd_ms || (e->fwd_folded && fold_bf16x3_supported(d, e->fold_eff)));
    if (e->planes_fresh) {
      fold_bf16x3_planes(d, e->fold_scratch, &u.a3, &u.a_plane, &u.b3, &u.b_plane);
      u.split_mode = e->fwd_ms ? 3 : e->fold_eff == 2 ? 2 : 1;        // (the many-stream bf16 launch: the operands themselves as bf16)
    }
    const bool want_wth = e->fwd_ms && e->use_copies;                 // (both tile forms of the fused epilogue write them next to wrT / wxT)
    if (want_wth && !e->wth_fresh) { klstm_status ws = refresh_wth(e); if (ws != KLSTM_OK) return ws; }   // (the guarded launch below may do nothing)
    e->wth_fresh = want_wth && e->wth_fresh;
    if (e->wth_fresh) { u.wrTh = e->wrTh; u.wxTh = e->wxTh; }
    // ... and INSTEAD of them while the per-XCD chains run this engine: nothing reads the fp32 wrT / wxT then (ensure_wT32 for whoever does)
    if (e->wth_fresh && e->bwd_xl && e->skip_wT32 && !e->replaying && !e->use_graph && e->gp_bf16) { u.no_wT32 = true; e->wT32_stale = true; }
    HIPCHK(launch_grads(dg, e->dgifo, e->dr, e->gp_in, e->gp_in_stride, e->rr, e->mm, e->cc, e->gp_mmt, e->corr, e->stream,
                        probe(e, "k_grads_update"), e->gp_bf16, &u, e->pctrl, nullptr, take_tail_job(e)));
  } else {
    { klstm_status ts = flush_tail(e); if (ts != KLSTM_OK) return ts; }
    const float *fold_grad = e->mmt_pending ? e->grads : nullptr;
    e->mmt_pending = false;
    // (data-parallel order: gradient -> all-reduce -> this) the planes of the fold operands come out of the same pass
    GradsUpdate u{e->params, learn_rate, clip_grad, e->wrT, e->wmT, e->wxT};
    e->planes_fresh = e->fold_scratch && (e->fwd_ms || (e->fwd_folded && fold_bf16x3_supported(d, e->fold_eff))) &&
                      update_repack_vectorised(d, e->params, e->corr, fold_grad, e->wrT, e->wmT, e->wxT);
    if (e->planes_fresh) {
      fold_bf16x3_planes(d, e->fold_scratch, &u.a3, &u.a_plane, &u.b3, &u.b_plane);
      u.split_mode = e->fwd_ms ? 3 : e->fold_eff == 2 ? 2 : 1;        // (the many-stream bf16 launch: the operands themselves as bf16)
    }
    const bool want_wth = e->planes_fresh && e->fwd_ms && e->use_copies;   // (planes_fresh: the vector kernel runs, and it is handed `u`)
    if (want_wth && !e->wth_fresh) { klstm_status ws = refresh_wth(e); if (ws != KLSTM_OK) return ws; }   // (as above: guard or peer-skip mark)
    e->wth_fresh = want_wth && e->wth_fresh;
    if (e->wth_fresh) { u.wrTh = e->wrTh; u.wxTh = e->wxTh; }
    const bool no32 = (e->wth_fresh && e->bwd_xl && e->skip_wT32 && !e->replaying && !e->use_graph) || natural_readers_only(e);   // (as above)
    if (no32) e->wT32_stale = true;
    HIPCHK(launch_update_repack(d, e->params, e->corr, fold_grad, e->mmt_value, learn_rate, clip_grad, no32 ? nullptr : e->wrT, e->wmT,
                                no32 ? nullptr : e->wxT, e->stream, probe(e, "k_update_repack"), e->pctrl, e->planes_fresh ? &u : nullptr,
                                ar_mark_if_reduced(e), e->pctrl ? e->pctrl + 10 : nullptr));   // (also when the momentum pass ran on its own: a getter in between)
  }
  // (Packing the BPTT operands on a second stream, overlapped with the next forward pass, was measured to cost
  // more in cross-stream event traffic than the ~4 us it hides; everything stays on the one stream.)
  // while the folded chain is in use only the step-1 gates operand (array 0) is read; the others are refreshed on demand
  // (and with the persistent forward kernel none at all: it reads the natural matrices)
  const int mask = e->fwd_persist ? 0 : e->fwd_ms ? (e->bwd_xl ? 0 : 12) : e->fwd_folded ? 1 : 15;   // (one chain per XCD in both directions: no packed operand is read)
  float *foldx = (e->fwd_folded && !e->fwd_persist && !e->use_bf16) ? e->pk_fold[0] : nullptr;
  if (e->pk[0] && (mask || foldx)) { klstm_status ws = ensure_wT32(e); if (ws != KLSTM_OK) return ws; }
  if (e->pk[0] && (mask || foldx)) HIPCHK(launch_pack(d, e->params, e->wrT, e->wmT, e->wxT, e->pk, mask, e->use_bf16, e->stream, probe(e, "k_pack"), foldx));
  e->pk_stale = 15 & ~mask;
  e->fold_dirty = true;
  e->foldx_fresh = foldx != nullptr;
  return KLSTM_OK;
}

extern "C" {
klstm_status klstm_update(klstm_engine *e, float learn_rate, float clip_grad) {
  if (!e) return fail(KLSTM_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->device));
  { klstm_status ps = poll_persist(e); if (ps != KLSTM_OK) return ps; }
  const klstm_status st = do_update(e, learn_rate, clip_grad);
  if (st != KLSTM_OK) return st;
  e->rec.have_upd = true; e->rec.lr = learn_rate; e->rec.clip = clip_grad;
  return verify_deferred(e);                          // ("persist_verify": the BPTT launch of this minibatch, with the Update already enqueued behind it)
}

klstm_status klstm_synchronize(klstm_engine *e) {
  if (!e) return fail(KLSTM_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->device));
  { klstm_status ts = flush_tail(e); if (ts != KLSTM_OK) return ts; }   // ("tail_merge": in_diff / d_r of a BPTT pass whose gradient launch has not come)
  for (;;) {
    HIPCHK(hipStreamSynchronize(e->stream));
    const klstm_status st = check_persist(e);
    if (st != KLSTM_RECOVERED) return st;              // (answered a give-up: wait for what that enqueued)
    e->persist_dirty = false;
  }
}

klstm_status klstm_get_activations_host(klstm_engine *e, int which, float *dst) {
  if (!e || !dst) return fail(KLSTM_ERR_ARG, "null argument");
  const int T = which == 0 ? e->T_fwd : e->T_bwd;
  if (T < 0) return fail(KLSTM_ERR_STATE, "klstm_get_activations_host: nothing has run yet");
  if (T == 0 || !e->gifo) { memset(dst, 0, (size_t)2 * e->S * (7 * e->C + e->R) * sizeof(float)); return KLSTM_OK; }
  HIPCHK(hipSetDevice(e->device));
  { klstm_status ss = klstm_synchronize(e); if (ss != KLSTM_OK) return ss; }
  const int S = e->S, C = e->C, R = e->R, W = 7 * C + R;
  const size_t nrows = (size_t)(T + 2) * S;
  memset(dst, 0, nrows * W * sizeof(float));
  std::vector<float> g4(nrows * 4 * C), c1(nrows * C), r1(nrows * R);
  auto scatter = [&](const std::vector<float> &src, int width, int col0, int tb0, int tb1) {
    for (size_t row = (size_t)tb0 * S; row < (size_t)(tb1 + 1) * S; row++)
      memcpy(dst + row * W + col0, src.data() + row * width, (size_t)width * sizeof(float));
  };
  if (which == 0) {
    HIPCHK(hipMemcpy(g4.data(), e->gifo, g4.size() * sizeof(float), hipMemcpyDeviceToHost));
    scatter(g4, 4 * C, 0, 1, T);
    HIPCHK(hipMemcpy(c1.data(), e->cc, c1.size() * sizeof(float), hipMemcpyDeviceToHost));
    scatter(c1, C, 4 * C, 0, T);
    HIPCHK(hipMemcpy(c1.data(), e->hh, c1.size() * sizeof(float), hipMemcpyDeviceToHost));
    scatter(c1, C, 5 * C, 1, T);
    HIPCHK(hipMemcpy(c1.data(), e->mm, c1.size() * sizeof(float), hipMemcpyDeviceToHost));
    scatter(c1, C, 6 * C, 1, T);
    HIPCHK(hipMemcpy(r1.data(), e->rr, r1.size() * sizeof(float), hipMemcpyDeviceToHost));
    scatter(r1, R, 7 * C, 0, T);
  } else {
    HIPCHK(hipMemcpy(g4.data(), e->dgifo, g4.size() * sizeof(float), hipMemcpyDeviceToHost));
    scatter(g4, 4 * C, 0, 1, T);
    HIPCHK(hipMemcpy(c1.data(), e->dc, c1.size() * sizeof(float), hipMemcpyDeviceToHost));
    scatter(c1, C, 4 * C, 1, T);
    HIPCHK(hipMemcpy(r1.data(), e->dr, r1.size() * sizeof(float), hipMemcpyDeviceToHost));
    scatter(r1, R, 7 * C, 1, T);
  }
  return KLSTM_OK;
}

klstm_status klstm_set_option(klstm_engine *e, const char *key, int value) {
  if (!e || !key) return fail(KLSTM_ERR_ARG, "null argument");
  if (!strcmp(key, "graph")) { e->use_graph = value < 0 ? 0 : value > 2 ? 2 : value; return KLSTM_OK; }
  if (!strcmp(key, "vector")) {          // 0: force the generic kernels (testing)
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    e->use_vector = value != 0;
    return KLSTM_OK;
  }
  if (!strcmp(key, "small_max")) {       // process-wide tuning knob (A-B experiments)
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    set_small_max(value);
    return KLSTM_OK;
  }
  if (!strcmp(key, "fat")) {             // 0: keep the 16-row tile kernels also for NumStream > 16 (testing / A-B)
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    e->use_fat = value != 0;
    return KLSTM_OK;
  }
  if (!strcmp(key, "bf16")) {            // 1: bf16 weight/activation operands with fp32 accumulate in the step kernels
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    if (value != 0 && !e->pk[0]) return fail(KLSTM_ERR_SHAPE, "bf16 mode needs I, C, R multiples of 8");
    // 1 = where it pays: from 9 streams on.  Up to 8 streams the fp32 engine runs the weights-resident chain (one launch per direction);
    // bf16 operand mode has no such chain below 9 streams and would step one launch per frame (measured at 40/800/512, T = 20: 4 streams
    // 291 us per minibatch against 147 in fp32, 8 streams 313 against 191; profiles/r06_scale_probe.txt) -- so a request for bf16 at
    // <= 8 streams is served by the fp32 chain: faster AND at fp32 accuracy (the option allows bf16 rounding, it does not demand it).
    // 2 = bf16 operands at any stream count (tests of the launch-per-step bf16 kernels, A-B runs).
    e->use_bf16 = value == 2 || (value != 0 && e->S > 8);
    { klstm_status ws = ensure_wT32(e); if (ws != KLSTM_OK) return ws; }
    HIPCHK(launch_pack(Dims{e->I, e->C, e->R, e->S, 0}, e->params, e->wrT, e->wmT, e->wxT, e->pk, 15, e->use_bf16, e->stream));
    if (value == 1 && !e->use_bf16)
      note("bf16 operand mode at %d streams: served by the fp32 weights-resident chain (one launch per direction up to 8 streams; the bf16 "
           "step kernels take about 1.9x the time per minibatch at 4 streams, 1.6x at 8); \"bf16\" = 2 forces bf16 operands", e->S);
    return KLSTM_OK;
  }
  if (!strcmp(key, "d2h_small")) {       // process-wide: 0 = klstm_memcpy_d2h always through hipMemcpyAsync (A-B runs)
    g_d2h_small = value != 0;
    return KLSTM_OK;
  }
  if (!strcmp(key, "fat_fine")) {
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    set_fat_fine(value);
    return KLSTM_OK;
  }
  if (!strcmp(key, "small_nt2")) {       // process-wide tuning knob (A-B experiments)
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    set_small_nt2(value);
    return KLSTM_OK;
  }
  if (!strcmp(key, "fuse_update")) {     // 0: KLSTM_BPTT_FUSE_UPDATE is ignored (A-B runs)
    { klstm_status gs = flush_grads(e); if (gs != KLSTM_OK) return gs; }
    e->fuse_update_ok = value != 0;
    return KLSTM_OK;
  }
  if (!strcmp(key, "gemm_copies_plan")) {
    const int nj = value >> 4, ks = value & 15;       // 16 nj + ks; 0 = the planner
    if (value != 0 && (value < 0 || (nj != 1 && nj != 2 && nj != 4) || (ks != 1 && ks != 2 && ks != 4 && ks != 8)))
      return fail(KLSTM_ERR_ARG, "gemm_copies_plan: 16 nj + ks with nj in {1, 2, 4} and ks in {1, 2, 4, 8} (or 0)");
    e->copies_plan = value;
    return KLSTM_OK;
  }
  if (!strcmp(key, "gemm_copies")) {     // 0: d_r + in_diff of the many-stream bf16 mode round their fp32 operands while staging them (no bf16 copies are written or read)
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    e->use_copies = value != 0;
    e->skip_wT32 = value == 1;
    if (!e->use_copies) e->wth_fresh = false;
    { klstm_status ws = ensure_wT32(e); if (ws != KLSTM_OK) return ws; }
    return KLSTM_OK;
  }
  if (!strcmp(key, "gemm_nt2")) {        // 0: the batched bf16 products around the many-stream chains on round 4's kernel + reduction launches
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    e->use_nt2 = value != 0;
    return KLSTM_OK;
  }
  if (!strcmp(key, "fold")) {            // -1 auto, 0 never, 1 whenever NumStream <= small_max
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    e->use_fold = value;
    return KLSTM_OK;
  }
  if (!strncmp(key, "persist", 7) && strcmp(key, "persist_tail")) {
    // "persist": -1 auto, 0 off, 1 forward launch only, 2 both directions whenever the shape allows.  The rest are knobs of
    // THIS engine's persistent launches (kernel arguments and geometry): A-B experiments and tests.
    HIPCHK(hipSetDevice(e->device));
    { klstm_status ss = klstm_synchronize(e); if (ss != KLSTM_OK) return ss; }
    drop_graphs(e);                                 // captured launches bake geometry and kernel arguments
    if (!strcmp(key, "persist")) e->use_persist = value;
    else if (!strcmp(key, "persist_tpw")) e->popt.tpw = value;
    else if (!strcmp(key, "persist_waves")) { e->popt.waves = value; e->popt.bwd_waves = value; }
    else if (!strcmp(key, "persist_bwd_waves")) e->popt.bwd_waves = value;
    else if (!strcmp(key, "persist_bwd_interleave")) e->popt.bwd_interleave = value;
    else if (!strcmp(key, "persist_fwd_interleave")) e->popt.fwd_interleave = value;
    else if (!strcmp(key, "persist_xl")) e->popt.xl = value;
    else if (!strcmp(key, "persist_xl_bwd")) e->popt.xl_bwd = value;
    else if (!strcmp(key, "persist_nap0")) e->popt.nap0 = value;
    else if (!strcmp(key, "persist_nap")) e->popt.nap = value;
    else if (!strcmp(key, "persist_nap0_bwd")) e->popt.nap0_bwd = value;
    else if (!strcmp(key, "persist_spin_us")) e->popt.spin_limit = (long long)value * 100;      // wall clock: 100 MHz
    else if (!strcmp(key, "persist_test_stall_fwd")) e->popt.test_stall_fwd = value;            // test hooks: force the timeout path
    else if (!strcmp(key, "persist_test_stall_bwd")) e->popt.test_stall_bwd = value;
    else if (!strcmp(key, "persist_ncu")) { e->ncu = value; e->popt.ncu = value; }                                      // test hook: pretend the device has this many CUs
    else if (!strcmp(key, "persist_verify")) e->persist_verify = value != 0;
    else if (!strcmp(key, "persist_verify_spin")) e->verify_spin = value != 0;
    else if (!strcmp(key, "persist_cooldown")) { e->cooldown_len = value < 0 ? 0 : value; e->cooldown_cur = 0; if (e->cooldown > e->cooldown_len) e->cooldown = e->cooldown_len; }
    else return fail(KLSTM_ERR_ARG, "klstm_set_option: unknown key '%s'", key);
    return KLSTM_OK;
  }
  if (!strcmp(key, "tail_merge")) {
    { klstm_status ts = flush_tail(e); if (ts != KLSTM_OK) return ts; }
    e->tail_merge = value != 0;
    return KLSTM_OK;
  }
  if (!strcmp(key, "persist_tail")) {
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    e->persist_tail = value;
    e->popt.tail_mode = value;                       // (1: inside, on tail workgroups where they fit; 2: inside, on the chain's workgroups; 0: after the launch)
    return KLSTM_OK;
  }
  // Process-wide knobs (which kernel a product runs on; A-B experiments and tests): the cached graphs of EVERY live engine hold
  // the old kernels, so all of them are dropped (knob_changed_everywhere), not only this engine's.
  if (!strcmp(key, "direct_nt_shape")) {         // value = 10*NI + waves
    set_direct_nt_shape(value / 10, value % 10);
    knob_changed_everywhere();
    HIPCHK(hipSetDevice(e->device));
    return KLSTM_OK;
  }
  if (!strcmp(key, "skinny_f16_pair")) {         // 0: d_r / in_diff of the folded BPTT tail on the tiled split-K kernel
    set_skinny_f16_pair(value);
    knob_changed_everywhere();
    HIPCHK(hipSetDevice(e->device));
    return KLSTM_OK;
  }
  if (!strcmp(key, "skinny_f16")) {              // 0: in_diff of a wide layer on the fp32 MFMA kernel (the two-job launch of the BPTT tail looks at it too)
    set_skinny_f16(value);
    knob_changed_everywhere();
    HIPCHK(hipSetDevice(e->device));
    return KLSTM_OK;
  }
  if (!strcmp(key, "outer_f16")) {               // 0: the wide gradient product on the fp32 tile kernel
    set_outer_f16(value);
    knob_changed_everywhere();
    HIPCHK(hipSetDevice(e->device));
    return KLSTM_OK;
  }
  if (!strcmp(key, "fold_direct")) {             // 0: the fold product on the generic 64x64-tile kernel
    set_fold_direct(value);
    knob_changed_everywhere();
    HIPCHK(hipSetDevice(e->device));
    return KLSTM_OK;
  }
  if (!strcmp(key, "fold_bf16x3")) {             // the fold product of THIS engine: 0 fp32 MFMA (klstm_fold.hip), 1 three bf16 planes, 2 two fp16 planes
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    e->fold_mode = e->fold_eff = value < 0 ? 0 : value > 2 ? 2 : value;   // (the planes at hand are in another format)
    e->planes_fresh = false;
    e->fold_dirty = true;
    return KLSTM_OK;
  }
  if (!strcmp(key, "fp16_products")) {           // 0: nothing of THIS engine -- and no stateless klstm_affine_* call on this device -- runs on fp16 planes;
    HIPCHK(hipSetDevice(e->device));             // 1: the defaults again, the guards' counters and cool-downs cleared.  Other engines keep their own state.
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    range_guard_reset(e->rg, value != 0);
    range_guard_reset(nullptr, value != 0);
    e->fold_mode = e->fold_eff = value ? 2 : 1;  // (the fold product keeps the matrix cores: three bf16 planes have the fp32 range)
    e->planes_fresh = false;
    e->fold_dirty = true;
    return KLSTM_OK;
  }
  if (!strcmp(key, "fuse_x")) {
    HIPCHK(hipStreamSynchronize(e->stream));
    drop_graphs(e);
    e->fuse_x = value;
    return KLSTM_OK;
  }
  if (!strcmp(key, "profile")) {
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (auto &r : e->probes) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    e->probes.clear();
    e->prof.clear();
    e->profile = value != 0;
    return KLSTM_OK;
  }
  return fail(KLSTM_ERR_ARG, "klstm_set_option: unknown key '%s'", key);
}

klstm_status klstm_profile_query(klstm_engine *e, const char *kernel, double *total_us, long *launches) {
  if (!e || !kernel || !total_us || !launches) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  { klstm_status ss = klstm_synchronize(e); if (ss != KLSTM_OK) return ss; }
  // event counters (no option needed): give-ups of the persistent chain and what became of them; range-guard events of the fp16-plane products
  {
    auto ev = [&](int which) { return range_guard_events(e->rg, which) + range_guard_events(nullptr, which); };
    const struct { const char *name; long v; } ctr[] = {
        {"persist_giveups", e->n_giveups}, {"persist_replayed", e->n_replayed}, {"persist_dropped", e->n_dropped},
        {"persist_launches", (long)e->pseq}, {"persist_cooldown", (long)e->cooldown}, {"gemm_copies_launches", e->n_copies},
        {"persist_tail_wgs", e->tail_wgs}, {"tail_merge_launches", (long)e->tr_seq},
        // range-guard events: this engine's own products + the stateless klstm_affine_* calls made on this device (its default guard)
        {"fp16_redo", ev(REDO_FOLD) + ev(REDO_NT) + ev(REDO_OUTER) + ev(REDO_SKINNY)},
        {"fp16_redo_fold", ev(REDO_FOLD)}, {"fp16_redo_nt", ev(REDO_NT)},
        {"fp16_redo_outer", ev(REDO_OUTER)}, {"fp16_redo_skinny", ev(REDO_SKINNY)}, {"fp16_redo_own", range_guard_events(e->rg, REDO_FOLD) +
         range_guard_events(e->rg, REDO_NT) + range_guard_events(e->rg, REDO_OUTER) + range_guard_events(e->rg, REDO_SKINNY)},
        {"fold_mode", (long)e->fold_eff}};
    for (const auto &c : ctr)
      if (!strcmp(kernel, c.name)) { *total_us = 0.0; *launches = c.v; return KLSTM_OK; }
    if (!strcmp(kernel, "tail_merge_timeouts")) {      // W_r_m tiles whose wait for the reduce workgroups of their launch expired (must stay 0)
      unsigned v = 0;
      if (e->tr_ctr) HIPCHK(hipMemcpy(&v, e->tr_ctr + 1, sizeof(v), hipMemcpyDeviceToHost));
      *total_us = 0.0; *launches = (long)v;
      return KLSTM_OK;
    }
    if (!strcmp(kernel, "dp_updates_left_out")) {      // Updates every rank left out because SOME rank's gradient of that minibatch was not real
      unsigned v = 0;
      if (e->pctrl) HIPCHK(hipMemcpy(&v, e->pctrl + 10, sizeof(v), hipMemcpyDeviceToHost));
      *total_us = 0.0; *launches = (long)v;
      return KLSTM_OK;
    }
  }
  for (auto &r : e->probes) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      auto &acc = e->prof[r.name];
      acc.first += (double)ms * 1e3;
      acc.second += 1;
    }
    e->event_pool.push_back(r.start); e->event_pool.push_back(r.stop);
  }
  (void)hipGetLastError();      // a probe whose launch was skipped leaves a sticky error from hipEventElapsedTime
  e->probes.clear();
  auto it = e->prof.find(kernel);
  *total_us = it == e->prof.end() ? 0.0 : it->second.first;
  *launches = it == e->prof.end() ? 0 : it->second.second;
  return KLSTM_OK;
}

klstm_status klstm_time_shift(const float *in, int rows, int cols, int in_stride, float *out, int out_stride, int shift,
                              void *hip_stream) {
  if (!in || !out) return fail(KLSTM_ERR_ARG, "klstm_time_shift: null argument");
  if (rows < 0 || cols < 0 || in_stride < cols || out_stride < cols) return fail(KLSTM_ERR_ARG, "klstm_time_shift: bad shape");
  if (rows == 0 || cols == 0) return KLSTM_OK;
  HIPCHK(launch_time_shift(in, rows, cols, in_stride, out, out_stride, shift, (hipStream_t)hip_stream));
  return KLSTM_OK;
}

// ---- device memory helpers (keep the C++ mirror and other FFI users free of HIP headers) ----
klstm_status klstm_malloc(void **p, size_t bytes) {
  if (!p) return fail(KLSTM_ERR_ARG, "klstm_malloc: null argument");
  HIPCHK(hipMalloc(p, bytes ? bytes : 4));
  return KLSTM_OK;
}
klstm_status klstm_free(void *p) { if (p) HIPCHK(hipFree(p)); return KLSTM_OK; }
klstm_status klstm_memcpy_h2d(void *dst, const void *src, size_t bytes, void *hip_stream) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
  return KLSTM_OK;
}
// Small device-to-host copies -- the three scalars Xent::EvalMasked reads back EVERY minibatch (google/nnet/nnet-loss.cc:110-141: what an
// unmodified bd-nnet-train-lstm-streams does), a CuVector::CopyToVec of a few numbers.  hipMemcpy of pageable memory costs ~45 us of idle
// GPU per call on this stack (staging + completion signal + wake-up: profiles/r05 kaldi_adapter with_d2h_per_minibatch), a third of a
// 150 us minibatch.  Up to D2H_SMALL_BYTES the copy is a ONE-WAVE KERNEL on the caller's stream that writes {sequence tag, word} granules
// into a host-mapped staging buffer (8-byte system-scope stores: tag and word cannot tear, the data is the flag -- no fence, no completion
// signal; the idiom of the persistent chains' granules and of "persist_verify"'s done word), and the host spins on the tags: it has the
// numbers one PCIe write after the kernel ran.  Same semantics as before: stream-ordered behind everything queued on hip_stream, returns
// when dst is filled.  Anything else (larger, unaligned, no mapped memory) takes hipMemcpyAsync + hipStreamSynchronize.
constexpr size_t D2H_SMALL_BYTES = 256;
__global__ __launch_bounds__(64) void k_d2h_small(const unsigned *__restrict__ src, unsigned long long *dst, int nwords, unsigned seq) {
  const int i = threadIdx.x;
  if (i < nwords)
    __hip_atomic_store(dst + i, ((unsigned long long)seq << 32) | src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct D2hStage {
  unsigned long long *host = nullptr, *dev = nullptr;
  int device = -1;
  unsigned seq = 0;
  bool broken = false;
  ~D2hStage() { if (host) (void)hipHostFree(host); }
};
static bool d2h_small(void *dst, const void *src, size_t bytes, hipStream_t st) {
  static thread_local D2hStage sg;    // (the call is synchronous: one staging buffer per calling thread)
  if (!g_d2h_small || sg.broken || bytes == 0 || bytes > D2H_SMALL_BYTES || (bytes & 3) || (reinterpret_cast<uintptr_t>(src) & 3)) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (!sg.host) {
    void *hp = nullptr;
    if (hipHostMalloc(&hp, (D2H_SMALL_BYTES / 4) * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError(); sg.broken = true; return false;
    }
    memset(hp, 0, (D2H_SMALL_BYTES / 4) * sizeof(unsigned long long));
    sg.host = static_cast<unsigned long long *>(hp);
  }
  if (sg.device != dev) {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, sg.host, 0) != hipSuccess) { (void)hipGetLastError(); sg.broken = true; return false; }
    sg.dev = static_cast<unsigned long long *>(dp); sg.device = dev;
  }
  if (++sg.seq == 0) sg.seq = 1;      // (0 = what the buffer holds before its first use)
  const int nw = (int)(bytes / 4);
  hipLaunchKernelGGL(k_d2h_small, dim3(1), dim3(64), 0, st, static_cast<const unsigned *>(src), sg.dev, nw, sg.seq);
  if (hipGetLastError() != hipSuccess) return false;
  const volatile unsigned long long *q = sg.host;
  const auto t0 = std::chrono::steady_clock::now();
  int have = 0;
  bool synced = false;
  for (unsigned spins = 1; have < nw; spins++) {
    while (have < nw && (unsigned)(q[have] >> 32) == sg.seq) have++;
    if (have == nw) break;
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
    if ((spins & 1023) == 0 && !synced && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
      // a long queue in front of the copy: sleep in the runtime instead of burning a core; the granules are there when it returns
      if (hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return false; }
      synced = true;
    } else if (synced && (spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      return false;                   // (never seen: the ordinary copy answers)
    }
  }
  unsigned *out = static_cast<unsigned *>(dst);
  if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0) for (int i = 0; i < nw; i++) out[i] = (unsigned)q[i];
  else for (int i = 0; i < nw; i++) { const unsigned w = (unsigned)q[i]; memcpy(static_cast<char *>(dst) + 4 * i, &w, 4); }
  return true;
}
klstm_status klstm_memcpy_d2h(void *dst, const void *src, size_t bytes, void *hip_stream) {
  if (d2h_small(dst, src, bytes, (hipStream_t)hip_stream)) return KLSTM_OK;
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
  return KLSTM_OK;
}
klstm_status klstm_memset_zero(void *dst, size_t bytes, void *hip_stream) {
  HIPCHK(hipMemsetAsync(dst, 0, bytes, (hipStream_t)hip_stream));
  return KLSTM_OK;
}
klstm_status klstm_stream_synchronize(void *hip_stream) {
  HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
  return KLSTM_OK;
}

// ---- AffineTransform / Softmax / Xent::EvalMasked (stateless) ----
klstm_status klstm_affine_propagate(const float *in, int rows, int in_dim, int in_stride, const float *W,
                                    const float *bias, float *out, int out_dim, int out_stride, void *hip_stream) {
  if (!in || !W || !out) return fail(KLSTM_ERR_ARG, "klstm_affine_propagate: null argument");
  if (direct_nt_supported(rows, out_dim, in_dim, in, in_stride, W, in_dim)) {     // few frames, wide layer: the output tail
    HIPCHK(launch_direct_nt(rows, out_dim, in_dim, in, in_stride, W, in_dim, out, out_stride, bias, (hipStream_t)hip_stream));
    return KLSTM_OK;
  }
  HIPCHK(launch_gemm(false, true, rows, out_dim, in_dim, in, in_stride, W, in_dim, 0.f, out, out_stride, bias,
                     (hipStream_t)hip_stream));
  return KLSTM_OK;
}
// split-K workspace of the stateless ops: one growing buffer per (device, stream), owned by the library
static klstm_status splitk_workspace(hipStream_t st, size_t floats, float **ws) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, std::pair<float *, size_t>> pool;
  int devid = 0;
  HIPCHK(hipGetDevice(&devid));
  std::lock_guard<std::mutex> lk(mu);
  auto &slot = pool[std::make_pair(devid, st)];
  if (slot.second < floats) {
    HIPCHK(hipStreamSynchronize(st));                // earlier launches may still read the old buffer
    if (slot.first) (void)hipFree(slot.first);
    slot.first = nullptr; slot.second = 0;
    HIPCHK(hipMalloc(&slot.first, floats * sizeof(float)));
    slot.second = floats;
  }
  *ws = slot.first;
  return KLSTM_OK;
}
klstm_status klstm_affine_backpropagate(const float *out_diff, int rows, int out_dim, int od_stride, const float *W,
                                        int in_dim, float *in_diff, int id_stride, void *hip_stream) {
  if (!out_diff || !W || !in_diff) return fail(KLSTM_ERR_ARG, "klstm_affine_backpropagate: null argument");
  hipStream_t st = (hipStream_t)hip_stream;
  if (skinny_nn_supported(rows, in_dim, out_dim, out_diff, od_stride, W, in_dim, in_diff, id_stride)) {   // few frames, wide layer (klstm_fold.hip)
    float *ws = nullptr;
    klstm_status s = splitk_workspace(st, skinny_nn_workspace_floats(rows, in_dim, out_dim), &ws);
    if (s != KLSTM_OK) return s;
    HIPCHK(launch_skinny_nn(rows, in_dim, out_dim, out_diff, od_stride, W, in_dim, in_diff, id_stride, ws, st));
    return KLSTM_OK;
  }
  int klen = 0;
  const int ks = gemm_splitk_plan(rows, in_dim, out_dim, &klen);       // contraction over the (long) output axis
  if (ks > 1) {
    float *ws = nullptr;
    klstm_status s = splitk_workspace(st, (size_t)ks * rows * in_dim, &ws);
    if (s != KLSTM_OK) return s;
    HIPCHK(launch_gemm_splitk(false, false, rows, in_dim, out_dim, out_diff, od_stride, W, in_dim, 0.f, in_diff, id_stride,
                              nullptr, ws, ks, klen, st));
    return KLSTM_OK;
  }
  HIPCHK(launch_gemm(false, false, rows, in_dim, out_dim, out_diff, od_stride, W, in_dim, 0.f, in_diff, id_stride, nullptr, st));
  return KLSTM_OK;
}
klstm_status klstm_affine_update(const float *in, int in_stride, const float *out_diff, int od_stride, int rows, int in_dim,
                                 int out_dim, float *W, float *bias, float *W_corr, float *bias_corr, float lr,
                                 float lr_bias, float momentum, void *hip_stream) {
  if (!in || !out_diff || !W || !bias || !W_corr || !bias_corr) return fail(KLSTM_ERR_ARG, "klstm_affine_update: null argument");
  hipStream_t st = (hipStream_t)hip_stream;
  if (outer_f16_supported(out_dim, in_dim, rows, out_diff, od_stride, in, in_stride, W_corr, in_dim, W, bias_corr)) {   // few frames, wide layer (klstm_outer.hip)
    if (reinterpret_cast<uintptr_t>(bias) & 15) {        // (the bias step rides along in 16-byte pieces)
      HIPCHK(launch_outer_f16(out_dim, in_dim, rows, out_diff, od_stride, in, in_stride, momentum, W_corr, in_dim, W, lr, momentum,
                              bias_corr, nullptr, 0.f, st));
      HIPCHK(launch_axpy(bias, bias_corr, -lr_bias, out_dim, st));
    } else {
      HIPCHK(launch_outer_f16(out_dim, in_dim, rows, out_diff, od_stride, in, in_stride, momentum, W_corr, in_dim, W, lr, momentum,
                              bias_corr, bias, lr_bias, st));
    }
    return KLSTM_OK;
  }
  if (in_dim % 4 == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(W_corr)) & 15) == 0) {
    // gradient product, momentum and Update of the weight matrix in one pass (no second trip over the 3 x 34 MB)
    HIPCHK(launch_gemm_tn_update(out_dim, in_dim, rows, out_diff, od_stride, in, in_stride, momentum, W_corr, W, in_dim, lr, st));
  } else {
    HIPCHK(launch_gemm(true, false, out_dim, in_dim, rows, out_diff, od_stride, in, in_stride, momentum, W_corr, in_dim, nullptr, st));
    HIPCHK(launch_axpy(W, W_corr, -lr, (long)out_dim * in_dim, st));
  }
  HIPCHK(launch_col_sum(out_diff, rows, out_dim, od_stride, momentum, bias_corr, st));
  HIPCHK(launch_axpy(bias, bias_corr, -lr_bias, out_dim, st));
  return KLSTM_OK;
}
klstm_status klstm_affine_gradient(const float *in, int in_stride, const float *out_diff, int od_stride, int rows, int in_dim,
                                   int out_dim, float *W_grad, float *bias_grad, void *hip_stream) {
  if (!in || !out_diff || !W_grad || !bias_grad) return fail(KLSTM_ERR_ARG, "klstm_affine_gradient: null argument");
  hipStream_t st = (hipStream_t)hip_stream;
  if (outer_f16_supported(out_dim, in_dim, rows, out_diff, od_stride, in, in_stride, W_grad, in_dim, nullptr, bias_grad)) {
    HIPCHK(launch_outer_f16(out_dim, in_dim, rows, out_diff, od_stride, in, in_stride, 0.f, W_grad, in_dim, nullptr, 0.f, 0.f,
                            bias_grad, nullptr, 0.f, st));
    return KLSTM_OK;
  }
  if (in_dim % 4 == 0 && (reinterpret_cast<uintptr_t>(W_grad) & 15) == 0)
    HIPCHK(launch_gemm_tn_coal(out_dim, in_dim, rows, out_diff, od_stride, in, in_stride, 0.f, W_grad, in_dim, st));
  else
    HIPCHK(launch_gemm(true, false, out_dim, in_dim, rows, out_diff, od_stride, in, in_stride, 0.f, W_grad, in_dim, nullptr, st));
  HIPCHK(launch_col_sum(out_diff, rows, out_dim, od_stride, 0.f, bias_grad, st));
  return KLSTM_OK;
}
klstm_status klstm_sgd_momentum_update(float *param, float *corr, const float *grad, long n, float momentum, float lr,
                                       void *hip_stream) {
  if (!param || !corr || !grad || n < 0) return fail(KLSTM_ERR_ARG, "klstm_sgd_momentum_update: bad argument");
  if (n == 0) return KLSTM_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  HIPCHK(launch_sgd_momentum(param, corr, grad, momentum, lr, n, st));
  return KLSTM_OK;
}
klstm_status klstm_softmax(const float *in, int rows, int cols, int in_stride, float *out, int out_stride, void *hip_stream) {
  if (!in || !out) return fail(KLSTM_ERR_ARG, "klstm_softmax: null argument");
  if (rows > 0) HIPCHK(launch_softmax(in, rows, cols, in_stride, out, out_stride, (hipStream_t)hip_stream));
  return KLSTM_OK;
}
klstm_status klstm_xent_eval_masked(const float *net_out, int rows, int cols, int stride, const int *targets_dev,
                                    const float *mask_dev, float *diff, int diff_stride, float *row_xent_dev,
                                    float *row_correct_dev, void *hip_stream) {
  if (!net_out || !targets_dev || !mask_dev || !diff || !row_xent_dev || !row_correct_dev)
    return fail(KLSTM_ERR_ARG, "klstm_xent_eval_masked: null argument");
  if (rows > 0) HIPCHK(launch_xent(net_out, rows, cols, stride, targets_dev, mask_dev, diff, diff_stride, row_xent_dev,
                                   row_correct_dev, (hipStream_t)hip_stream));
  return KLSTM_OK;
}

// ticket word of the one-pass loss kernel (which workgroup is the last of a launch): one per (device, stream), zero between launches
static klstm_status loss_ticket(hipStream_t st, unsigned **ticket) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, unsigned *> pool;
  int devid = 0;
  HIPCHK(hipGetDevice(&devid));
  std::lock_guard<std::mutex> lk(mu);
  unsigned *&slot = pool[std::make_pair(devid, st)];
  if (!slot) {
    HIPCHK(hipMalloc(&slot, sizeof(unsigned)));
    HIPCHK(hipMemset(slot, 0, sizeof(unsigned)));       // (synchronous with respect to the host: done before any launch uses it)
  }
  *ticket = slot;
  return KLSTM_OK;
}
klstm_status klstm_softmax_xent_masked(const float *net_in, int rows, int cols, int in_stride, float *post, int post_stride,
                                       const int *targets_dev, const float *mask_dev, float *diff, int diff_stride, float *row_xent_dev,
                                       float *row_correct_dev, double *totals_dev, void *hip_stream) {
  if (!net_in || !targets_dev || !mask_dev || !diff || !row_xent_dev || !row_correct_dev)
    return fail(KLSTM_ERR_ARG, "klstm_softmax_xent_masked: null argument");
  if (rows <= 0) return KLSTM_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  unsigned *ticket = nullptr;
  if (totals_dev) {
    const klstm_status s = loss_ticket(st, &ticket);
    if (s != KLSTM_OK) return s;
  }
  const hipError_t e = launch_softmax_xent(net_in, rows, cols, in_stride, post, post_stride, targets_dev, mask_dev, diff, diff_stride,
                                           row_xent_dev, row_correct_dev, totals_dev, ticket, st);
  if (e == hipSuccess) return KLSTM_OK;
  if (e != hipErrorNotSupported) HIPCHK(e);
  // rows the one-pass kernel does not serve: the two kernels, through the caller's posterior matrix
  if (!post) return fail(KLSTM_ERR_ARG, "klstm_softmax_xent_masked: this shape needs the posterior matrix (post) as the buffer between its two kernels");
  HIPCHK(launch_softmax(net_in, rows, cols, in_stride, post, post_stride, st));
  HIPCHK(launch_xent(post, rows, cols, post_stride, targets_dev, mask_dev, diff, diff_stride, row_xent_dev, row_correct_dev, st));
  if (totals_dev) HIPCHK(launch_xent_accumulate(row_xent_dev, row_correct_dev, mask_dev, rows, totals_dev, st));
  return KLSTM_OK;
}

klstm_status klstm_xent_accumulate(const float *row_xent_dev, const float *row_correct_dev, const float *mask_dev, int rows,
                                   double *totals_dev, void *hip_stream) {
  if (!row_xent_dev || !row_correct_dev || !mask_dev || !totals_dev || rows < 0) return fail(KLSTM_ERR_ARG, "klstm_xent_accumulate: bad argument");
  if (rows > 0) HIPCHK(launch_xent_accumulate(row_xent_dev, row_correct_dev, mask_dev, rows, totals_dev, (hipStream_t)hip_stream));
  return KLSTM_OK;
}

klstm_status klstm_xent_eval_masked_post(const float *net_out, int rows, int cols, int stride, const int *post_offsets_dev,
                                         const int *post_pdf_dev, const float *post_weight_dev, const float *mask_dev, float *diff,
                                         int diff_stride, float *row_xent_dev, float *row_entropy_dev, float *row_correct_dev,
                                         void *hip_stream) {
  if (!net_out || !post_offsets_dev || !mask_dev || !diff || !row_xent_dev || !row_entropy_dev || !row_correct_dev)
    return fail(KLSTM_ERR_ARG, "klstm_xent_eval_masked_post: null argument");
  if (rows > 0) HIPCHK(launch_xent_post(net_out, rows, cols, stride, post_offsets_dev, post_pdf_dev, post_weight_dev, mask_dev, diff,
                                        diff_stride, row_xent_dev, row_entropy_dev, row_correct_dev, (hipStream_t)hip_stream));
  return KLSTM_OK;
}

// ---- RCCL (data-parallel training over utterance streams: ONE sum-all-reduce of the gradient blob per minibatch) ----
// libklstm.so has no link-time dependency on RCCL: the entry points are looked up in the process when the first DP call
// arrives -- first among the symbols already loaded (a host framework may have brought its own librccl, e.g. PyTorch's
// bundled one; two copies of RCCL in one process must not be mixed), then by dlopen("librccl.so.1").
}  // extern "C"
namespace {
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  std::string why;
};
RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void *h = nullptr;
    auto sym = [&](const char *name) -> void * {
      void *p = dlsym(RTLD_DEFAULT, name);
      if (!p) {
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) p = dlsym(h, name);
      }
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
    if (!api.ok) api.why = std::string("RCCL entry points not found (") + (dlerror() ? dlerror() : "librccl.so.1 not loadable") + ")";
  });
  return api;
}
klstm_status rccl_fail(const char *what, ncclResult_t r) {
  RcclApi &a = rccl();
  return fail(KLSTM_ERR_HIP, "%s failed: %s", what, a.GetErrorString ? a.GetErrorString(r) : "RCCL error");
}
}  // namespace
extern "C" {

klstm_status klstm_comm_get_unique_id(void *id128) {
  if (!id128) return fail(KLSTM_ERR_ARG, "klstm_comm_get_unique_id: null argument");
  RcclApi &a = rccl();
  if (!a.ok) return fail(KLSTM_ERR_HIP, "%s", a.why.c_str());
  ncclUniqueId id;
  const ncclResult_t r = a.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  static_assert(sizeof(id) == KLSTM_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return KLSTM_OK;
}
klstm_status klstm_comm_init_rank(int device, int nranks, int rank, const void *id128, void **comm) {
  if (!id128 || !comm || nranks <= 0 || rank < 0 || rank >= nranks) return fail(KLSTM_ERR_ARG, "klstm_comm_init_rank: bad argument");
  RcclApi &a = rccl();
  if (!a.ok) return fail(KLSTM_ERR_HIP, "%s", a.why.c_str());
  HIPCHK(hipSetDevice(device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = a.CommInitRank(&c, nranks, id, rank);
  if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
  *comm = c;
  return KLSTM_OK;
}
klstm_status klstm_comm_destroy(void *comm) {
  if (!comm) return KLSTM_OK;
  RcclApi &a = rccl();
  if (!a.ok) return fail(KLSTM_ERR_HIP, "%s", a.why.c_str());
  const ncclResult_t r = a.CommDestroy(static_cast<ncclComm_t>(comm));
  return r == ncclSuccess ? KLSTM_OK : rccl_fail("ncclCommDestroy", r);
}
klstm_status klstm_comm_count(void *comm, int *nranks) {
  if (!comm || !nranks) return fail(KLSTM_ERR_ARG, "klstm_comm_count: null argument");
  RcclApi &a = rccl();
  if (!a.ok || !a.CommCount) return fail(KLSTM_ERR_HIP, "%s", a.ok ? "ncclCommCount not found" : a.why.c_str());
  const ncclResult_t r = a.CommCount(static_cast<ncclComm_t>(comm), nranks);
  return r == ncclSuccess ? KLSTM_OK : rccl_fail("ncclCommCount", r);
}
klstm_status klstm_allreduce_buffer(float *buf_dev, size_t n, void *rccl_comm, void *hip_stream) {
  if (!buf_dev || !rccl_comm) return fail(KLSTM_ERR_ARG, "klstm_allreduce_buffer: null argument");
  if (n == 0) return KLSTM_OK;
  RcclApi &a = rccl();
  if (!a.ok) return fail(KLSTM_ERR_HIP, "%s", a.why.c_str());
  const ncclResult_t r = a.AllReduce(buf_dev, buf_dev, n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(rccl_comm),
                                     static_cast<hipStream_t>(hip_stream));
  return r == ncclSuccess ? KLSTM_OK : rccl_fail("ncclAllReduce", r);
}
klstm_status klstm_allreduce_grads(klstm_engine *e, void *rccl_comm) {
  if (!e) return fail(KLSTM_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->device));
  // A persistent launch of this minibatch that gave up left the gradient products undone (they are guarded): the blob must not
  // reach the other ranks like that -- they would apply the step, this rank would not.
  // Two ways to keep that from happening.  (a) The engine's own blob carries a validity word behind the gradient (written by the
  // gradient kernel: 0 = real, 1 = stopped by the guard) that is summed with it; the Update kernels of EVERY rank leave the step out
  // when the sum is non-zero -- no host wait, replicas identical, the minibatch counts as dropped ("dp_updates_left_out").  What the
  // host has already heard of is still answered first (a look at the host-mapped word, no wait).  (b) With "persist_verify", or a
  // blob bound by the caller (no room for the word): wait and look, as before.
  float *mark = ar_mark(e);
  if (e->persist_verify || !mark) { klstm_status ss = settle(e); if (ss != KLSTM_OK) return ss; }
  else { klstm_status ss = poll_persist(e); if (ss != KLSTM_OK) return ss; }
  { klstm_status fs = flush_momentum(e); if (fs != KLSTM_OK) return fs; }     // a pending corr += grads must see the LOCAL sums
  // option "profile": the exposed time of the collective between two events on the engine's stream ("rccl_allreduce")
  const LaunchProbe pr = probe(e, "rccl_allreduce");
  if (pr.start) HIPCHK(hipEventRecord(pr.start, e->stream));
  const klstm_status st = klstm_allreduce_buffer(e->grads, (size_t)(mark ? e->ar_len : e->nparams), rccl_comm, e->stream);   // in place, on the engine's stream: no event hops
  if (pr.stop) HIPCHK(hipEventRecord(pr.stop, e->stream));
  if (st == KLSTM_OK) { e->ar_marked = mark != nullptr; e->rec.have_ar = true; }
  return st;
}

// The same step through the one-shot exchange over peer-mapped blobs (klstm_oneshot.hip: prepared, off by default, never run
// across devices).  The group must have been created on THIS engine's gradient blob (klstm_grad_blob_ptr / a bound blob).
klstm_status klstm_allreduce_grads_oneshot(klstm_engine *e, klstm_oneshot *group, int timeout_ms) {
  if (!e || !group) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  // (as in klstm_allreduce_grads: the validity word rides along when the group was created over klstm_grad_blob_len() floats)
  float *mark = ar_mark(e);
  const long gn = klstm_oneshot_floats(group);
  if (mark && gn != e->ar_len) mark = nullptr;
  if (!mark && gn != e->nparams) return fail(KLSTM_ERR_ARG, "klstm_allreduce_grads_oneshot: the group spans %ld floats, this engine's blob %ld (or %ld)", gn, e->nparams, e->ar_len);
  if (e->persist_verify || !mark) { klstm_status ss = settle(e); if (ss != KLSTM_OK) return ss; }
  else { klstm_status ss = poll_persist(e); if (ss != KLSTM_OK) return ss; }
  { klstm_status fs = flush_momentum(e); if (fs != KLSTM_OK) return fs; }
  { klstm_status es = ensure_persist(e); if (es != KLSTM_OK) return es; }     // (the control words: a timeout of the exchange gates this engine's Update)
  klstm_oneshot_set_abort_words(group, e->pctrl + 9, e->popt.hstat);
  e->persist_dirty = true;
  const LaunchProbe pr = probe(e, "oneshot_allreduce");
  if (pr.start) HIPCHK(hipEventRecord(pr.start, e->stream));
  const klstm_status st = klstm_oneshot_allreduce(group, e->stream, timeout_ms);
  if (pr.stop) HIPCHK(hipEventRecord(pr.stop, e->stream));
  if (st != KLSTM_OK) return fail(st, "klstm_oneshot_allreduce: %s", klstm_oneshot_last_error());
  e->ar_marked = mark != nullptr; e->rec.have_ar = true;
  return KLSTM_OK;
}

}  // extern "C"

// ---- test support: hold compute units busy (uneven-load tests of the persistent chain) ----
namespace {
__global__ __launch_bounds__(1024) void k_occupy(long long ticks, unsigned *where) {
  extern __shared__ float hog[];                   // 96 KB: nothing else fits next to this workgroup
  hog[threadIdx.x] = (float)threadIdx.x;
  if (where && threadIdx.x == 0) {                   // which XCD / SE / CU this workgroup landed on (HW_REG_XCC_ID = 20, HW_REG_HW_ID = 4)
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    where[2 * blockIdx.x] = xcc; where[2 * blockIdx.x + 1] = hwid;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (hog[threadIdx.x] < 0.f && where) where[0] = 1u;
}
}  // namespace

extern "C" klstm_status klstm_debug_occupy(int device, int workgroups, int microseconds, void *hip_stream, unsigned *where_dev) {
  if (workgroups <= 0 || microseconds <= 0) return fail(KLSTM_ERR_ARG, "klstm_debug_occupy: bad arguments");
  HIPCHK(hipSetDevice(device));
  hipStream_t st = (hipStream_t)hip_stream;
  if (!st) {                                         // a stream of its own per device (streams belong to the device they were made on)
    static std::mutex mu;
    static std::map<int, hipStream_t> own;
    std::lock_guard<std::mutex> lk(mu);
    hipStream_t &o = own[device];
    if (!o) HIPCHK(hipStreamCreateWithFlags(&o, hipStreamNonBlocking));
    st = o;
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL(k_occupy, dim3(workgroups), dim3(1024), 96 * 1024, st, (long long)microseconds * 100, where_dev);
  HIPCHK(hipGetLastError());
  return KLSTM_OK;
}

// Test / probe support for klstm_gemm16.hip (tools/gemm16_probe.py, tests/test_gemm16_gpu.py): C = A B^T (+ bias) (+ add) for one or
// two products that share their K, bf16-rounded operands, fp32 accumulate, on hip_stream (NULL: the default stream).
//   mnk: 3 ints per job; ptrs: A, B, C, bias, add per job (device pointers, bias / add may be null); lds: lda, ldb, ldc, add_ld per job
//   force_nj / force_ks: 0 = the launcher's plan; plan_out (or null): nj, ks, output tiles of the plan that ran
extern "C" klstm_status klstm_debug_gemm_bf16_nt2(int njobs, const int *mnk, const float *const *ptrs, const int *lds, int force_nj, int force_ks,
                                                  void *hip_stream, int *plan_out) {
  return klstm_debug_gemm_bf16_nt2h(njobs, mnk, ptrs, lds, nullptr, force_nj, force_ks, hip_stream, plan_out);
}
extern "C" klstm_status klstm_debug_gemm_bf16_nt2h(int njobs, const int *mnk, const float *const *ptrs, const int *lds,
                                                   const unsigned short *const *copies, int force_nj, int force_ks, void *hip_stream, int *plan_out) {
  if (njobs == 0) { gemm_bf16_nt2_debug_buffer(reinterpret_cast<long long *>(const_cast<int *>(mnk))); return KLSTM_OK; }   // (probe: timing buffer on / off)
  if (njobs < 1 || njobs > 2 || !mnk || !ptrs || !lds) return fail(KLSTM_ERR_ARG, "klstm_debug_gemm_bf16_nt2: bad arguments");
  Nt2Job jobs[2];
  for (int q = 0; q < njobs; q++) {
    jobs[q] = Nt2Job{mnk[3 * q], mnk[3 * q + 1], mnk[3 * q + 2], ptrs[5 * q], lds[4 * q], ptrs[5 * q + 1], lds[4 * q + 1],
                     const_cast<float *>(ptrs[5 * q + 2]), lds[4 * q + 2], ptrs[5 * q + 3], ptrs[5 * q + 4], lds[4 * q + 3]};
    if (!gemm_bf16_nt2_supported(jobs[q])) return fail(KLSTM_ERR_SHAPE, "klstm_debug_gemm_bf16_nt2: job %d not supported by the kernel", q);
    if (copies) {
      jobs[q].Ah = copies[2 * q]; jobs[q].Bh = copies[2 * q + 1];
      if ((jobs[q].Ah || jobs[q].Bh) && !gemm_bf16_nt2_copies_usable(jobs[q]))
        return fail(KLSTM_ERR_SHAPE, "klstm_debug_gemm_bf16_nt2h: job %d: both bf16 copies, 16-byte aligned, leading dimensions %% 8 == 0", q);
    }
  }
  const Nt2Plan pl = gemm_bf16_nt2_plan(jobs, njobs, force_nj, force_ks);
  static std::mutex mu;
  static float *ws = nullptr; static size_t ws_floats = 0; static unsigned *tickets = nullptr;
  std::lock_guard<std::mutex> lk(mu);
  hipStream_t st = (hipStream_t)hip_stream;
  if (pl.ws_floats > ws_floats) {
    HIPCHK(hipDeviceSynchronize());
    if (ws) (void)hipFree(ws);
    ws = nullptr; ws_floats = 0;
    HIPCHK(hipMalloc(&ws, pl.ws_floats * sizeof(float)));
    ws_floats = pl.ws_floats;
  }
  if (!tickets) {
    HIPCHK(hipMalloc(&tickets, NT2_TICKETS * sizeof(unsigned)));
    HIPCHK(hipMemset(tickets, 0, NT2_TICKETS * sizeof(unsigned)));
  }
  if (pl.nt > NT2_TICKETS) return fail(KLSTM_ERR_SHAPE, "klstm_debug_gemm_bf16_nt2: %d output tiles", pl.nt);
  HIPCHK(launch_gemm_bf16_nt2(jobs, njobs, pl, ws, ws_floats, tickets, NT2_TICKETS, st));
  if (plan_out) { plan_out[0] = pl.nj; plan_out[1] = pl.ks; plan_out[2] = pl.nt; }
  return KLSTM_OK;
}
