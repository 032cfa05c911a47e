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
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/klstm.h"
#include "klstm_kernels.h"

using namespace klstm;

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
#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(KLSTM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

struct ProbeRec { std::string name; hipEvent_t start, stop; };

struct klstm_engine {
  int I = 0, C = 0, R = 0, S = 0, device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  long nparams = 0;
  float *params = nullptr, *grads = nullptr, *corr = nullptr;
  float *wrT = nullptr, *wmT = nullptr;
  float *prev_c = nullptr, *prev_r = nullptr;
  int *flags_dev = nullptr;
  // activation planes, (T_alloc+2) time blocks each
  int T_alloc = 0;
  float *gifo = nullptr, *cc = nullptr, *hh = nullptr, *mm = nullptr, *rr = nullptr;
  float *dgifo = nullptr, *dc = nullptr, *dr = nullptr, *dr_part = nullptr;
  int ks = 1;
  int T_fwd = -1;     // T of the last propagate (-1: none yet)
  int T_bwd = -1;
  bool use_graph = true;
  bool profile = false;
  std::vector<ProbeRec> probes;
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
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return pr;
  e->probes.push_back(r);
  pr.start = r.start; pr.stop = r.stop;
  return pr;
}

static void free_planes(klstm_engine *e) {
  float **ps[] = {&e->gifo, &e->cc, &e->hh, &e->mm, &e->rr, &e->dgifo, &e->dc, &e->dr, &e->dr_part};
  for (float **p : ps) { if (*p) (void)hipFree(*p); *p = nullptr; }
}
static void drop_graphs(klstm_engine *e) {
  for (auto &kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
  e->graphs.clear();
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
  HIPCHK(hipMalloc(&e->dc, nb * e->C * sizeof(float)));
  HIPCHK(hipMalloc(&e->dr, nb * e->R * sizeof(float)));
  HIPCHK(hipMalloc(&e->dr_part, (size_t)e->ks * e->S * e->R * sizeof(float)));
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

static klstm_status refresh_transposes(klstm_engine *e) {
  HIPCHK(launch_transpose(e->params + e->o_wr(), 4 * e->C, e->R, e->wrT, e->stream, probe(e, "k_transpose")));
  HIPCHK(launch_transpose(e->params + e->o_wm(), e->R, e->C, e->wmT, e->stream, probe(e, "k_transpose")));
  return KLSTM_OK;
}

extern "C" {

const char *klstm_last_error(void) { return g_err.c_str(); }
const char *klstm_version(void) { return "klstm 0.1 gfx950 (f32 MFMA 16x16x4)"; }

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
  e->I = input_dim; e->C = cell_dim; e->R = recur_dim; e->S = num_stream; e->device = device;
  e->nparams = e->o_wm() + (long)e->R * e->C;
  if (hip_stream) { e->stream = (hipStream_t)hip_stream; e->own_stream = false; }
  else {
    hipError_t er = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (er != hipSuccess) { delete e; return fail(KLSTM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(er)); }
    e->own_stream = true;
  }
  const size_t pb = (size_t)e->nparams * sizeof(float);
  klstm_status st = KLSTM_OK;
  auto alloc0 = [&](float **p, size_t bytes) {
    if (st != KLSTM_OK) return;
    hipError_t er = hipMalloc(p, bytes);
    if (er == hipSuccess) er = hipMemsetAsync(*p, 0, bytes, e->stream);
    if (er != hipSuccess) st = fail(KLSTM_ERR_HIP, "hipMalloc/Memset(%zu): %s", bytes, hipGetErrorString(er));
  };
  alloc0(&e->params, pb); alloc0(&e->grads, pb); alloc0(&e->corr, pb);
  alloc0(&e->wrT, (size_t)4 * e->C * e->R * sizeof(float));
  alloc0(&e->wmT, (size_t)e->R * e->C * sizeof(float));
  alloc0(&e->prev_c, (size_t)e->S * e->C * sizeof(float));
  alloc0(&e->prev_r, (size_t)e->S * e->R * sizeof(float));
  if (st == KLSTM_OK && hipMalloc(&e->flags_dev, (size_t)e->S * sizeof(int)) != hipSuccess)
    st = fail(KLSTM_ERR_HIP, "hipMalloc(flags) failed");
  if (st != KLSTM_OK) { klstm_destroy(e); return st; }
  *out = e;
  return KLSTM_OK;
}

void klstm_destroy(klstm_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  drop_graphs(e);
  for (auto &r : e->probes) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  free_planes(e);
  float *ps[] = {e->params, e->grads, e->corr, e->wrT, e->wmT, e->prev_c, e->prev_r};
  for (float *p : ps) if (p) (void)hipFree(p);
  if (e->flags_dev) (void)hipFree(e->flags_dev);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int klstm_input_dim(const klstm_engine *e) { return e ? e->I : -1; }
int klstm_cell_dim(const klstm_engine *e) { return e ? e->C : -1; }
int klstm_recur_dim(const klstm_engine *e) { return e ? e->R : -1; }
int klstm_num_stream(const klstm_engine *e) { return e ? e->S : -1; }
long klstm_num_params(const klstm_engine *e) { return e ? e->nparams : -1; }
float *klstm_grad_blob(klstm_engine *e) { return e ? e->grads : nullptr; }
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
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)e->nparams * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return KLSTM_OK;
}

klstm_status klstm_set_params_host(klstm_engine *e, const float *flat) {
  klstm_status st = blob_h2d(e, e ? e->params : nullptr, flat);
  if (st != KLSTM_OK) return st;
  return refresh_transposes(e);
}
klstm_status klstm_set_params_device(klstm_engine *e, const float *flat_dev) {
  if (!e || !flat_dev) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(e->params, flat_dev, (size_t)e->nparams * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
  return refresh_transposes(e);
}
klstm_status klstm_get_params_host(klstm_engine *e, float *flat) { return blob_d2h(e, flat, e ? e->params : nullptr); }
klstm_status klstm_get_corr_host(klstm_engine *e, float *flat) { return blob_d2h(e, flat, e ? e->corr : nullptr); }
klstm_status klstm_set_corr_host(klstm_engine *e, const float *flat) { return blob_h2d(e, e ? e->corr : nullptr, flat); }
klstm_status klstm_get_grads_host(klstm_engine *e, float *flat) { return blob_d2h(e, flat, e ? e->grads : nullptr); }

klstm_status klstm_reset(klstm_engine *e, const int *flags, int n) {
  if (!e || !flags) return fail(KLSTM_ERR_ARG, "klstm_reset: null argument");
  if (n != e->S) return fail(KLSTM_ERR_SHAPE, "klstm_reset: %d flags for %d streams", n, e->S);
  HIPCHK(hipSetDevice(e->device));
  // zero contiguous runs of flagged streams (the reference issues one SetZero per stream, :215-219)
  int s = 0;
  while (s < n) {
    if (flags[s] != 1) { s++; continue; }
    int s1 = s;
    while (s1 < n && flags[s1] == 1) s1++;
    HIPCHK(hipMemsetAsync(e->prev_c + (size_t)s * e->C, 0, (size_t)(s1 - s) * e->C * sizeof(float), e->stream));
    HIPCHK(hipMemsetAsync(e->prev_r + (size_t)s * e->R, 0, (size_t)(s1 - s) * e->R * sizeof(float), e->stream));
    s = s1;
  }
  return KLSTM_OK;
}

klstm_status klstm_get_state_host(klstm_engine *e, float *c, float *r) {
  if (!e || !c || !r) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(c, e->prev_c, (size_t)e->S * e->C * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(r, e->prev_r, (size_t)e->S * e->R * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return KLSTM_OK;
}
klstm_status klstm_set_state_host(klstm_engine *e, const float *c, const float *r) {
  if (!e || !c || !r) return fail(KLSTM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(e->prev_c, c, (size_t)e->S * e->C * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->prev_r, r, (size_t)e->S * e->R * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return KLSTM_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// launch sequences
// ------------------------------------------------------------------------------------------------
static FwdPtrs fwd_ptrs(klstm_engine *e) {
  FwdPtrs p;
  p.wx = e->params + e->o_wx(); p.wr = e->params + e->o_wr(); p.bias = e->params + e->o_b();
  p.pi = e->params + e->o_pi(); p.pf = e->params + e->o_pf(); p.po = e->params + e->o_po();
  p.wm = e->params + e->o_wm();
  p.gifo = e->gifo; p.cc = e->cc; p.hh = e->hh; p.mm = e->mm; p.rr = e->rr;
  p.prev_c = e->prev_c; p.prev_r = e->prev_r;
  return p;
}
static BwdPtrs bwd_ptrs(klstm_engine *e) {
  BwdPtrs p;
  p.wrT = e->wrT; p.wmT = e->wmT;
  p.pi = e->params + e->o_pi(); p.pf = e->params + e->o_pf(); p.po = e->params + e->o_po();
  p.gifo = e->gifo; p.cc = e->cc; p.hh = e->hh;
  p.dgifo = e->dgifo; p.dc = e->dc; p.dr = e->dr; p.dr_part = e->dr_part; p.ks = e->ks;
  return p;
}

static klstm_status seq_forward(klstm_engine *e, const float *in, int in_stride, float *out, int out_stride, int T) {
  const Dims d{e->I, e->C, e->R, e->S, T};
  const FwdPtrs p = fwd_ptrs(e);
  hipStream_t st = e->stream;
  HIPCHK(launch_begin(d, p, st, probe(e, "k_begin")));
  // x -> g,i,f,o for all frames at once + bias (...streams.h:246, :259)
  HIPCHK(launch_gemm(false, true, T * d.S, 4 * d.C, d.I, in, in_stride, p.wx, d.I, 0.f,
                     e->gifo + (size_t)d.S * 4 * d.C, 4 * d.C, p.bias, st, probe(e, "k_gemm_xproj")));
  for (int t = 1; t <= T; t++) {
    HIPCHK(launch_gates_step(d, p, t, st, probe(e, "k_gates_step")));
    HIPCHK(launch_proj_step(d, p, t, out, out_stride, st, probe(e, "k_proj_step")));
  }
  HIPCHK(launch_end(d, p, st, probe(e, "k_end")));
  return KLSTM_OK;
}

static klstm_status seq_backward(klstm_engine *e, const float *in, int in_stride, const float *out_diff,
                                 int od_stride, float *in_diff, int id_stride, int T, float mmt, int flags) {
  const Dims d{e->I, e->C, e->R, e->S, T};
  const BwdPtrs p = bwd_ptrs(e);
  hipStream_t st = e->stream;
  const int S = d.S, C = d.C, R = d.R, I = d.I;
  for (int t = T; t >= 1; t--) {
    if (t < T) HIPCHK(launch_dr_step(d, p, t, st, probe(e, "k_dr_step")));
    HIPCHK(launch_dm_step(d, p, t, out_diff, od_stride, st, probe(e, "k_dm_step")));
  }
  const float *dg1 = e->dgifo + (size_t)S * 4 * C;            // DGIFO[1..T]
  if (in_diff)                                                 // :457
    HIPCHK(launch_gemm(false, false, T * S, I, 4 * C, dg1, 4 * C, e->params + e->o_wx(), I, 0.f, in_diff,
                       id_stride, nullptr, st, probe(e, "k_gemm_indiff")));
  const bool defer = (flags & KLSTM_BPTT_DEFER_MOMENTUM) != 0;
  float *dst = defer ? e->grads : e->corr;
  const float beta = defer ? 0.f : mmt;
  HIPCHK(launch_gemm(true, false, 4 * C, I, T * S, dg1, 4 * C, in, in_stride, beta, dst + e->o_wx(), I,
                     nullptr, st, probe(e, "k_gemm_dwx")));                                  // :468
  HIPCHK(launch_gemm(true, false, 4 * C, R, T * S, dg1, 4 * C, e->rr, R, beta, dst + e->o_wr(), R,
                     nullptr, st, probe(e, "k_gemm_dwr")));                                  // :471
  HIPCHK(launch_vec_grads(d, e->dgifo, e->cc, beta, dst + e->o_b(), dst + e->o_pi(), dst + e->o_pf(),
                          dst + e->o_po(), st, probe(e, "k_vec_grads")));                    // :474-484
  HIPCHK(launch_gemm(true, false, R, C, T * S, e->dr + (size_t)S * R, R, e->mm + (size_t)S * C, C, beta,
                     dst + e->o_wm(), C, nullptr, st, probe(e, "k_gemm_dwm")));              // :486
  return KLSTM_OK;
}

template <class F>
static klstm_status run_graphed(klstm_engine *e, const klstm_engine::Key &key, F &&seq) {
  if (!e->use_graph || e->profile) return seq();
  auto it = e->graphs.find(key);
  if (it == e->graphs.end()) {
    if (e->graphs.size() >= 64) drop_graphs(e);
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

extern "C" {

klstm_status klstm_propagate(klstm_engine *e, const float *in, int rows, int in_stride, float *out, int out_stride) {
  if (!e || !in || !out) return fail(KLSTM_ERR_ARG, "klstm_propagate: null argument");
  if (rows <= 0 || rows % e->S != 0)
    return fail(KLSTM_ERR_SHAPE, "klstm_propagate: rows (%d) %% num_stream (%d) != 0", rows, e->S);
  if (in_stride < e->I || out_stride < e->R) return fail(KLSTM_ERR_ARG, "klstm_propagate: stride smaller than row width");
  HIPCHK(hipSetDevice(e->device));
  const int T = rows / e->S;
  klstm_status st = ensure_planes(e, T);
  if (st != KLSTM_OK) return st;
  klstm_engine::Key key(T, in, in_stride, out, out_stride, nullptr, 0, 0.f, -1);
  st = run_graphed(e, key, [&]() { return seq_forward(e, in, in_stride, out, out_stride, T); });
  if (st != KLSTM_OK) return st;
  e->T_fwd = T;
  e->T_bwd = -1;
  return KLSTM_OK;
}

klstm_status klstm_backpropagate(klstm_engine *e, const float *in, int in_stride, const float *out_diff,
                                 int out_diff_stride, float *in_diff, int in_diff_stride, int rows,
                                 float momentum, int flags) {
  if (!e || !in || !out_diff) return fail(KLSTM_ERR_ARG, "klstm_backpropagate: null argument");
  if (e->T_fwd < 0) return fail(KLSTM_ERR_STATE, "klstm_backpropagate: no preceding klstm_propagate");
  if (rows != e->T_fwd * e->S)
    return fail(KLSTM_ERR_SHAPE, "klstm_backpropagate: rows (%d) differ from the preceding propagate (%d)", rows, e->T_fwd * e->S);
  if (in_stride < e->I || out_diff_stride < e->R || (in_diff && in_diff_stride < e->I))
    return fail(KLSTM_ERR_ARG, "klstm_backpropagate: stride smaller than row width");
  HIPCHK(hipSetDevice(e->device));
  const int T = e->T_fwd;
  klstm_engine::Key key(-T, in, in_stride, out_diff, out_diff_stride, in_diff, in_diff_stride, momentum, flags);
  klstm_status st = run_graphed(e, key, [&]() {
    return seq_backward(e, in, in_stride, out_diff, out_diff_stride, in_diff, in_diff_stride, T, momentum, flags);
  });
  if (st != KLSTM_OK) return st;
  e->T_bwd = T;
  return KLSTM_OK;
}

klstm_status klstm_apply_momentum(klstm_engine *e, float momentum) {
  if (!e) return fail(KLSTM_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(launch_apply_momentum(e->corr, e->grads, momentum, e->nparams, e->stream, probe(e, "k_apply_momentum")));
  return KLSTM_OK;
}

klstm_status klstm_update(klstm_engine *e, float learn_rate, float clip_grad) {
  if (!e) return fail(KLSTM_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(launch_update(e->params, e->corr, learn_rate, clip_grad, e->nparams, e->stream, probe(e, "k_update")));
  return refresh_transposes(e);
}

klstm_status klstm_synchronize(klstm_engine *e) {
  if (!e) return fail(KLSTM_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  return KLSTM_OK;
}

klstm_status klstm_get_activations_host(klstm_engine *e, int which, float *dst) {
  if (!e || !dst) return fail(KLSTM_ERR_ARG, "null argument");
  const int T = which == 0 ? e->T_fwd : e->T_bwd;
  if (T < 0) return fail(KLSTM_ERR_STATE, "klstm_get_activations_host: nothing has run yet");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
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
  if (!strcmp(key, "graph")) { e->use_graph = value != 0; return KLSTM_OK; }
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
  HIPCHK(hipStreamSynchronize(e->stream));
  for (auto &r : e->probes) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      auto &acc = e->prof[r.name];
      acc.first += (double)ms * 1e3;
      acc.second += 1;
    }
    (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop);
  }
  e->probes.clear();
  auto it = e->prof.find(kernel);
  *total_us = it == e->prof.end() ? 0.0 : it->second.first;
  *launches = it == e->prof.end() ? 0 : it->second.second;
  return KLSTM_OK;
}

}  // extern "C"
