/*
 * oracle/lstmp_oracle.c -- CPU restatement of the reference's LstmProjectedStreams
 * forward / truncated-BPTT / update path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (kaldi-lstm_amd/, include/)
 * may link, import or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (dophist/kaldi-lstm) ships no tests, golden vectors or
 * known-answer fixtures for this path, and it cannot be compiled in this image: it is a
 * patch overlay over Kaldi nnet1 whose base/ util/ matrix/ cudamatrix/ nnet/ headers are
 * not vendored (writing stand-ins for them is not a reference build).  This restatement
 * is therefore pinned only by construction (op-for-op against the cited lines) and by
 * the independent checks in tests/test_oracle.py (fp64 autograd of the forward
 * equations, finite differences, stream/chunk invariants, and -- for the peephole-free
 * case -- agreement with an independent LSTMP implementation, torch.nn.LSTM with
 * proj_size, on outputs, state and every gradient to 1e-11 in fp64).
 *
 * What it follows (all paths relative to /root/reference):
 *   google/nnet/bd-nnet-lstm-projected-streams.h
 *       :212-220  Reset            :222-332  PropagateFnc
 *       :334-499  BackpropagateFnc :501-512  Update
 *   standard/nnet/nnet-lstm-projected.h:480-505  Update with +-50 gradient clipping
 *   google/matrix/kaldi-matrix.cc
 *       :160-175  AddMatMat (GEMM semantics)    :448-473  AddMatDiagVec
 *       :476-497  AddMatDotMat                  :1869-1886 ApplyFloor/ApplyCeiling
 *       :2546-2559 Sigmoid  :2458-2471 Tanh     :2562-2576 DiffSigmoid (double literal 1.0)
 *       :2579-2593 DiffTanh (double literal 1.0):2598-2609 AddVecToRows
 *   Sigmoid/Tanh scalar forms live in un-vendored kaldi-vector.cc; the overflow-safe
 *   split used by 2014-era Kaldi is restated in k_sigmoid()/k_tanh() below.
 *
 * The op sequence is deliberately UN-FUSED: one loop nest per reference matrix call, one
 * GEMM per reference AddMatMat, the reference's own buffer layout
 * [(T+2)*S rows] x [G|I|F|O|C|H|M (C wide each) | R (R wide)], time-major rows (t*S+s).
 * That is what Kaldi executes when CuDevice is disabled (cu-matrix.cc:816-820,940-944),
 * so timing this file is the "port" CPU baseline.
 *
 * Build: see oracle/Makefile (REAL=float -> liblstmp_oracle_f32.so, double -> _f64.so).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int I, C, R, S;
  int W;  /* buffer width 7C+R */
  /* parameters, reference member order (bd-nnet-lstm-projected-streams.h:589-613) */
  REAL *w_gifo_x;  /* [4C x I]  rows ordered g,i,f,o */
  REAL *w_gifo_r;  /* [4C x R] */
  REAL *bias;      /* [4C] */
  REAL *peep_i, *peep_f, *peep_o; /* [C] each */
  REAL *w_r_m;     /* [R x C] */
  REAL *w_gifo_x_corr, *w_gifo_r_corr, *bias_corr;
  REAL *peep_i_corr, *peep_f_corr, *peep_o_corr, *w_r_m_corr;
  REAL *prev_state;     /* [S x W] */
  REAL *prop;           /* [(T+2)S x W] */
  REAL *bprop;          /* [(T+2)S x W] */
  int T;                /* T of the last propagate */
  int T_alloc;
  int nthreads;
} lstmp_oracle;

static REAL *zalloc(size_t n) { return (REAL *)calloc(n ? n : 1, sizeof(REAL)); }

/* ---- scalar activations ([UPSTREAM-unvendored] kaldi-vector.cc forms) ---- */
static inline REAL k_exp(REAL x) { return sizeof(REAL) == 4 ? (REAL)expf((float)x) : (REAL)exp((double)x); }
static inline REAL k_sigmoid(REAL x) {
  if (x > (REAL)0) { return (REAL)1 / ((REAL)1 + k_exp(-x)); }
  else { REAL ex = k_exp(x); return ex / (ex + (REAL)1); }
}
static inline REAL k_tanh(REAL x) {
  if (x > (REAL)0) { REAL inv = k_exp(-x); return (REAL)-1 + (REAL)2 / ((REAL)1 + inv * inv); }
  else { REAL e = k_exp(x); return (REAL)1 - (REAL)2 / ((REAL)1 + e * e); }
}

/* ---- GEMMs: C = alpha*op(A)*op(B) + beta*C  (kaldi-matrix.cc:160-175) ----
 * Three operand shapes occur on the path.  Each is a plain triple loop written so that
 * the innermost loop is unit-stride (gcc -O3 vectorises it); summation order over k is
 * ascending, as in a reference (non-blocked) sgemm. */

/* Optional BLAS back end for the timed CPU baseline: the reference's CPU path calls cblas_sgemm for every AddMatMat
 * (google/matrix/kaldi-matrix.cc:160-175, cblas_Xgemm).  No CBLAS headers exist in this image; bench.py hands in the
 * address of the ILP64 cblas_sgemm of the OpenBLAS that ships inside numpy (oracle/oracle.py: use_openblas()).  Process-wide;
 * fp32 build only; NULL = the plain loops below (the parity tests always use the plain loops: fixed summation order). */
typedef void (*cblas_sgemm64_fn)(int order, int transA, int transB, long M, long N, long K, float alpha, const float *A,
                                 long lda, const float *B, long ldb, float beta, float *C, long ldc);
static cblas_sgemm64_fn g_sgemm = 0;
void lstmp_oracle_set_sgemm(void *fn) { g_sgemm = (cblas_sgemm64_fn)fn; }
int lstmp_oracle_has_sgemm(void) { return g_sgemm != 0 && sizeof(REAL) == 4; }
enum { CblasRowMajor_ = 101, CblasNoTrans_ = 111, CblasTrans_ = 112 };
static inline int blas_gemm(int ta, int tb, int M, int N, int K, const REAL *A, int lda, const REAL *B, int ldb, REAL beta,
                            REAL *Cm, int ldc) {
  if (!g_sgemm || sizeof(REAL) != 4) return 0;
  g_sgemm(CblasRowMajor_, ta ? CblasTrans_ : CblasNoTrans_, tb ? CblasTrans_ : CblasNoTrans_, M, N, K, 1.0f, (const float *)A, lda,
          (const float *)B, ldb, (float)beta, (float *)Cm, ldc);
  return 1;
}

/* C[MxN] = A[MxK] * B[NxK]^T + beta*C   (kNoTrans,kTrans) -- forward products */
static void gemm_nt(int M, int N, int K, const REAL *A, int lda, const REAL *B, int ldb,
                    REAL beta, REAL *Cm, int ldc, int nthreads) {
  (void)nthreads;
  if (blas_gemm(0, 1, M, N, K, A, lda, B, ldb, beta, Cm, ldc)) return;
#pragma omp parallel for if (nthreads > 1) num_threads(nthreads) schedule(static)
  for (int n = 0; n < N; n++) {
    const REAL *b = B + (size_t)n * ldb;
    for (int m = 0; m < M; m++) {
      const REAL *a = A + (size_t)m * lda;
      REAL acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int k = 0;
      for (; k + 8 <= K; k += 8)
        for (int u = 0; u < 8; u++) acc[u] += a[k + u] * b[k + u];
      REAL s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
      for (; k < K; k++) s += a[k] * b[k];
      REAL *c = Cm + (size_t)m * ldc + n;
      *c = (beta == (REAL)0 ? (REAL)0 : beta * *c) + s;
    }
  }
}

/* C[MxN] = A[MxK] * B[KxN] + beta*C   (kNoTrans,kNoTrans) -- backward data products */
static void gemm_nn(int M, int N, int K, const REAL *A, int lda, const REAL *B, int ldb,
                    REAL beta, REAL *Cm, int ldc, int nthreads) {
  (void)nthreads;
  if (blas_gemm(0, 0, M, N, K, A, lda, B, ldb, beta, Cm, ldc)) return;
  for (int m = 0; m < M; m++) {
    REAL *c = Cm + (size_t)m * ldc;
    if (beta == (REAL)0) { for (int n = 0; n < N; n++) c[n] = 0; }
    else if (beta != (REAL)1) { for (int n = 0; n < N; n++) c[n] *= beta; }
  }
#pragma omp parallel for if (nthreads > 1) num_threads(nthreads) schedule(static)
  for (int nb = 0; nb < N; nb += 256) {
    int ne = nb + 256 < N ? nb + 256 : N;
    for (int m = 0; m < M; m++) {
      REAL *c = Cm + (size_t)m * ldc;
      const REAL *a = A + (size_t)m * lda;
      for (int k = 0; k < K; k++) {
        const REAL av = a[k];
        const REAL *b = B + (size_t)k * ldb;
        for (int n = nb; n < ne; n++) c[n] += av * b[n];
      }
    }
  }
}

/* C[MxN] = A[KxM]^T * B[KxN] + beta*C   (kTrans,kNoTrans) -- weight gradients */
static void gemm_tn(int M, int N, int K, const REAL *A, int lda, const REAL *B, int ldb,
                    REAL beta, REAL *Cm, int ldc, int nthreads) {
  (void)nthreads;
  if (blas_gemm(1, 0, M, N, K, A, lda, B, ldb, beta, Cm, ldc)) return;
#pragma omp parallel for if (nthreads > 1) num_threads(nthreads) schedule(static)
  for (int m = 0; m < M; m++) {
    REAL *c = Cm + (size_t)m * ldc;
    if (beta == (REAL)0) { for (int n = 0; n < N; n++) c[n] = 0; }
    else if (beta != (REAL)1) { for (int n = 0; n < N; n++) c[n] *= beta; }
    for (int k = 0; k < K; k++) {
      const REAL av = A[(size_t)k * lda + m];
      const REAL *b = B + (size_t)k * ldb;
      for (int n = 0; n < N; n++) c[n] += av * b[n];
    }
  }
}

/* ---- elementwise matrix ops on [rows x cols] views with a row stride ---- */
/* kaldi-matrix.cc:448-473: this = beta*this + alpha * M * diag(v); only beta==1 is used */
static void add_mat_diag_vec(int rows, int cols, REAL *d, int ldd, const REAL *M, int ldm, const REAL *v) {
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) d[(size_t)i * ldd + j] += (REAL)1 * v[j] * M[(size_t)i * ldm + j];
}
/* kaldi-matrix.cc:476-497: this = beta*this + alpha*A.*B  (beta*data evaluated even for beta=0) */
static void add_mat_dot_mat(int rows, int cols, REAL *d, int ldd, const REAL *A, int lda,
                            const REAL *B, int ldb, REAL beta) {
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++)
      d[(size_t)i * ldd + j] = beta * d[(size_t)i * ldd + j] + (REAL)1 * A[(size_t)i * lda + j] * B[(size_t)i * ldb + j];
}
static void sigmoid_mat(int rows, int cols, REAL *d, int ldd) {
  for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) d[(size_t)i * ldd + j] = k_sigmoid(d[(size_t)i * ldd + j]);
}
static void tanh_mat(int rows, int cols, REAL *d, int ldd, const REAL *s, int lds) {
  for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) d[(size_t)i * ldd + j] = k_tanh(s[(size_t)i * lds + j]);
}
/* kaldi-matrix.cc:2562-2576: data = diff * value * (1.0 - value) with a DOUBLE literal */
static void diff_sigmoid(int rows, int cols, REAL *d, int ldd, const REAL *y, int ldy) {
  for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) {
    REAL dv = d[(size_t)i * ldd + j], yv = y[(size_t)i * ldy + j];
    d[(size_t)i * ldd + j] = (REAL)((double)(REAL)(dv * yv) * (1.0 - (double)yv));
  }
}
/* kaldi-matrix.cc:2579-2593: data = diff * (1.0 - value*value) */
static void diff_tanh(int rows, int cols, REAL *d, int ldd, const REAL *y, int ldy) {
  for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) {
    REAL dv = d[(size_t)i * ldd + j], yv = y[(size_t)i * ldy + j];
    d[(size_t)i * ldd + j] = (REAL)((double)dv * (1.0 - (double)(REAL)(yv * yv)));
  }
}
static void apply_floor_ceil(int rows, int cols, REAL *d, int ldd, REAL lo, REAL hi) {
  for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) {
    REAL v = d[(size_t)i * ldd + j];
    v = (v < lo ? lo : v);   /* ApplyFloor  kaldi-matrix.cc:1869 */
    v = (v > hi ? hi : v);   /* ApplyCeiling kaldi-matrix.cc:1879 */
    d[(size_t)i * ldd + j] = v;
  }
}

/* ======================================================================== */

lstmp_oracle *lstmp_oracle_create(int I, int C, int R, int S) {
  lstmp_oracle *o = (lstmp_oracle *)calloc(1, sizeof(*o));
  o->I = I; o->C = C; o->R = R; o->S = S; o->W = 7 * C + R;
  o->w_gifo_x = zalloc((size_t)4 * C * I); o->w_gifo_x_corr = zalloc((size_t)4 * C * I);
  o->w_gifo_r = zalloc((size_t)4 * C * R); o->w_gifo_r_corr = zalloc((size_t)4 * C * R);
  o->bias = zalloc((size_t)4 * C);         o->bias_corr = zalloc((size_t)4 * C);
  o->peep_i = zalloc(C); o->peep_f = zalloc(C); o->peep_o = zalloc(C);
  o->peep_i_corr = zalloc(C); o->peep_f_corr = zalloc(C); o->peep_o_corr = zalloc(C);
  o->w_r_m = zalloc((size_t)R * C);        o->w_r_m_corr = zalloc((size_t)R * C);
  o->prev_state = zalloc((size_t)S * o->W);
  o->nthreads = 1;
  return o;
}

void lstmp_oracle_destroy(lstmp_oracle *o) {
  if (!o) return;
  free(o->w_gifo_x); free(o->w_gifo_x_corr); free(o->w_gifo_r); free(o->w_gifo_r_corr);
  free(o->bias); free(o->bias_corr); free(o->peep_i); free(o->peep_f); free(o->peep_o);
  free(o->peep_i_corr); free(o->peep_f_corr); free(o->peep_o_corr);
  free(o->w_r_m); free(o->w_r_m_corr); free(o->prev_state); free(o->prop); free(o->bprop);
  free(o);
}

void lstmp_oracle_set_threads(lstmp_oracle *o, int n) { o->nthreads = n < 1 ? 1 : n; }

long lstmp_oracle_num_params(const lstmp_oracle *o) {
  /* NumParams, ...streams.h:152-160 */
  return (long)4 * o->C * o->I + (long)4 * o->C * o->R + 4 * o->C + 3 * o->C + (long)o->R * o->C;
}

/* flat order of GetParams (...streams.h:162-189): w_gifo_x, w_gifo_r, bias, peep_i, peep_f, peep_o, w_r_m */
static void blob_copy(lstmp_oracle *o, REAL *flat, int to_flat, int corr) {
  REAL *parts[7] = { corr ? o->w_gifo_x_corr : o->w_gifo_x, corr ? o->w_gifo_r_corr : o->w_gifo_r,
                     corr ? o->bias_corr : o->bias, corr ? o->peep_i_corr : o->peep_i,
                     corr ? o->peep_f_corr : o->peep_f, corr ? o->peep_o_corr : o->peep_o,
                     corr ? o->w_r_m_corr : o->w_r_m };
  size_t lens[7] = { (size_t)4 * o->C * o->I, (size_t)4 * o->C * o->R, (size_t)4 * o->C,
                     (size_t)o->C, (size_t)o->C, (size_t)o->C, (size_t)o->R * o->C };
  size_t off = 0;
  for (int p = 0; p < 7; p++) {
    if (to_flat) memcpy(flat + off, parts[p], lens[p] * sizeof(REAL));
    else memcpy(parts[p], flat + off, lens[p] * sizeof(REAL));
    off += lens[p];
  }
}
void lstmp_oracle_set_params(lstmp_oracle *o, const REAL *flat) { blob_copy(o, (REAL *)flat, 0, 0); }
void lstmp_oracle_get_params(lstmp_oracle *o, REAL *flat) { blob_copy(o, flat, 1, 0); }
void lstmp_oracle_set_corr(lstmp_oracle *o, const REAL *flat) { blob_copy(o, (REAL *)flat, 0, 1); }
void lstmp_oracle_get_corr(lstmp_oracle *o, REAL *flat) { blob_copy(o, flat, 1, 1); }

void lstmp_oracle_get_state(lstmp_oracle *o, REAL *dst) { memcpy(dst, o->prev_state, (size_t)o->S * o->W * sizeof(REAL)); }
void lstmp_oracle_set_state(lstmp_oracle *o, const REAL *src) { memcpy(o->prev_state, src, (size_t)o->S * o->W * sizeof(REAL)); }
/* raw views of the two activation slabs [(T+2)S x (7C+R)] for intermediate comparisons */
const REAL *lstmp_oracle_prop_buf(const lstmp_oracle *o) { return o->prop; }
const REAL *lstmp_oracle_bprop_buf(const lstmp_oracle *o) { return o->bprop; }

/* Reset, ...streams.h:212-220 */
int lstmp_oracle_reset(lstmp_oracle *o, const int *flags, int n) {
  if (n != o->S) return -1;  /* KALDI_ASSERT :214 */
  for (int s = 0; s < n; s++)
    if (flags[s] == 1) memset(o->prev_state + (size_t)s * o->W, 0, (size_t)o->W * sizeof(REAL));
  return 0;
}

static void ensure_bufs(lstmp_oracle *o, int T) {
  if (T > o->T_alloc) {
    free(o->prop); free(o->bprop);
    o->prop = zalloc((size_t)(T + 2) * o->S * o->W);
    o->bprop = zalloc((size_t)(T + 2) * o->S * o->W);
    o->T_alloc = T;
  }
}

/* PropagateFnc, ...streams.h:222-332.  in [rows x I] (ld_in), out [rows x R] (ld_out). */
int lstmp_oracle_propagate(lstmp_oracle *o, const REAL *in, int rows, int ld_in, REAL *out, int ld_out) {
  const int S = o->S, C = o->C, R = o->R, I = o->I, W = o->W, nt = o->nthreads;
  if (rows % S != 0) return -1;                /* KALDI_ASSERT :225 */
  const int T = rows / S;
  ensure_bufs(o, T);
  o->T = T;
  REAL *Y = o->prop;
  memset(Y, 0, (size_t)(T + 2) * S * W * sizeof(REAL));            /* Resize(kSetZero) :230 */
  memcpy(Y, o->prev_state, (size_t)S * W * sizeof(REAL));          /* :231 */
  REAL *YG = Y, *YI = Y + C, *YF = Y + 2 * C, *YO = Y + 3 * C, *YC = Y + 4 * C,
       *YH = Y + 5 * C, *YM = Y + 6 * C, *YR = Y + 7 * C;
#define ROWS(p, t) ((p) + (size_t)(t) * S * W)
  /* x -> g,i,f,o all at once :246 ; bias :259 */
  gemm_nt(T * S, 4 * C, I, in, ld_in, o->w_gifo_x, I, (REAL)0, ROWS(YG, 1), W, nt);
  for (int r = 0; r < T * S; r++) { REAL *d = ROWS(YG, 1) + (size_t)r * W; for (int j = 0; j < 4 * C; j++) d[j] += (REAL)1 * o->bias[j]; }

  for (int t = 1; t <= T; t++) {
    REAL *y_g = ROWS(YG, t), *y_i = ROWS(YI, t), *y_f = ROWS(YF, t), *y_o = ROWS(YO, t),
         *y_c = ROWS(YC, t), *y_h = ROWS(YH, t), *y_m = ROWS(YM, t), *y_r = ROWS(YR, t);
    /* r(t-1) -> g,i,f,o :275 */
    gemm_nt(S, 4 * C, R, ROWS(YR, t - 1), W, o->w_gifo_r, R, (REAL)1, y_g, W, nt);
    add_mat_diag_vec(S, C, y_i, W, ROWS(YC, t - 1), W, o->peep_i);      /* :278 */
    add_mat_diag_vec(S, C, y_f, W, ROWS(YC, t - 1), W, o->peep_f);      /* :281 */
    sigmoid_mat(S, C, y_i, W);                                           /* :284 */
    sigmoid_mat(S, C, y_f, W);                                           /* :285 */
    tanh_mat(S, C, y_g, W, y_g, W);                                      /* :288 */
    add_mat_dot_mat(S, C, y_c, W, y_g, W, y_i, W, (REAL)0);              /* :291 */
    add_mat_dot_mat(S, C, y_c, W, ROWS(YC, t - 1), W, y_f, W, (REAL)1);  /* :294 */
    apply_floor_ceil(S, C, y_c, W, (REAL)-50, (REAL)50);                 /* :296-297 */
    tanh_mat(S, C, y_h, W, y_c, W);                                      /* :300 */
    add_mat_diag_vec(S, C, y_o, W, y_c, W, o->peep_o);                   /* :303 */
    sigmoid_mat(S, C, y_o, W);                                           /* :306 */
    add_mat_dot_mat(S, C, y_m, W, y_h, W, y_o, W, (REAL)0);              /* :309 */
    gemm_nt(S, R, C, y_m, W, o->w_r_m, C, (REAL)0, y_r, W, nt);          /* :312 */
  }
  for (int r = 0; r < T * S; r++)                                        /* out = YR[1..T] :328 */
    memcpy(out + (size_t)r * ld_out, ROWS(YR, 1) + (size_t)r * W, (size_t)R * sizeof(REAL));
  memcpy(o->prev_state, ROWS(Y, T), (size_t)S * W * sizeof(REAL));       /* :331 */
  return 0;
}

/* BackpropagateFnc, ...streams.h:334-499.  Uses the prop buffer of the preceding propagate.
 * in_diff may be NULL (first component of a net gets none). */
int lstmp_oracle_backpropagate(lstmp_oracle *o, const REAL *in, int rows, int ld_in,
                               const REAL *out_diff, int ld_od, REAL *in_diff, int ld_id, REAL mmt) {
  const int S = o->S, C = o->C, R = o->R, I = o->I, W = o->W, nt = o->nthreads;
  if (rows % S != 0 || rows / S != o->T) return -1;
  const int T = rows / S;
  REAL *Y = o->prop, *D = o->bprop;
  REAL *YG = Y, *YI = Y + C, *YF = Y + 2 * C, *YO = Y + 3 * C, *YC = Y + 4 * C,
       *YH = Y + 5 * C, *YM = Y + 6 * C, *YR = Y + 7 * C;
  memset(D, 0, (size_t)(T + 2) * S * W * sizeof(REAL));                  /* :352 */
  REAL *DG = D, *DI = D + C, *DF = D + 2 * C, *DO = D + 3 * C, *DC = D + 4 * C,
       *DH = D + 5 * C, *DM = D + 6 * C, *DR = D + 7 * C;
  for (int r = 0; r < T * S; r++)                                        /* DR[1..T] = out_diff :367 */
    memcpy(ROWS(DR, 1) + (size_t)r * W, out_diff + (size_t)r * ld_od, (size_t)R * sizeof(REAL));

  for (int t = T; t >= 1; t--) {
    REAL *y_g = ROWS(YG, t), *y_i = ROWS(YI, t), *y_f = ROWS(YF, t), *y_o = ROWS(YO, t), *y_h = ROWS(YH, t);
    REAL *d_g = ROWS(DG, t), *d_i = ROWS(DI, t), *d_f = ROWS(DF, t), *d_o = ROWS(DO, t),
         *d_c = ROWS(DC, t), *d_h = ROWS(DH, t), *d_m = ROWS(DM, t), *d_r = ROWS(DR, t);
    /* d_r += DGIFO[t+1] * w_gifo_r  :391 ("version 1, precise gradients") */
    gemm_nn(S, R, 4 * C, ROWS(DG, t + 1), W, o->w_gifo_r, R, (REAL)1, d_r, W, nt);
    /* d_m = d_r * w_r_m :408 */
    gemm_nn(S, C, R, d_r, W, o->w_r_m, C, (REAL)0, d_m, W, nt);
    add_mat_dot_mat(S, C, d_h, W, d_m, W, y_o, W, (REAL)0);  diff_tanh(S, C, d_h, W, y_h, W);     /* :411-412 */
    add_mat_dot_mat(S, C, d_o, W, d_m, W, y_h, W, (REAL)0);  diff_sigmoid(S, C, d_o, W, y_o, W);  /* :415-416 */
    /* d_c: five terms :424-428 */
    for (int i = 0; i < S; i++) for (int j = 0; j < C; j++) d_c[(size_t)i * W + j] += (REAL)1 * d_h[(size_t)i * W + j];
    add_mat_dot_mat(S, C, d_c, W, ROWS(DC, t + 1), W, ROWS(YF, t + 1), W, (REAL)1);
    add_mat_diag_vec(S, C, d_c, W, ROWS(DI, t + 1), W, o->peep_i);
    add_mat_diag_vec(S, C, d_c, W, ROWS(DF, t + 1), W, o->peep_f);
    add_mat_diag_vec(S, C, d_c, W, d_o, W, o->peep_o);
    add_mat_dot_mat(S, C, d_f, W, d_c, W, ROWS(YC, t - 1), W, (REAL)0); diff_sigmoid(S, C, d_f, W, y_f, W); /* :431-432 */
    add_mat_dot_mat(S, C, d_i, W, d_c, W, y_g, W, (REAL)0);             diff_sigmoid(S, C, d_i, W, y_i, W); /* :435-436 */
    add_mat_dot_mat(S, C, d_g, W, d_c, W, y_i, W, (REAL)0);             diff_tanh(S, C, d_g, W, y_g, W);    /* :439-440 */
  }
  /* in_diff = DGIFO[1..T] * w_gifo_x :457 */
  if (in_diff) gemm_nn(T * S, I, 4 * C, ROWS(DG, 1), W, o->w_gifo_x, I, (REAL)0, in_diff, ld_id, nt);
  /* gradient / momentum accumulation :465-487 */
  gemm_tn(4 * C, I, T * S, ROWS(DG, 1), W, in, ld_in, mmt, o->w_gifo_x_corr, I, nt);          /* :468 */
  gemm_tn(4 * C, R, T * S, ROWS(DG, 1), W, ROWS(YR, 0), W, mmt, o->w_gifo_r_corr, R, nt);      /* :471 */
  for (int j = 0; j < 4 * C; j++) {                                                             /* AddRowSumMat :474 */
    REAL s = 0; for (int r = 0; r < T * S; r++) s += ROWS(DG, 1)[(size_t)r * W + j];
    o->bias_corr[j] = mmt * o->bias_corr[j] + s;
  }
  for (int j = 0; j < C; j++) {                                                                 /* AddDiagMatMat :477-484 */
    REAL si = 0, sf = 0, so = 0;
    for (int r = 0; r < T * S; r++) {
      si += ROWS(DI, 1)[(size_t)r * W + j] * ROWS(YC, 0)[(size_t)r * W + j];
      sf += ROWS(DF, 1)[(size_t)r * W + j] * ROWS(YC, 0)[(size_t)r * W + j];
      so += ROWS(DO, 1)[(size_t)r * W + j] * ROWS(YC, 1)[(size_t)r * W + j];
    }
    o->peep_i_corr[j] = mmt * o->peep_i_corr[j] + si;
    o->peep_f_corr[j] = mmt * o->peep_f_corr[j] + sf;
    o->peep_o_corr[j] = mmt * o->peep_o_corr[j] + so;
  }
  gemm_tn(R, C, T * S, ROWS(DR, 1), W, ROWS(YM, 1), W, mmt, o->w_r_m_corr, C, nt);              /* :486 */
#undef ROWS
  return 0;
}

static void axpy(size_t n, REAL a, const REAL *x, REAL *y) { for (size_t i = 0; i < n; i++) y[i] += a * x[i]; }
static void clip(size_t n, REAL *x, REAL th) { for (size_t i = 0; i < n; i++) { x[i] = x[i] < -th ? -th : x[i]; x[i] = x[i] > th ? th : x[i]; } }

/* Update, ...streams.h:501-512.  clip_grad > 0 reproduces standard/nnet/nnet-lstm-projected.h:480-493
 * (in-place +-clip of every *_corr element before the step). */
void lstmp_oracle_update(lstmp_oracle *o, REAL lr, REAL clip_grad) {
  const size_t C = o->C, I = o->I, R = o->R;
  if (clip_grad > (REAL)0) {
    clip(4 * C * I, o->w_gifo_x_corr, clip_grad); clip(4 * C * R, o->w_gifo_r_corr, clip_grad);
    clip(4 * C, o->bias_corr, clip_grad);
    clip(C, o->peep_i_corr, clip_grad); clip(C, o->peep_f_corr, clip_grad); clip(C, o->peep_o_corr, clip_grad);
    clip(R * C, o->w_r_m_corr, clip_grad);
  }
  axpy(4 * C * I, -lr, o->w_gifo_x_corr, o->w_gifo_x);
  axpy(4 * C * R, -lr, o->w_gifo_r_corr, o->w_gifo_r);
  axpy(4 * C, -lr, o->bias_corr, o->bias);
  axpy(C, -lr, o->peep_i_corr, o->peep_i);
  axpy(C, -lr, o->peep_f_corr, o->peep_f);
  axpy(C, -lr, o->peep_o_corr, o->peep_o);
  axpy(R * C, -lr, o->w_r_m_corr, o->w_r_m);
}

int lstmp_oracle_sizeof_real(void) { return (int)sizeof(REAL); }
