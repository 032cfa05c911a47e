/*
 * include/klstm.h -- C-ABI of the MI355X-native LstmProjectedStreams engine.
 *
 * This is the drop-in boundary for ONE path of dophist/kaldi-lstm: the multi-stream LSTMP
 * forward + truncated BPTT + SGD update of
 *     google/nnet/bd-nnet-lstm-projected-streams.h   (class LstmProjectedStreams)
 * It replaces, for that component only, the google/cudamatrix (CuMatrix + cuBLAS +
 * bd-cu-kernels.cu) layer the reference component calls into.  A Kaldi Component that
 * forwards its virtuals to these entry points is shown in INTEGRATION.md; the C++ mirror of
 * the reference class lives in include/klstm_component.hpp.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no exceptions cross this boundary.  Every call
 *     returns a klstm_status; klstm_last_error() gives the message of the calling thread's
 *     last failure (the adapter maps non-zero to KALDI_ERR, i.e. std::runtime_error, which is
 *     how the reference reports CU_SAFE_CALL failures, cu-matrix.cc:813).
 *   - all matrices are fp32 (Kaldi BaseFloat), row-major with an explicit row stride in
 *     ELEMENTS (CuMatrixBase::Stride(), cu-matrix.h:479-489: rows are pitched, always honour
 *     the stride).  Minibatch matrices are time-major: row = t*num_stream + s
 *     (bd-nnet-train-lstm-streams.cc:191-199; ...streams.h:263-272).
 *   - `in`, `out`, `out_diff`, `in_diff` are DEVICE pointers owned and pre-sized by the caller
 *     (Nnet owns them in Kaldi); the engine owns parameters, gradient/momentum buffers, the
 *     carried stream state and the activation slabs (...streams.h:577-620).
 *   - one engine = one GPU = one HIP stream; calls on one handle must be host-serialised
 *     (the reference is single-threaded, bd-nnet-train-lstm-streams.cc:93).  All calls are
 *     asynchronous on the engine's stream except the *_host copies and klstm_synchronize.
 *   - the flat parameter/gradient blob order is GetParams order (...streams.h:162-189):
 *       w_gifo_x [4C x I] | w_gifo_r [4C x R] | bias [4C] | peephole_i_c [C] |
 *       peephole_f_c [C] | peephole_o_c [C] | w_r_m [R x C]      (4C rows ordered g,i,f,o)
 */
#ifndef KLSTM_H_
#define KLSTM_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct klstm_engine klstm_engine;

typedef enum {
  KLSTM_OK = 0,
  KLSTM_ERR_ARG = 1,    /* bad argument (null pointer, negative size)                     */
  KLSTM_ERR_SHAPE = 2,  /* KALDI_ASSERT-class violation: rows % num_stream != 0 (:225),
                           reset flag count != num_stream (:214), bwd rows != fwd rows     */
  KLSTM_ERR_STATE = 3,  /* call order violated (backpropagate without propagate)           */
  KLSTM_ERR_HIP = 4,    /* a HIP runtime call failed                                       */
  KLSTM_ERR_NOGPU = 5   /* no usable gfx950 device: there is NO CPU fallback               */
} klstm_status;

/* flags for klstm_backpropagate */
#define KLSTM_BPTT_DEFAULT        0
/* data-parallel mode: leave the pure local gradient (beta = 0) in the gradient blob and do
 * NOT touch the momentum buffers; the caller all-reduces klstm_grad_blob() and then calls
 * klstm_apply_momentum().  (The reference folds momentum into the gradient GEMM's beta,
 * ...streams.h:465-487, which would multiply momentum by the number of ranks.) */
#define KLSTM_BPTT_DEFER_MOMENTUM 1
/* "klstm_update follows immediately, with the input rows unchanged": Kaldi's Component::Backpropagate runs
 * BackpropagateFnc and then Update on the same (input, out_diff) pair, so the gradient products (:468-487) may wait for
 * the learning rate and run as ONE pass together with the Update (:504-512): corr = momentum*corr + grad, theta -= lr*corr,
 * transposed copies refreshed -- one launch and 16 MB of HBM traffic less per minibatch.  Results are identical.  The
 * engine keeps the `in` pointer until then: any other call that observes or changes gradients, momentum buffers,
 * parameters or activations in between (get_corr, set_params, propagate, ...) first runs the gradient products the
 * ordinary way, so only a caller that overwrites the rows of `in` before klstm_update breaks the contract.
 * Ignored together with KLSTM_BPTT_DEFER_MOMENTUM. */
#define KLSTM_BPTT_FUSE_UPDATE    2

/* Message of the calling thread's most recent failing call ("" if none). */
const char *klstm_last_error(void);
/* Library / kernel-arch identification string, e.g. "klstm 0.4 gfx950 (...)". */
const char *klstm_version(void);

/* Replaces: LstmProjectedStreams(input_dim, output_dim) + <CellDim>/<NumStream> of
 * InitData/ReadData (...streams.h:27-33, :55-99, :101-131).  recur_dim == output_dim.
 * Parameters start at zero, momentum buffers and stream state at zero (kSetZero, :76-97).
 * device: HIP device ordinal.  hip_stream: a hipStream_t to run on (e.g. the caller
 * framework's current stream; must not be the legacy NULL stream, graphs cannot be captured
 * on it) or NULL to use the library's process-wide per-device BLOCKING stream: it is shared by
 * all engines created this way (stacked LSTM components stay ordered with each other) and is
 * implicitly ordered with the legacy default stream like everything else in a Kaldi process. */
klstm_status klstm_create(int input_dim, int cell_dim, int recur_dim, int num_stream,
                          int device, void *hip_stream, klstm_engine **out);
void klstm_destroy(klstm_engine *e);

/* dims / NumParams (...streams.h:152-160) */
int klstm_input_dim(const klstm_engine *e);
int klstm_cell_dim(const klstm_engine *e);
int klstm_recur_dim(const klstm_engine *e);
int klstm_num_stream(const klstm_engine *e);
long klstm_num_params(const klstm_engine *e);

/* Parameter / momentum / gradient blobs, flat GetParams order.  *_host variants take host
 * pointers and synchronise; the device variants are stream-ordered.
 * Replaces ReadData's tensor loads (:109-117), GetParams (:162-189) and the *_corr_ members
 * that InfoGradient inspects (:201-210). */
klstm_status klstm_set_params_host(klstm_engine *e, const float *flat);
klstm_status klstm_get_params_host(klstm_engine *e, float *flat);
klstm_status klstm_set_params_device(klstm_engine *e, const float *flat_dev);
klstm_status klstm_get_corr_host(klstm_engine *e, float *flat);   /* momentum buffers *_corr_ */
klstm_status klstm_set_corr_host(klstm_engine *e, const float *flat);
klstm_status klstm_get_grads_host(klstm_engine *e, float *flat);  /* pure gradient (DP mode) */
/* Device address of the contiguous gradient blob (num_params floats) for an in-place
 * all-reduce (RCCL ncclAllReduce / torch.distributed.all_reduce) in DP mode. */
float *klstm_grad_blob(klstm_engine *e);
/* Floats a collective over the engine's OWN gradient blob should cover: num_params rounded up to a multiple of 4, + 4.  The first of
 * the extra floats is the VALIDITY WORD of data-parallel runs: the gradient kernel writes 0 (this rank's gradient is real) or 1 (a
 * persistent launch of this minibatch gave up and the device-side guard stopped the gradient products); summed over the ranks together
 * with the gradient, it tells every rank's momentum / Update kernels after the all-reduce whether ANY rank's gradient was not real, and
 * then EVERY rank leaves that Update out -- replicas stay identical without a host wait in front of the collective
 * (klstm_allreduce_grads below).  A caller that runs its own all-reduce over num_params floats only (torch.distributed on
 * klstm_grad_blob()) does not get this: it sets "persist_verify" = 1 instead.  With a blob bound by klstm_bind_grad_blob: num_params. */
long klstm_grad_blob_len(const klstm_engine *e);
/* Use caller-owned device storage (num_params floats, 16-byte aligned) as this engine's gradient blob from now
 * on: a stacked net places the blobs of all its layers back to back in ONE buffer so that a minibatch needs one
 * all-reduce for the whole model.  NULL returns to the engine's own buffer. */
klstm_status klstm_bind_grad_blob(klstm_engine *e, float *grad_dev);
float *klstm_param_blob(klstm_engine *e);

/* Reset (...streams.h:212-220): for every s with flags[s] == 1 zero stream s's carried
 * state.  n must equal num_stream (KALDI_ASSERT :214 -> KLSTM_ERR_SHAPE). */
klstm_status klstm_reset(klstm_engine *e, const int *flags, int n);
/* Carried state of prev_nnet_state_ that is ever consumed: c [S x C] and r [S x R]
 * (only these column groups are read back, :275,:278,:281,:294). Host pointers. */
klstm_status klstm_get_state_host(klstm_engine *e, float *c, float *r);
klstm_status klstm_set_state_host(klstm_engine *e, const float *c, const float *r);

/* PropagateFnc (...streams.h:222-332).  in [rows x I], out [rows x R], device pointers,
 * strides in elements.  rows % num_stream must be 0 (:225).  Carries state in and out
 * (:231, :331). */
klstm_status klstm_propagate(klstm_engine *e, const float *in, int rows, int in_stride,
                             float *out, int out_stride);

/* BackpropagateFnc (...streams.h:334-499) for the minibatch of the immediately preceding
 * klstm_propagate (it reuses that call's activation slab, :342-349; rows must match).
 * in_diff may be NULL (the reference always computes it, :457; NULL skips that GEMM).
 * momentum = opts_.momentum (:465).  flags: KLSTM_BPTT_*.
 * Default mode leaves  corr = momentum*corr + grad  in the momentum buffers (:468-487). */
klstm_status klstm_backpropagate(klstm_engine *e, const float *in, int in_stride,
                                 const float *out_diff, int out_diff_stride,
                                 float *in_diff, int in_diff_stride, int rows,
                                 float momentum, int flags);

/* The same two calls for HOST matrices (a Kaldi CuMatrix holds host memory when the process runs with
 * --use-gpu=no, cu-matrix.h:479-481; SURVEY 8(b) "data type at the boundary").  Rows are staged through
 * engine-owned device buffers with stream-ordered 2-D copies, the computation is the device path above (there is
 * no CPU path), and the call returns after out / in_diff have landed in host memory. */
klstm_status klstm_propagate_host(klstm_engine *e, const float *in_host, int rows, int in_stride,
                                  float *out_host, int out_stride);
klstm_status klstm_backpropagate_host(klstm_engine *e, const float *in_host, int in_stride,
                                      const float *out_diff_host, int out_diff_stride,
                                      float *in_diff_host, int in_diff_stride, int rows,
                                      float momentum, int flags);
/* 1 if p is device-accessible memory of the engine's GPU (hipMalloc / managed / registered host memory),
 * 0 if it is plain host memory, <0 on error: lets an adapter pick the right pair of calls once per matrix. */
int klstm_pointer_on_device(const klstm_engine *e, const void *p);

/* Data parallelism over utterance streams: every rank runs klstm_backpropagate(.., KLSTM_BPTT_DEFER_MOMENTUM),
 * then ONE in-place fp32 sum over the ranks of the gradient blob, then klstm_apply_momentum + klstm_update on every rank.
 *   klstm_allreduce_grads   ncclAllReduce(sum, fp32) of this engine's gradient blob (also a blob bound with
 *                           klstm_bind_grad_blob) over the ranks of `rccl_comm` (an ncclComm_t), enqueued on the engine's stream.
 *                           A rank whose persistent chain gave up in this minibatch must not hand its (missing) gradient to the
 *                           others: with the engine's own blob the validity word of klstm_grad_blob_len() rides along -- no host
 *                           wait; the minibatch is left out by every rank and counted (klstm_profile_query "dp_updates_left_out"),
 *                           and what the host has already heard of is answered (run again) before the collective is enqueued;
 *                           with "persist_verify" = 1 or a bound blob the call waits for the engine's stream and answers a
 *                           give-up first, so that every rank reduces a real gradient.
 *   klstm_allreduce_buffer  the same for any device buffer of n floats, e.g. the fused blob of a stacked net, on hip_stream
 *   klstm_comm_*            thin pass-throughs to ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy for callers without RCCL
 *                           headers: rank 0 obtains the 128-byte id and hands it to the other ranks by its own means (file,
 *                           MPI, the launcher's store), every rank then calls klstm_comm_init_rank (collective).
 * RCCL is resolved in the running process at the first of these calls (no link-time dependency of libklstm.so). */
#define KLSTM_COMM_ID_BYTES 128
klstm_status klstm_comm_get_unique_id(void *id128);
klstm_status klstm_comm_init_rank(int device, int nranks, int rank, const void *id128, void **comm);
klstm_status klstm_comm_destroy(void *comm);
klstm_status klstm_comm_count(void *comm, int *nranks);       /* ncclCommCount: how many ranks the communicator really spans */
klstm_status klstm_allreduce_grads(klstm_engine *e, void *rccl_comm);
klstm_status klstm_allreduce_buffer(float *buf_dev, size_t n, void *rccl_comm, void *hip_stream);

/* One-shot all-reduce over peer-mapped blobs (kaldi-lstm_amd/csrc/klstm_oneshot.hip).  PREPARED AND OFF: nothing in
 * the library calls it, and it has never run across devices (one-GPU lease) -- only as a 1-rank self-loop and between two
 * processes on one GPU.  Every rank: create (its own blob of n floats, hipMalloc memory), export two handles, hand them to every
 * peer by any means, connect with everybody's handles in rank order, then once per minibatch klstm_oneshot_allreduce (in place,
 * on hip_stream; all ranks, same order).  Sums are added in rank order by ONE rank per 1/N slice: bit-identical on all ranks.
 * Every wait is bounded (timeout_ms, default 2000); klstm_oneshot_status reads 0 or the phase that expired (after a stream
 * synchronisation). */
typedef struct klstm_oneshot klstm_oneshot;
typedef struct { unsigned char bytes[80]; } klstm_ipc_handle;     /* hipIpcMemHandle_t of the allocation + the blob's offset in it */
klstm_status klstm_oneshot_create(int device, float *blob_dev, long n, klstm_oneshot **out);
klstm_status klstm_oneshot_export(klstm_oneshot *g, klstm_ipc_handle *blob, klstm_ipc_handle *flags);
klstm_status klstm_oneshot_connect(klstm_oneshot *g, int rank, int nranks, const klstm_ipc_handle *blobs, const klstm_ipc_handle *flags);
klstm_status klstm_oneshot_allreduce(klstm_oneshot *g, void *hip_stream, int timeout_ms);
klstm_status klstm_oneshot_status(klstm_oneshot *g, unsigned *status);
klstm_status klstm_oneshot_destroy(klstm_oneshot *g);
const char *klstm_oneshot_last_error(void);
/* on the engine's stream, behind its gradient products; the group was created on klstm_grad_blob(engine) over klstm_grad_blob_len(engine)
 * floats (the validity word rides along, as in klstm_allreduce_grads) or over num_params floats (then the call waits and looks first) */
klstm_status klstm_allreduce_grads_oneshot(klstm_engine *e, klstm_oneshot *group, int timeout_ms);

/* DP mode only, after the all-reduce of klstm_grad_blob():  corr = momentum*corr + grad. */
klstm_status klstm_apply_momentum(klstm_engine *e, float momentum);

/* Update (...streams.h:501-512):  theta -= learn_rate * theta_corr  for all seven tensors.
 * clip_grad > 0 first clips every corr element to +-clip_grad IN PLACE, which is the
 * standard/ LstmProjected::Update behaviour (standard/nnet/nnet-lstm-projected.h:480-493,
 * max_grad = 50); pass 0 for the google/ component. */
klstm_status klstm_update(klstm_engine *e, float learn_rate, float clip_grad);

/* Block until everything queued on the engine's stream has finished. */
klstm_status klstm_synchronize(klstm_engine *e);

/* Test / diagnostics hook (the reference's DEBUG dumps, :314-324, :443-453): copy the
 * activation slab of the last propagate (which = 0) or backpropagate (which = 1) to host in
 * the REFERENCE layout [(T+2)*S rows] x [G|I|F|O|C|H|M|R], T = rows/S of that call.
 * Row-blocks the engine never materialises (the slab's dummy blocks) read as zero. */
klstm_status klstm_get_activations_host(klstm_engine *e, int which, float *dst);

/* Stateless helpers for the two trivial components the reference puts in front of the LSTM
 * (README.md:46-49).  Device pointers, strides in elements, enqueued on hip_stream (NULL = default).
 *   TimeShift::PropagateFnc (standard/nnet/nnet-time-shift.h:42-51): out[t] = in[clamp(t + shift, 0, rows-1)]
 *   TransmitComponent::PropagateFnc / BackpropagateFnc (standard/nnet/nnet-transmit-component.h:26-33):
 *   identity = shift 0. */
klstm_status klstm_time_shift(const float *in, int rows, int cols, int in_stride, float *out,
                              int out_stride, int shift, void *hip_stream);

/* Device memory helpers so that FFI users (the C++ mirror, ctypes) need no HIP headers.  The memcpy
 * calls synchronise hip_stream. */
klstm_status klstm_malloc(void **p, size_t bytes);
klstm_status klstm_free(void *p);
klstm_status klstm_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *hip_stream);
klstm_status klstm_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *hip_stream);
klstm_status klstm_memset_zero(void *dst_dev, size_t bytes, void *hip_stream);
klstm_status klstm_stream_synchronize(void *hip_stream);

/* Output tail of the nnet the trainer runs behind the LSTM (README.md:24-29): stateless ops on device
 * matrices, strides in elements.  The AffineTransform / Softmax classes themselves are NOT vendored in the
 * reference ([UPSTREAM] nnet-affine-transform.h, nnet-activation.h); only their use is (google/nnet.proto:4-5).
 *   affine_propagate     out = in * W^T + bias                (W is [out_dim x in_dim], Kaldi's linearity_)
 *   affine_backpropagate in_diff = out_diff * W
 *   affine_update        W_corr = momentum*W_corr + out_diff^T * in ; bias_corr = momentum*bias_corr + colsum(out_diff);
 *                        W -= lr * W_corr ; bias -= lr_bias * bias_corr
 *   softmax              per-row softmax
 *   xent_eval_masked     Xent::EvalMasked, google/nnet/nnet-loss.cc:76-142, for one-hot targets:
 *                        diff = (net_out - onehot(target)) * mask ; row_xent[r] = -mask*log(net_out[r][target]) ;
 *                        row_correct[r] = (mask == 1 && argmax(net_out[r]) == target).  The caller sums the two
 *                        per-row arrays on the host (the reference also copies scalars back, :110-141). */
klstm_status klstm_affine_propagate(const float *in, int rows, int in_dim, int in_stride, const float *W,
                                    const float *bias, float *out, int out_dim, int out_stride, void *hip_stream);
klstm_status klstm_affine_backpropagate(const float *out_diff, int rows, int out_dim, int od_stride,
                                        const float *W, int in_dim, float *in_diff, int id_stride, void *hip_stream);
klstm_status klstm_affine_update(const float *in, int in_stride, const float *out_diff, int od_stride, int rows,
                                 int in_dim, int out_dim, float *W, float *bias, float *W_corr, float *bias_corr,
                                 float lr, float lr_bias, float momentum, void *hip_stream);
/* data-parallel pieces of the same layer: the pure local gradient (no momentum, no update) into caller storage,
 * and the post-all-reduce step  corr = momentum*corr + grad ; param -= lr*corr  on any flat tensor */
klstm_status klstm_affine_gradient(const float *in, int in_stride, const float *out_diff, int od_stride, int rows,
                                   int in_dim, int out_dim, float *W_grad, float *bias_grad, void *hip_stream);
klstm_status klstm_sgd_momentum_update(float *param, float *corr, const float *grad, long n, float momentum, float lr,
                                       void *hip_stream);
klstm_status klstm_softmax(const float *in, int rows, int cols, int in_stride, float *out, int out_stride,
                           void *hip_stream);
klstm_status klstm_xent_eval_masked(const float *net_out, int rows, int cols, int stride, const int *targets_dev,
                                    const float *mask_dev, float *diff, int diff_stride, float *row_xent_dev,
                                    float *row_correct_dev, void *hip_stream);

/* Softmax followed by Xent::EvalMasked (one-hot targets) as ONE pass over the rows: the same arithmetic in the same order as
 * klstm_softmax + klstm_xent_eval_masked (bit-identical diff and statistics), without the trip of the posterior matrix
 * through memory.  post may be NULL (a training step only needs diff); if not NULL it receives the softmax output.
 * Rows the one-pass kernel does not serve (cols % 4 != 0, cols < 2048 or > 32768, unaligned rows) run the two kernels through
 * `post`, which must then be given (KLSTM_ERR_ARG otherwise).  totals_dev (may be NULL): three doubles on the device that the
 * statistics of this minibatch are added to, as klstm_xent_accumulate does, inside the same launch (the workgroup that
 * finishes last adds the rows up in a fixed order). */
klstm_status klstm_softmax_xent_masked(const float *net_in, int rows, int cols, int in_stride, float *post, int post_stride,
                                       const int *targets_dev, const float *mask_dev, float *diff, int diff_stride,
                                       float *row_xent_dev, float *row_correct_dev, double *totals_dev, void *hip_stream);

/* The three per-minibatch statistics of Xent::EvalMasked added onto device totals (the reference adds them to loss_, frames_,
 * correct_ on the host after copying the scalars back, google/nnet/nnet-loss.cc:110-142): totals_dev[0] += sum row_xent,
 * totals_dev[1] += sum row_correct, totals_dev[2] += sum mask, in double, fixed order, no synchronisation.  A trainer that
 * reports every N minibatches (bd-nnet-train-lstm-streams.cc:240-257) reads the three doubles once per report. */
klstm_status klstm_xent_accumulate(const float *row_xent_dev, const float *row_correct_dev, const float *mask_dev, int rows,
                                   double *totals_dev, void *hip_stream);

/* Xent::EvalMasked for GENERAL posteriors (google/nnet/nnet-loss.cc:76-142): frame r carries the entries
 * post_pdf/post_weight[post_offsets[r] .. post_offsets[r+1]) (CSR, device arrays; repeated pdfs of a frame add up like the
 * reference's `tgt(t, pdf) += weight`, :86-96).  diff = (net_out - target) * mask; per row: cross entropy -mask*sum t*log(y),
 * target entropy -mask*sum t*log(t + 1e-20), and 1 if mask == 1 and the arg-maxima of net_out and of the target row agree
 * (lowest index on ties, zeros of the dense target row included).  The dense rows x cols target matrix of the reference
 * (5.3 MB per minibatch at 80 x 16624, built on the host and copied) never exists.  pdf range checks are the caller's
 * (the reference raises on the host while it builds the matrix, :89-92). */
klstm_status klstm_xent_eval_masked_post(const float *net_out, int rows, int cols, int stride, const int *post_offsets_dev,
                                         const int *post_pdf_dev, const float *post_weight_dev, const float *mask_dev, float *diff,
                                         int diff_stride, float *row_xent_dev, float *row_entropy_dev, float *row_correct_dev,
                                         void *hip_stream);

/* Engine knobs (not part of the reference interface).  Keys:
 *   "graph"   0/1/2  issue plain stream launches (default 0: measured equal or faster at every stream count while the host
 *                  thread keeps ahead, and indifferent to callers that hand in fresh buffers every minibatch) or replay the
 *                  per-call launch sequence from a hipGraph keyed on the caller's pointers (a busy host thread: 40-80
 *                  launches per call on the launch-per-step chains).  1: a graph per call unless the call is one or
 *                  two launches anyway (persistent chain); 2: always
 *   "fold"    -1/0/1  folded recurrence W_rm = W_gifo_r * W_r_m: one kernel per step and direction instead of two
 *                  (DESIGN.md 4a); -1 = auto (NumStream <= 8 and >= 12 frames per stream), 1 = whenever NumStream <=
 *                  16 and I, C, R are multiples of 8.  Same results up to fp32 summation order.
 *   "persist" -1/0/1/2  weights-resident persistent chain for NumStream <= 8: ONE launch per direction runs
 *                  all T steps with the folded operands held in registers and the per-step all-to-all done inside the
 *                  launch.  0 = off, 1 = forward launch only, 2 = both directions whenever the shape allows, -1 = auto (both
 *                  directions from 8 frames per stream).  Same results up to fp32 summation order.  Needs one compute unit
 *                  per workgroup (C/4 backward, C/4 or C/8 forward): an engine on a device (partition) with fewer compute
 *                  units keeps to one launch per step, and klstm_last_error() says so when the option asked for it.
 *                  Every in-kernel wait is bounded.  A launch that GIVES UP (its workgroups were not all resident: another
 *                  process holds part of the chip) records its ordinal and a status word; on the device, every persistent
 *                  launch and every gradient / momentum / Update kernel queued behind it sees the word and does nothing, and
 *                  the carried state the minibatch started from is intact (double-buffered).  The first call that looks
 *                  (propagate / backpropagate / update / reset poll a host-mapped word without a synchronisation;
 *                  klstm_synchronize, the getters and klstm_allreduce_grads read the device words) answers it WITHOUT an
 *                  error: the calls of the CURRENT minibatch -- since the last klstm_propagate / klstm_reset began -- are run
 *                  again on the launch-per-step chain with the arguments they came with (bit-identical to an engine that never
 *                  used the persistent chain), so `in`, `out_diff` and the output buffers of a minibatch must stay the
 *                  caller-provided, unchanged buffers until the next klstm_propagate / klstm_reset on the engine (Kaldi's
 *                  Nnet keeps them exactly that long).  A minibatch older than that cannot be run again: it is DROPPED -- no
 *                  Update, no state advance -- and counted.  `persist_cooldown` minibatches (default 64) on the launch-per-step
 *                  chain follow, then the persistent chain is tried again; a give-up that follows a re-arm closely doubles the
 *                  cool-down (up to 65536 minibatches), a clean run as long as the last cool-down resets it.  klstm_last_error() carries a remark;
 *                  klstm_profile_query(e, "persist_giveups" | "persist_replayed" | "persist_dropped" | "persist_launches" |
 *                  "dp_updates_left_out", ..) returns the counts in *launches (no "profile" option needed).
 *                  "persist_verify" 0/1: 1 = klstm_propagate / klstm_backpropagate wait for their persistent launch, so a
 *                  give-up is answered INSIDE the call, before the caller (or a neighbouring component) has read `out` /
 *                  `in_diff`: fully transparent, at the price of one host wait per call (~6 us of launch gap each; Kaldi
 *                  synchronises per minibatch anyway: the C++ mirror turns it on).  0 (default of the C-ABI): asynchronous, as above.
 *                  The wait is a spin on a host-mapped word the last workgroup of the launch writes (launch count + give-up bit),
 *                  not a stream synchronisation ("persist_verify_spin" 0: hipStreamSynchronize instead; A-B runs).
 *                  With KLSTM_BPTT_FUSE_UPDATE ("klstm_update follows immediately") the wait of klstm_backpropagate happens at the
 *                  END OF THAT klstm_update instead: gradient products + Update (guarded: they do nothing behind a give-up) are
 *                  enqueued while the BPTT launch runs, `in_diff` is valid when klstm_update has returned -- Kaldi's
 *                  Component::Backpropagate returns only then.  (A promised klstm_update that never comes: the next
 *                  klstm_propagate / synchronising call waits and answers.)
 *                  Not supported: a hipGraph captured by the CALLER around engine calls that take the persistent chain (the
 *                  engine cannot count launches inside a foreign graph): use "persist" = 0 there.
 *                  Per-engine knobs (A-B experiments and tests): "persist_waves" (forward and backward geometry: 8, 12, 16),
 *                  "persist_xl" (bf16, 9..32 streams, 1024 cells: 1 = the forward launch as one chain per XCD, the default; 0 = one
 *                  copy of the weights over all CUs), "persist_xl_bwd" (the same for the BPTT chain; 0 = one launch per step),
 *                  "persist_bwd_waves" (12, 16), "persist_bwd_interleave" (5..8 streams: 1 = the two stream groups of the backward
 *                  launch as interleaved chains, the default; 0 = one after the other; bit-identical results),
 *                  "persist_tpw", "persist_nap0", "persist_nap", "persist_nap0_bwd",
 *                  "persist_spin_us" (bound of a single in-kernel wait, default 50 000), "persist_ncu" (pretend CU count),
 *                  "persist_test_stall_fwd" / "persist_test_stall_bwd" (workgroup 0 withholds its publish of that step:
 *                  forces the give-up path)
 *   "fold_bf16x3"  0/1/2  the fold product W_rm = W_gifo_r * W_r_m of THIS engine on the 16-bit matrix cores at fp32 accuracy
 *                  (every partial product exact in fp32, fp32 accumulation): 2 (default) = two fp16 planes per
 *                  operand, three products, dropped terms ~7e-7 relative; 1 = three bf16 planes, six products, dropped terms
 *                  below 2^-24; 0 = the fp32 MFMA kernel.  RANGE: identical to the reference's fp32 products in every mode -- in
 *                  mode 2 a parameter at or beyond 65520 (the fp16 range) is noticed by the product itself (range guard, below),
 *                  the affected tiles are recomputed in fp32 and the engine moves to mode 1 by itself.  Parameters below 2^-14
 *                  carry an absolute error of 2^-36 (not a relative one of 2^-22): nothing for a weight matrix whose largest
 *                  entries are above 1e-4.  "fold_direct" 0: the generic tile kernel (process-wide)
 *   "direct_nt_shape", "outer_f16", "skinny_f16", "skinny_f16_pair"  the three products of a WIDE AffineTransform at few frames
 *                  (the output layer of a small-minibatch step) run on the f16 matrix cores at fp32 accuracy:
 *                  both operands split into two fp16 numbers on the fly (x = h1 + h2 / 2048), three products with fp32
 *                  accumulation, the dropped term ~2^-22 relative; measured error against float64 at or below the fp32 MFMA
 *                  kernels'.  RANGE GUARD: the numeric range is the reference's (fp32).  Upper side: an operand at or beyond
 *                  65520 becomes Inf in its first plane and every accumulator it meets Inf / NaN -- never a finite wrong
 *                  number; each wave looks at its accumulators after its K loop, recomputes its outputs in plain fp32 when it
 *                  finds one, and counts the event in a host-mapped word; the launcher reads the word before every launch and
 *                  keeps that product on its fp32 kernel from then on (klstm_profile_query(e, "fp16_redo" | "fp16_redo_nt" |
 *                  "fp16_redo_outer" | "fp16_redo_skinny" | "fp16_redo_fold", ..) returns the counts in *launches).  Lower side:
 *                  DERIVATIVE operands are scaled by a power of two before the split and the result scaled back (exact): every
 *                  column of out_diff by its own in klstm_affine_gradient / _update, out_diff / dgifo by 2^12 in
 *                  klstm_affine_backpropagate and the BPTT tail -- entries of 1e-7 (late training) keep fp32 accuracy (22 bits
 *                  down to 2^-26, an absolute error below 2^-47 under that; derivatives of 16 and more take the guard's path).
 *                  klstm_affine_propagate of <= 80 rows into > 8192 columns: "direct_nt_shape" 99 = the same layout on the fp32
 *                  MFMA, 0 = the register-direct fp32 kernel, 10*NI + waves = geometry of that kernel, 97 = the f16 x 2 product with the
 *                  input rows resident in registers and K cut into four quarters (round 6: measured, not faster at 80 rows);
 *                  klstm_affine_gradient / klstm_affine_update of <= 96 rows and >= 2048 outputs: "outer_f16" 0 = fp32 tiles;
 *                  klstm_affine_backpropagate of <= 80 rows over >= 4096 outputs: "skinny_f16" 0 = fp32 MFMA;
 *                  d_r / in_diff of an engine whose input is too wide for the persistent backward launch: "skinny_f16_pair" 0 =
 *                  the tiled split-K kernel.  All process-wide (A-B experiments and tests)
 *   "fp16_products"  0/1  0 = no product of THIS engine and no stateless klstm_affine_* call on this engine's device runs on fp16 planes
                  (all of the above on their fp32 kernels, the fold product on three bf16 planes): saves a net whose activations,
                  weights or derivatives pass 65504 all the time the slow call of the range guard.  1 = the defaults again, the
                  guards' counters and cool-downs cleared.  Other engines keep their own state.
                  RANGE GUARD STATE: every engine has its own (event words, which families are latched, cool-downs); the stateless
                  klstm_affine_* calls share one per device.  A family whose guard fired runs on its fp32-range kernel for a
                  cool-down (64 fold products = Updates; 2048 looks of a stateless family), then the fp16 planes are tried again; a
                  trigger that follows a re-arm closely doubles the cool-down (up to 2^20), a clean run as long as the last cool-down
                  resets it; klstm_last_error() carries a remark when a family latches and when it comes back.
                  klstm_profile_query(e, "fp16_redo*") = this engine's events + its device's stateless events; "fp16_redo_own" =
                  this engine's only; "fold_mode" = the fold product's format as it runs (1 while latched)
 *   "persist_tail"  0/1/2  d_r / in_diff inside the persistent backward launch (1, default) or as batched products after it (0).
 *                  Inside: on TAIL WORKGROUPS of the same launch where the device has compute units to spare next to the chain's
 *                  C / 4 (they read the chain's exchange without being waited for: the chain runs at its bare pace), else --
 *                  or with 2 -- on the chain's own workgroups (rounds 3-5).  Same contraction order either way: bit-identical results.
 *                  klstm_profile_query(e, "persist_tail_wgs") = tail workgroups of the last backward launch (0: none).
 *   "bf16"    0/1/2  bf16 operands (weights, staged activations, gradient products from 256 frames on) with fp32
 *                  accumulate, fp32 masters (the reference is fp32 only).  Needs I, C, R multiples of 8.
 *                  1 = where it pays: from 9 streams on; an engine of up to 8 streams keeps its fp32 weights-resident chain (faster
 *                  there, and exact; remark in klstm_last_error()).  2 = bf16 operands at any stream count.
 *   "tail_merge"  0/1  1 = the reduction of the tail workgroups' partial d_r / in_diff rows runs on the first workgroups of the gradient
 *                  launch behind the BPTT launch instead of in a launch of its own (k_tail_reduce).  With KLSTM_BPTT_FUSE_UPDATE that launch is
 *                  klstm_update's: `in_diff` is then complete when klstm_update's launches are (klstm_synchronize in between runs it
 *                  at once).  Bit-identical; measured SLOWER (146.0 -> 148.1 us per minibatch at 40/800/512 x 4): default 0, kept for A-B runs.
 *                  klstm_profile_query "tail_merge_launches", "tail_merge_timeouts" (must stay 0).
 *   "fuse_update"  0/1  0 = KLSTM_BPTT_FUSE_UPDATE is ignored: gradient products and Update as separate passes (A-B runs; this engine)
 *   "gemm_copies"  0/1/2  bf16 mode, 9..32 streams with the per-XCD BPTT chain: that chain writes a bf16 copy of its dgifo rows, the Update
 *                  kernels bf16 copies of W_gifo_r^T / W_gifo_x^T, and the batched d_r + in_diff product reads THE COPIES by LDS-DMA
 *                  (1, default; 0: it rounds the fp32 operands while staging them -- the same roundings, twice the bytes).  With 1 the
 *                  Update also leaves the fp32 W_gifo_r^T / W_gifo_x^T out while those chains run (nothing reads them then; whoever
 *                  does -- a launch-per-step chain after a give-up or "persist" = 0, k_pack, the fp32-operand product -- gets them
 *                  transposed from the parameters first); 2 = copies AND fp32 matrices on every Update (A-B runs, tests).  The copies
 *                  are used from the first minibatch after an Update on (klstm_set_params refreshes only the fp32 matrices);
 *                  klstm_profile_query(e, "gemm_copies_launches") counts the launches that read them.  "gemm_copies_plan" = 16 nj + ks
 *                  forces tile width (32 nj columns) and K slices of those launches (0: the planner; A-B runs and tests).
   "fuse_x"  -1/0/1  x(t) W_gifo_x^T inside the step kernel (auto: NumStream <= 16) or as one batched product (:246)
 *   "vector", "fat", "small_max", "small_nt2"  kernel-family selection for A-B experiments and tests
 *   "profile" 0/1  run every kernel eagerly between its own start/stop HIP events on the
 *                  engine's stream (hipExtLaunchKernelGGL); setting the key clears the
 *                  statistics.  Used by bench.py for the roofline line. */
klstm_status klstm_set_option(klstm_engine *e, const char *key, int value);

/* With "profile" on: device time (microseconds, summed) and launch count of `kernel`
 * ("k_gates_step", "k_gates_fold", "k_dmf_step", "k_proj_step", "k_dr_step", "k_dm_step", "k_fold", "k_grads", ...)
 * over everything executed since the option was set.  Synchronises the stream. */
klstm_status klstm_profile_query(klstm_engine *e, const char *kernel, double *total_us,
                                 long *launches);

/* Test support: occupy `workgroups` compute units of `device` for about `microseconds` (one 1024-thread workgroup with
 * 96 KB of LDS per unit, spinning on the wall clock) on `hip_stream` (NULL: a private non-blocking stream).  Used by the
 * uneven-load tests of the persistent chain (MI355X_MICROARCH.md: test every hand-off under uneven load).  where_dev (or
 * NULL): 2 words per workgroup, filled with the XCC id and the HW_ID register of the compute unit it landed on. */
klstm_status klstm_debug_occupy(int device, int workgroups, int microseconds, void *hip_stream, unsigned *where_dev);

/* Test / probe support for the pipelined bf16 product of the many-stream chains (kaldi-lstm_amd/csrc/klstm_gemm16.hip):
 * C = A B^T (+ bias[n]) (+ add[m][n]) for one or two products that share their K in ONE launch, operands rounded to bf16 when staged,
 * fp32 accumulation, K split inside the launch.  mnk: M, N, K per job; ptrs: A [M x K], B [N x K], C, bias (or NULL), add (or NULL)
 * per job, device pointers; lds: lda, ldb, ldc, add_ld per job; force_nj (1 / 2 / 4: tile width 32 nj) and force_ks (1 / 2 / 4 / 8
 * K slices), 0 = the launcher's own plan; plan_out (or NULL) receives nj, ks and the number of output tiles. */
klstm_status klstm_debug_gemm_bf16_nt2(int njobs, const int *mnk, const float *const *ptrs, const int *lds, int force_nj, int force_ks,
                                       void *hip_stream, int *plan_out);
/* The same with bf16 copies of the operands in memory: copies = Ah [M x K], Bh [N x K] per job (unsigned short = bf16 bits, the RNE
 * roundings of A and B, same leading dimensions in elements, % 8 == 0, 16-byte aligned).  When every job has both, the kernel reads
 * THEM by LDS-DMA (half the bytes, no conversion pass) -- bit-identical results; NULL entries: as klstm_debug_gemm_bf16_nt2. */
klstm_status klstm_debug_gemm_bf16_nt2h(int njobs, const int *mnk, const float *const *ptrs, const int *lds,
                                        const unsigned short *const *copies, int force_nj, int force_ks, void *hip_stream, int *plan_out);

#ifdef __cplusplus
}
#endif
#endif /* KLSTM_H_ */
