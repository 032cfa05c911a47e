// include/klstm_component.hpp -- C++ host-side mirror of the reference components for the hot path,
// written against the C-ABI of klstm.h (header-only, no Kaldi dependency).
//
//   klstm_kaldi::LstmProjectedStreams  <->  google/nnet/bd-nnet-lstm-projected-streams.h:25-621
//   klstm_kaldi::LstmProjected         <->  standard/nnet/nnet-lstm-projected.h (S = 1, whole-utterance
//                                           BPTT, zero initial state, +-50 gradient clip in Update)
//
// Same method names, argument meaning and error behaviour as the reference classes:
//   InitData / ReadData / WriteData / NumParams / GetParams / Info / InfoGradient / Reset /
//   PropagateFnc / BackpropagateFnc / Update / Copy,  KALDI_ASSERT / KALDI_ERR -> std::runtime_error.
// MatrixView is layout-identical to CuMatrixBase<BaseFloat> (cu-matrix.h:479-489: data_, num_cols_,
// num_rows_, stride_), so a Kaldi build can reinterpret_cast a CuMatrixBase<float> to it -- the actual
// Kaldi-side subclass a maintainer would add is shown in INTEGRATION.md.
//
// Parameters live on the device inside the engine; this class keeps a host shadow so that model
// I/O (ReadData / WriteData / InitData) works without a GPU and the engine is created lazily at the
// first Reset / PropagateFnc.
#pragma once
#include <cmath>
#include <cstdlib>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "klstm.h"
#include "klstm_kaldi_io.hpp"

namespace klstm_kaldi {

struct MatrixView {            // == CuMatrixBase<BaseFloat> field order
  BaseFloat *data_;
  int32 num_cols_;
  int32 num_rows_;
  int32 stride_;
  MatrixView() : data_(nullptr), num_cols_(0), num_rows_(0), stride_(0) {}
  MatrixView(BaseFloat *d, int32 rows, int32 cols, int32 stride) : data_(d), num_cols_(cols), num_rows_(rows), stride_(stride) {}
  int32 NumRows() const { return num_rows_; }
  int32 NumCols() const { return num_cols_; }
  int32 Stride() const { return stride_; }
  const BaseFloat *Data() const { return data_; }
  BaseFloat *Data() { return data_; }
};

struct NnetTrainOptions {      // [UPSTREAM-unvendored nnet-trnopts.h]; only these two are read (:465, :502)
  BaseFloat learn_rate, momentum, l2_penalty, l1_penalty;
  NnetTrainOptions() : learn_rate(0.008f), momentum(0.f), l2_penalty(0.f), l1_penalty(0.f) {}
};

// [UPSTREAM-unvendored nnet-various.h] MomentStatistics: same fields, same order.
inline std::string MomentStatistics(const BaseFloat *v, size_t n) {
  double mean = 0, var = 0, skew = 0, kurt = 0, mn = 0, mx = 0;
  if (n) {
    mn = mx = v[0];
    for (size_t i = 0; i < n; i++) { mean += v[i]; if (v[i] < mn) mn = v[i]; if (v[i] > mx) mx = v[i]; }
    mean /= n;
    for (size_t i = 0; i < n; i++) { const double d = v[i] - mean; var += d * d; skew += d * d * d; kurt += d * d * d * d; }
    var /= n; skew /= n; kurt /= n;
    if (var > 0) { skew /= std::pow(var, 1.5); kurt = kurt / (var * var) - 3.0; }
  }
  std::ostringstream os;
  os << " ( min " << mn << ", max " << mx << ", mean " << mean << ", variance " << var << ", skewness " << skew
     << ", kurtosis " << kurt << " ) ";
  return os.str();
}

class LstmProjectedStreams {
 public:
  LstmProjectedStreams(int32 input_dim, int32 output_dim)          // ...streams.h:27-33
      : input_dim_(input_dim), output_dim_(output_dim), ncell_(0), nrecur_(output_dim), nstream_(0),
        device_(0), stream_(nullptr), eng_(nullptr), host_fresh_(true), corr_pending_(false), dp_comm_(nullptr) {}
  virtual ~LstmProjectedStreams() { if (eng_) klstm_destroy(eng_); }
  LstmProjectedStreams(const LstmProjectedStreams &o)               // Copy() copy-constructs every buffer (:38)
      : input_dim_(o.input_dim_), output_dim_(o.output_dim_), ncell_(o.ncell_), nrecur_(o.nrecur_),
        nstream_(o.nstream_), opts_(o.opts_), device_(o.device_), stream_(o.stream_), eng_(nullptr),
        host_fresh_(true), corr_pending_(false), options_(o.options_), dp_comm_(o.dp_comm_) {
    o.PullParams();
    params_ = o.params_;
    corr_ = o.HostCorr();
    corr_pending_ = !corr_.empty();
    if (o.eng_) {                                     // prev_nnet_state_ is copy-constructed too (:38): the carried c / r
      state_c_.resize((size_t)nstream_ * ncell_); state_r_.resize((size_t)nstream_ * nrecur_);
      Check(klstm_get_state_host(o.eng_, state_c_.data(), state_r_.data()));
    } else { state_c_ = o.state_c_; state_r_ = o.state_r_; }
  }
  LstmProjectedStreams &operator=(const LstmProjectedStreams &) = delete;
  virtual LstmProjectedStreams *Copy() const { return new LstmProjectedStreams(*this); }
  virtual const char *Marker() const { return "<LstmProjectedStreams>"; }

  int32 InputDim() const { return input_dim_; }
  int32 OutputDim() const { return output_dim_; }
  void SetTrainOptions(const NnetTrainOptions &opts) { opts_ = opts; }
  const NnetTrainOptions &GetTrainOptions() const { return opts_; }
  // where the engine lives (not part of the reference API: Kaldi has one process-wide CuDevice)
  void SetDevice(int device, void *hip_stream = nullptr) { device_ = device; stream_ = hip_stream; }

  // InitData, ...streams.h:55-99.  Proto tokens <CellDim> <NumStream> <ParamScale>.
  virtual void InitData(std::istream &is) {
    float param_scale = 0.02f;
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<CellDim>") ReadBasicType(is, false, &ncell_);
      else if (token == "<NumStream>" && HasStreams()) ReadBasicType(is, false, &nstream_);
      else if (token == "<ParamScale>") ReadBasicType(is, false, &param_scale);
      else KLSTM_ERR("Unknown token " << token << ", a typo in config?" << (HasStreams() ? " (CellDim|NumStream|ParamScale)" : " (CellDim|ParamScale)"));
      is >> std::ws;
    }
    DropEngine();
    params_.resize(NumParams());
    for (size_t i = 0; i < params_.size(); i++)       // uniform in [-scale, +scale] (:41-53)
      params_[i] = (BaseFloat)((RandUniform() - 0.5) * 2 * param_scale);
    host_fresh_ = true;
  }

  // ReadData, ...streams.h:101-131
  virtual void ReadData(std::istream &is, bool binary) {
    ExpectToken(is, binary, "<CellDim>");
    ReadBasicType(is, binary, &ncell_);
    if (HasStreams()) {
      ExpectToken(is, binary, "<NumStream>");
      ReadBasicType(is, binary, &nstream_);
    }
    DropEngine();
    const int32 C = ncell_, R = nrecur_, I = input_dim_;
    params_.assign(NumParams(), 0.f);
    size_t off = 0;
    ReadMat(is, binary, 4 * C, I, "w_gifo_x_", &off);
    ReadMat(is, binary, 4 * C, R, "w_gifo_r_", &off);
    ReadVec(is, binary, 4 * C, "bias_", &off);
    ReadVec(is, binary, C, "peephole_i_c_", &off);
    ReadVec(is, binary, C, "peephole_f_c_", &off);
    ReadVec(is, binary, C, "peephole_o_c_", &off);
    ReadMat(is, binary, R, C, "w_r_m_", &off);
    host_fresh_ = true;      // state and *_corr_ start at zero, like the Resize(kSetZero) calls (:119-130)
  }

  // WriteData, ...streams.h:133-150
  virtual void WriteData(std::ostream &os, bool binary) const {
    PullParams();
    WriteToken(os, binary, "<CellDim>");
    WriteBasicType(os, binary, ncell_);
    if (HasStreams()) {
      WriteToken(os, binary, "<NumStream>");
      WriteBasicType(os, binary, nstream_);
    }
    const int32 C = ncell_, R = nrecur_, I = input_dim_;
    const BaseFloat *p = params_.data();
    WriteMatrix(os, binary, p, 4 * C, I, I); p += (size_t)4 * C * I;
    WriteMatrix(os, binary, p, 4 * C, R, R); p += (size_t)4 * C * R;
    WriteVector(os, binary, p, 4 * C); p += 4 * C;
    WriteVector(os, binary, p, C); p += C;
    WriteVector(os, binary, p, C); p += C;
    WriteVector(os, binary, p, C); p += C;
    WriteMatrix(os, binary, p, R, C, C);
  }

  // Component::Write / Read framing [UPSTREAM-unvendored nnet-component.cc]: marker, OUTPUT dim,
  // INPUT dim, then the payload (cf. "<LstmProjectedStreams> 512 40 <CellDim> 800 <NumStream> 4  [",
  // README.md:40).
  void Write(std::ostream &os, bool binary) const {
    WriteToken(os, binary, Marker());
    WriteBasicType(os, binary, output_dim_);
    WriteBasicType(os, binary, input_dim_);
    WriteData(os, binary);
  }

  int32 NumParams() const {                              // :152-160
    return 4 * ncell_ * input_dim_ + 4 * ncell_ * nrecur_ + 4 * ncell_ + 3 * ncell_ + nrecur_ * ncell_;
  }
  void GetParams(std::vector<BaseFloat> *wei_copy) const {   // :162-189 (flat, same order)
    PullParams();
    *wei_copy = params_;
  }
  void SetParams(const std::vector<BaseFloat> &wei) {
    KLSTM_ASSERT((int32)wei.size() == NumParams());
    params_ = wei;
    host_fresh_ = true;
    if (eng_) Check(klstm_set_params_host(eng_, params_.data()));
  }

  std::string Info() const {                              // :190-199
    PullParams();
    return std::string("    ") + Stats(params_);
  }
  std::string InfoGradient() const {                      // :201-210
    return std::string("    ") + Stats(HostCorr(), "_corr_");
  }

  // Reset, ...streams.h:212-220
  void Reset(std::vector<int> &stream_reset_flag) {
    KLSTM_ASSERT(nstream_ == (int32)stream_reset_flag.size());   // :214
    EnsureEngine();
    Check(klstm_reset(eng_, stream_reset_flag.data(), (int)stream_reset_flag.size()));
  }

  // PropagateFnc, ...streams.h:222-332.  in/out hold device pointers (a CuMatrix with the GPU enabled) or host pointers.
  virtual void PropagateFnc(const MatrixView &in, MatrixView *out) {
    KLSTM_ASSERT(in.NumRows() % nstream_ == 0);                   // :225
    KLSTM_ASSERT(in.NumCols() == input_dim_ && out->NumCols() == output_dim_ && out->NumRows() == in.NumRows());
    EnsureEngine();
    // a CuMatrix holds host memory when Kaldi runs with the GPU disabled (cu-matrix.h:479-481): staged, same device path.
    // All matrices of a call live on the same side in Kaldi; a mixed call would hand a host pointer to a kernel.
    const int in_dev = in.NumRows() > 0 ? klstm_pointer_on_device(eng_, in.Data()) : 1;
    KLSTM_ASSERT(in.NumRows() == 0 || klstm_pointer_on_device(eng_, out->Data()) == in_dev);
    if (in_dev == 0)
      Check(klstm_propagate_host(eng_, in.Data(), in.NumRows(), in.Stride(), out->Data(), out->Stride()));
    else
      Check(klstm_propagate(eng_, in.Data(), in.NumRows(), in.Stride(), out->Data(), out->Stride()));
  }

  // BackpropagateFnc, ...streams.h:334-499
  virtual void BackpropagateFnc(const MatrixView &in, const MatrixView &out, const MatrixView &out_diff,
                                MatrixView *in_diff) {
    (void)out;
    EnsureEngine();
    const int in_dev = in.NumRows() > 0 ? klstm_pointer_on_device(eng_, in.Data()) : 1;
    KLSTM_ASSERT(in.NumRows() == 0 || (klstm_pointer_on_device(eng_, out_diff.Data()) == in_dev &&
                                       (!in_diff || klstm_pointer_on_device(eng_, in_diff->Data()) == in_dev)));
    if (in_dev == 0)
      Check(klstm_backpropagate_host(eng_, in.Data(), in.Stride(), out_diff.Data(), out_diff.Stride(),
                                     in_diff ? in_diff->Data() : nullptr, in_diff ? in_diff->Stride() : 0, in.NumRows(),
                                     opts_.momentum, BpttFlags()));
    else
      Check(klstm_backpropagate(eng_, in.Data(), in.Stride(), out_diff.Data(), out_diff.Stride(),
                                in_diff ? in_diff->Data() : nullptr, in_diff ? in_diff->Stride() : 0, in.NumRows(),
                                opts_.momentum, BpttFlags()));
  }
  // The caller that KNOWS Update follows immediately on the same (input, out_diff) pair says so; the engine may then leave the
  // gradient products to klstm_update and run them in one pass with the Update (KLSTM_BPTT_FUSE_UPDATE keeps the raw `in`
  // pointer until then) -- the 4-launch minibatch bench.py measures.  Inside Kaldi that is ALWAYS the case: BackpropagateFnc is
  // protected there and its one caller, Component::Backpropagate (nnet-component.h), runs Update right behind it -- the shim of
  // INTEGRATION.md sets the flag in its constructor; Backpropagate() below is that Kaldi method for users of this mirror.
  // A bare BackpropagateFnc (gradient checks, custom adapters) keeps the engine's default: the products run inside the call.
  void SetUpdateFollows(bool v) { update_follows_ = v; }
  // Kaldi nnet1's Component::Backpropagate for an updatable component: BackpropagateFnc, then Update on the same pair.
  void Backpropagate(const MatrixView &in, const MatrixView &out, const MatrixView &out_diff, MatrixView *in_diff) {
    const bool keep = update_follows_;
    update_follows_ = true;
    try { BackpropagateFnc(in, out, out_diff, in_diff); } catch (...) { update_follows_ = keep; throw; }
    update_follows_ = keep;
    Update(in, out_diff);
  }
  // "persist_verify" (klstm.h): wait for every persistent launch and answer a give-up inside the call, so that no neighbouring
  // component ever reads an invalid `out` / `in_diff`.  ON by default in this mirror -- a Kaldi trainer synchronises per
  // minibatch anyway (Xent::EvalMasked copies its sums to the host, nnet-loss.cc:110-141) and shares its GPU with whatever else
  // the machine runs; a pipelined trainer that owns the GPU turns it off for ~8 % more throughput at 4 streams.
  void SetPersistVerify(bool v) { persist_verify_ = v; if (eng_) Check(klstm_set_option(eng_, "persist_verify", v ? 1 : 0)); }
  int BpttFlags() const { return dp_comm_ ? KLSTM_BPTT_DEFER_MOMENTUM : update_follows_ ? KLSTM_BPTT_FUSE_UPDATE : KLSTM_BPTT_DEFAULT; }

  // Update, ...streams.h:501-512 (arguments unused there too)
  virtual void Update(const MatrixView &input, const MatrixView &diff) {
    (void)input; (void)diff;
    EnsureEngine();
    if (dp_comm_) {                                   // sum of the ranks' gradients, then corr = momentum*corr + sum (:465-487 per rank)
      Check(klstm_allreduce_grads(eng_, dp_comm_));
      Check(klstm_apply_momentum(eng_, opts_.momentum));
    }
    Check(klstm_update(eng_, opts_.learn_rate, ClipGrad()));
    host_fresh_ = false;
  }

  int32 CellDim() const { return ncell_; }
  int32 NumStream() const { return nstream_; }
  klstm_engine *Engine() { EnsureEngine(); return eng_; }
  // Engine tuning knobs without a reference counterpart ("fold", "bf16", "graph", ...; include/klstm.h klstm_set_option)
  void SetEngineOption(const char *key, int value) {
    EnsureEngine();
    Check(klstm_set_option(eng_, key, value));
    options_.push_back(std::make_pair(std::string(key), value));   // re-applied when Copy() / a re-created engine needs them
  }
  // Data-parallel training over utterance streams (no counterpart in the reference, which is single-GPU): with an
  // RCCL communicator set, BackpropagateFnc leaves the pure LOCAL gradient in the blob and Update first sums it over the
  // ranks (klstm_allreduce_grads: one in-place fp32 all-reduce on the engine's stream), then applies momentum and the step.
  // Every rank holds the same parameters and momentum buffers and feeds its own streams.  comm = an ncclComm_t (or the
  // handle klstm_comm_init_rank returned); nullptr switches back to single-GPU training.
  void SetDataParallel(void *rccl_comm) { dp_comm_ = rccl_comm; }

 protected:
  virtual bool HasStreams() const { return true; }       // <NumStream> is serialised (:136-137)
  virtual float ClipGrad() const { return 0.f; }

  static double RandUniform() { return (std::rand() + 1.0) / (RAND_MAX + 2.0); }   // [UPSTREAM-unvendored kaldi-math.h]

  void Check(klstm_status st) const {
    if (st != KLSTM_OK) KLSTM_ERR("klstm: " << klstm_last_error() << " (status " << (int)st << ")");
  }
  void DropEngine() {
    if (eng_) { klstm_destroy(eng_); eng_ = nullptr; }
    corr_.clear(); corr_pending_ = false;
  }
  void EnsureEngine() {
    if (eng_) return;
    KLSTM_ASSERT(ncell_ > 0 && nstream_ > 0);
    Check(klstm_create(input_dim_, ncell_, nrecur_, nstream_, device_, stream_, &eng_));
    if ((int32)params_.size() == NumParams()) Check(klstm_set_params_host(eng_, params_.data()));
    if (corr_pending_ && (int32)corr_.size() == NumParams()) Check(klstm_set_corr_host(eng_, corr_.data()));
    corr_pending_ = false;
    if (state_c_.size() == (size_t)nstream_ * ncell_ && state_r_.size() == (size_t)nstream_ * nrecur_)
      Check(klstm_set_state_host(eng_, state_c_.data(), state_r_.data()));          // carried over by Copy()
    state_c_.clear(); state_r_.clear();
    Check(klstm_set_option(eng_, "persist_verify", persist_verify_ ? 1 : 0));
    for (const auto &kv : options_) Check(klstm_set_option(eng_, kv.first.c_str(), kv.second));
  }
  void PullParams() const {
    if (eng_ && !host_fresh_) {
      params_.resize(NumParams());
      Check(klstm_get_params_host(eng_, params_.data()));
      host_fresh_ = true;
    }
  }
  std::vector<BaseFloat> HostCorr() const {
    if (!eng_) return corr_pending_ ? corr_ : std::vector<BaseFloat>(ncell_ > 0 ? NumParams() : 0, 0.f);
    std::vector<BaseFloat> c(NumParams());
    Check(klstm_get_corr_host(eng_, c.data()));
    return c;
  }
  std::string Stats(const std::vector<BaseFloat> &b, const char *suffix = "_") const {
    const int32 C = ncell_, R = nrecur_, I = input_dim_;
    const size_t len[7] = {(size_t)4 * C * I, (size_t)4 * C * R, (size_t)4 * C, (size_t)C, (size_t)C, (size_t)C, (size_t)R * C};
    const char *names[7] = {"w_gifo_x", "w_gifo_r", "bias", "peephole_i_c", "peephole_f_c", "peephole_o_c", "w_r_m"};
    std::string s;
    size_t off = 0;
    for (int i = 0; i < 7; i++) {
      s += std::string("\n  ") + names[i] + suffix + "  " + (b.empty() ? std::string("( empty )") : MomentStatistics(b.data() + off, len[i]));
      off += len[i];
    }
    return s;
  }
  void ReadMat(std::istream &is, bool binary, int32 rows, int32 cols, const char *name, size_t *off) {
    std::vector<BaseFloat> d;
    int32 r = 0, c = 0;
    ReadMatrix(is, binary, &d, &r, &c);
    if (r != rows || c != cols) KLSTM_ERR("ReadData: " << name << " is " << r << " x " << c << ", expected " << rows << " x " << cols);
    std::copy(d.begin(), d.end(), params_.begin() + *off);
    *off += d.size();
  }
  void ReadVec(std::istream &is, bool binary, int32 dim, const char *name, size_t *off) {
    std::vector<BaseFloat> d;
    ReadVector(is, binary, &d);
    if ((int32)d.size() != dim) KLSTM_ERR("ReadData: " << name << " has dim " << d.size() << ", expected " << dim);
    std::copy(d.begin(), d.end(), params_.begin() + *off);
    *off += d.size();
  }

  int32 input_dim_, output_dim_;
  int32 ncell_, nrecur_, nstream_;
  NnetTrainOptions opts_;
  int device_;
  void *stream_;
  klstm_engine *eng_;
  mutable std::vector<BaseFloat> params_;   // host shadow, GetParams order
  mutable bool host_fresh_;                 // params_ == device parameters
  bool update_follows_ = false;             // set by the caller that runs Update right behind BackpropagateFnc
  bool persist_verify_ = true;              // SetPersistVerify
  std::vector<BaseFloat> corr_;             // only to carry *_corr_ across Copy()
  bool corr_pending_;
  std::vector<BaseFloat> state_c_, state_r_;        // only to carry prev_nnet_state_ (c, r columns) across Copy()
  std::vector<std::pair<std::string, int> > options_;   // engine options set through SetEngineOption
  void *dp_comm_;                           // RCCL communicator of data-parallel training, or nullptr
};

// standard/nnet/nnet-lstm-projected.h: one utterance per call, no state bridge (:228-231, :314-315 are
// commented out there), <NumStream> absent from the model file, Update clips every *_corr_ to +-50 first
// (:480-493).
class LstmProjected : public LstmProjectedStreams {
 public:
  LstmProjected(int32 input_dim, int32 output_dim) : LstmProjectedStreams(input_dim, output_dim) { nstream_ = 1; }
  LstmProjected(const LstmProjected &o) : LstmProjectedStreams(o) {}
  LstmProjectedStreams *Copy() const override { return new LstmProjected(*this); }
  const char *Marker() const override { return "<LstmProjected>"; }
  void PropagateFnc(const MatrixView &in, MatrixView *out) override {
    std::vector<int> one(1, 1);
    Reset(one);                      // history is never bridged between sentences
    LstmProjectedStreams::PropagateFnc(in, out);
  }
 protected:
  bool HasStreams() const override { return false; }
  float ClipGrad() const override { return 50.f; }
};

// standard/nnet/nnet-time-shift.h: out[t] = in[clamp(t + shift)], no backprop (:53-56).  Gives the targets
// delay at decode time (README.md:46-49).
class TimeShift {
 public:
  TimeShift(int32 dim_in, int32 dim_out) : input_dim_(dim_in), output_dim_(dim_out), shift_(0), stream_(nullptr) {}
  const char *Marker() const { return "<TimeShift>"; }
  int32 InputDim() const { return input_dim_; }
  int32 OutputDim() const { return output_dim_; }
  int32 Shift() const { return shift_; }
  void SetStream(void *hip_stream) { stream_ = hip_stream; }
  void InitData(std::istream &is) {                           // :29-37
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<Shift>") ReadBasicType(is, false, &shift_);
      else KLSTM_ERR("Unknown token " << token << ", a typo in config?" << " (Shift)");
      is >> std::ws;
    }
  }
  void ReadData(std::istream &is, bool binary) { ExpectToken(is, binary, "<Shift>"); ReadBasicType(is, binary, &shift_); }   // :38-41
  void WriteData(std::ostream &os, bool binary) const {       // :42-46 (the reference appends a newline)
    WriteToken(os, binary, "<Shift>");
    WriteBasicType(os, binary, shift_);
    os << "\n";
  }
  void Write(std::ostream &os, bool binary) const {
    WriteToken(os, binary, Marker()); WriteBasicType(os, binary, output_dim_); WriteBasicType(os, binary, input_dim_);
    WriteData(os, binary);
  }
  void PropagateFnc(const MatrixView &in, MatrixView *out) {   // :42-51
    KLSTM_ASSERT(in.NumRows() == out->NumRows() && in.NumCols() == out->NumCols());
    const klstm_status st = klstm_time_shift(in.Data(), in.NumRows(), in.NumCols(), in.Stride(), out->Data(), out->Stride(), shift_, stream_);
    if (st != KLSTM_OK) KLSTM_ERR("klstm: " << klstm_last_error());
  }
  void BackpropagateFnc(const MatrixView &, const MatrixView &, const MatrixView &, MatrixView *) {}   // :53-56: meaningless
 private:
  int32 input_dim_, output_dim_, shift_;
  void *stream_;
};

// standard/nnet/nnet-transmit-component.h: identity forward and backward; exists only so that the LSTM is
// not component 0 (README.md:49).
class TransmitComponent {
 public:
  TransmitComponent(int32 dim_in, int32 dim_out) : input_dim_(dim_in), output_dim_(dim_out), stream_(nullptr) {}
  const char *Marker() const { return "<Transmit>"; }
  int32 InputDim() const { return input_dim_; }
  int32 OutputDim() const { return output_dim_; }
  void SetStream(void *hip_stream) { stream_ = hip_stream; }
  void Write(std::ostream &os, bool binary) const {
    WriteToken(os, binary, Marker()); WriteBasicType(os, binary, output_dim_); WriteBasicType(os, binary, input_dim_);
  }
  void PropagateFnc(const MatrixView &in, MatrixView *out) { Copy(in, out); }                                      // :26-28
  void BackpropagateFnc(const MatrixView &, const MatrixView &, const MatrixView &out_diff, MatrixView *in_diff) { // :30-33
    if (in_diff) Copy(out_diff, in_diff);
  }
 private:
  void Copy(const MatrixView &a, MatrixView *b) {
    KLSTM_ASSERT(a.NumRows() == b->NumRows() && a.NumCols() == b->NumCols());
    const klstm_status st = klstm_time_shift(a.Data(), a.NumRows(), a.NumCols(), a.Stride(), b->Data(), b->Stride(), 0, stream_);
    if (st != KLSTM_OK) KLSTM_ERR("klstm: " << klstm_last_error());
  }
  int32 input_dim_, output_dim_;
  void *stream_;
};

// Component::Read framing: "<Marker> out_dim in_dim" then ReadData.  Returns nullptr at "</Nnet>".
inline LstmProjectedStreams *ReadLstmComponent(std::istream &is, bool binary) {
  std::string token;
  ReadToken(is, binary, &token);
  if (token == "<Nnet>") ReadToken(is, binary, &token);
  if (token == "</Nnet>") return nullptr;
  int32 dim_out, dim_in;
  ReadBasicType(is, binary, &dim_out);
  ReadBasicType(is, binary, &dim_in);
  std::unique_ptr<LstmProjectedStreams> c;
  if (token == "<LstmProjectedStreams>") c.reset(new LstmProjectedStreams(dim_in, dim_out));
  else if (token == "<LstmProjected>") c.reset(new LstmProjected(dim_in, dim_out));
  else KLSTM_ERR("Unknown component marker " << token << " (this reader covers <LstmProjectedStreams> and <LstmProjected>)");
  c->ReadData(is, binary);
  return c.release();
}

}  // namespace klstm_kaldi
