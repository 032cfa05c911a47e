// include/klstm_nnet.hpp -- minimal nnet1 container around the hot path: the `Nnet` the reference trainer
// drives (google/nnet/nnet-nnet.h:36-150, only the declaration is vendored), the components that appear
// in the reference topologies (google/nnet.proto, standard/nnet.proto, README.md:24-45), the masked
// cross-entropy (google/nnet/nnet-loss.cc:76-164, 293-307) and a workalike of the training loop of
// google/nnetbin/bd-nnet-train-lstm-streams.cc (:143-304) on in-memory utterances.
// Header-only C++ over the C-ABI (klstm.h); no HIP or Kaldi headers needed.
#pragma once
#include <chrono>
#include <memory>

#include "klstm_component.hpp"
#include "klstm_trainer.hpp"

namespace klstm_kaldi {

inline void KCheck(klstm_status st) { if (st != KLSTM_OK) KLSTM_ERR("klstm: " << klstm_last_error() << " (status " << (int)st << ")"); }

// CuMatrix stand-in: owning, pitched device matrix (cu-matrix.cc:51-84: rows are pitched; here the stride is the
// column count rounded up to 64 floats = 256 B).
class DeviceMatrix {
 public:
  DeviceMatrix() : data_(nullptr), rows_(0), cols_(0), stride_(0), cap_(0) {}
  ~DeviceMatrix() { if (data_) klstm_free(data_); }
  DeviceMatrix(const DeviceMatrix &) = delete;
  DeviceMatrix &operator=(const DeviceMatrix &) = delete;
  void Resize(int32 rows, int32 cols, bool set_zero = true) {     // no realloc when the shape is unchanged (cu-matrix.cc:56-59)
    const int32 stride = (cols + 63) / 64 * 64;
    const size_t need = (size_t)rows * stride;
    if (need > cap_) {
      if (data_) KCheck(klstm_free(data_));
      void *p = nullptr;
      KCheck(klstm_malloc(&p, need * sizeof(BaseFloat)));
      data_ = (BaseFloat *)p; cap_ = need;
    }
    rows_ = rows; cols_ = cols; stride_ = stride;
    if (set_zero && need) KCheck(klstm_memset_zero(data_, need * sizeof(BaseFloat), nullptr));
  }
  int32 NumRows() const { return rows_; }
  int32 NumCols() const { return cols_; }
  int32 Stride() const { return stride_; }
  MatrixView View() const { return MatrixView(data_, rows_, cols_, stride_); }
  void CopyFromHost(const BaseFloat *src, int32 rows, int32 cols) {          // CuMatrix(const Matrix&), cu-matrix.cc:287-311
    Resize(rows, cols, false);
    std::vector<BaseFloat> tmp((size_t)rows * stride_, 0.f);
    for (int32 r = 0; r < rows; r++) std::memcpy(&tmp[(size_t)r * stride_], src + (size_t)r * cols, cols * sizeof(BaseFloat));
    if (!tmp.empty()) KCheck(klstm_memcpy_h2d(data_, tmp.data(), tmp.size() * sizeof(BaseFloat), nullptr));
  }
  void CopyToHost(std::vector<BaseFloat> *dst) const {
    std::vector<BaseFloat> tmp((size_t)rows_ * stride_);
    if (!tmp.empty()) KCheck(klstm_memcpy_d2h(tmp.data(), data_, tmp.size() * sizeof(BaseFloat), nullptr));
    dst->resize((size_t)rows_ * cols_);
    for (int32 r = 0; r < rows_; r++) std::memcpy(&(*dst)[(size_t)r * cols_], &tmp[(size_t)r * stride_], cols_ * sizeof(BaseFloat));
  }
 private:
  BaseFloat *data_;
  int32 rows_, cols_, stride_;
  size_t cap_;
};

// Polymorphic view of one nnet1 component (Component / UpdatableComponent, [UPSTREAM-unvendored] nnet-component.h).
class Layer {
 public:
  virtual ~Layer() {}
  virtual const char *Marker() const = 0;
  virtual int32 InputDim() const = 0;
  virtual int32 OutputDim() const = 0;
  virtual bool IsUpdatable() const { return false; }
  virtual void ReadData(std::istream &, bool) {}
  virtual void WriteData(std::ostream &, bool) const {}
  virtual void PropagateFnc(const MatrixView &in, MatrixView *out) = 0;
  virtual void BackpropagateFnc(const MatrixView &in, const MatrixView &out, const MatrixView &out_diff, MatrixView *in_diff) = 0;
  virtual void Update(const MatrixView &, const MatrixView &) {}
  virtual void SetUpdateFollows(bool) {}                     // Nnet::Backpropagate: Update comes right behind BackpropagateFnc
  virtual void SetTrainOptions(const NnetTrainOptions &) {}
  virtual void Reset(std::vector<int> &) {}                  // the overlay adds Reset to every Component (nnet-nnet.h:133-137)
  virtual int32 NumParams() const { return 0; }
  virtual void GetParams(std::vector<BaseFloat> *p) const { p->clear(); }
  void Write(std::ostream &os, bool binary) const {
    WriteToken(os, binary, Marker());
    WriteBasicType(os, binary, OutputDim());
    WriteBasicType(os, binary, InputDim());
    const std::streampos before = os.tellp();
    WriteData(os, binary);
    if (!binary && os.tellp() == before) os << "\n";
  }
};

class LstmLayer : public Layer {            // LstmProjectedStreams / LstmProjected
 public:
  explicit LstmLayer(LstmProjectedStreams *c) : c_(c) {}
  const char *Marker() const override { return c_->Marker(); }
  int32 InputDim() const override { return c_->InputDim(); }
  int32 OutputDim() const override { return c_->OutputDim(); }
  bool IsUpdatable() const override { return true; }
  void ReadData(std::istream &is, bool b) override { c_->ReadData(is, b); }
  void WriteData(std::ostream &os, bool b) const override { c_->WriteData(os, b); }
  void PropagateFnc(const MatrixView &in, MatrixView *out) override { c_->PropagateFnc(in, out); }
  void BackpropagateFnc(const MatrixView &in, const MatrixView &out, const MatrixView &od, MatrixView *id) override { c_->BackpropagateFnc(in, out, od, id); }
  void Update(const MatrixView &a, const MatrixView &b) override { c_->Update(a, b); }
  void SetUpdateFollows(bool v) override { c_->SetUpdateFollows(v); }
  void SetTrainOptions(const NnetTrainOptions &o) override { c_->SetTrainOptions(o); }
  void Reset(std::vector<int> &f) override { if (std::string(c_->Marker()) == "<LstmProjectedStreams>") c_->Reset(f); }
  int32 NumParams() const override { return c_->NumParams(); }
  void GetParams(std::vector<BaseFloat> *p) const override { c_->GetParams(p); }
  LstmProjectedStreams *Impl() { return c_.get(); }
 private:
  std::unique_ptr<LstmProjectedStreams> c_;
};

class TimeShiftLayer : public Layer {
 public:
  TimeShiftLayer(int32 i, int32 o) : c_(i, o) {}
  const char *Marker() const override { return c_.Marker(); }
  int32 InputDim() const override { return c_.InputDim(); }
  int32 OutputDim() const override { return c_.OutputDim(); }
  void ReadData(std::istream &is, bool b) override { c_.ReadData(is, b); }
  void WriteData(std::ostream &os, bool b) const override { c_.WriteData(os, b); }
  void PropagateFnc(const MatrixView &in, MatrixView *out) override { c_.PropagateFnc(in, out); }
  void BackpropagateFnc(const MatrixView &a, const MatrixView &b, const MatrixView &c, MatrixView *d) override { c_.BackpropagateFnc(a, b, c, d); }
 private:
  TimeShift c_;
};

class TransmitLayer : public Layer {
 public:
  TransmitLayer(int32 i, int32 o) : c_(i, o) {}
  const char *Marker() const override { return c_.Marker(); }
  int32 InputDim() const override { return c_.InputDim(); }
  int32 OutputDim() const override { return c_.OutputDim(); }
  void PropagateFnc(const MatrixView &in, MatrixView *out) override { c_.PropagateFnc(in, out); }
  void BackpropagateFnc(const MatrixView &a, const MatrixView &b, const MatrixView &c, MatrixView *d) override { c_.BackpropagateFnc(a, b, c, d); }
 private:
  TransmitComponent c_;
};

// [UPSTREAM-unvendored nnet-affine-transform.h] AffineTransform: out = in * linearity^T + bias; model line
// "<AffineTransform> 16624 512 <LearnRateCoef> 1 <BiasLearnRateCoef> 1 <MaxNorm> 0  [ ..." (README.md:27).
class AffineLayer : public Layer {
 public:
  AffineLayer(int32 in, int32 out) : in_(in), out_(out), lr_coef_(1.f), bias_lr_coef_(1.f), max_norm_(0.f),
                                     W_(nullptr), b_(nullptr), Wc_(nullptr), bc_(nullptr) {}
  ~AffineLayer() override { klstm_free(W_); klstm_free(b_); klstm_free(Wc_); klstm_free(bc_); }
  const char *Marker() const override { return "<AffineTransform>"; }
  int32 InputDim() const override { return in_; }
  int32 OutputDim() const override { return out_; }
  bool IsUpdatable() const override { return true; }
  void ReadData(std::istream &is, bool binary) override {
    while ('<' == Peek(is, binary)) {                     // optional learning-rate tokens
      std::string tok;
      ReadToken(is, binary, &tok);
      if (tok == "<LearnRateCoef>") ReadBasicType(is, binary, &lr_coef_);
      else if (tok == "<BiasLearnRateCoef>") ReadBasicType(is, binary, &bias_lr_coef_);
      else if (tok == "<MaxNorm>") ReadBasicType(is, binary, &max_norm_);
      else KLSTM_ERR("Unknown token " << tok);
    }
    std::vector<BaseFloat> w, b;
    int32 r, c;
    ReadMatrix(is, binary, &w, &r, &c);
    ReadVector(is, binary, &b);
    if (r != out_ || c != in_ || (int32)b.size() != out_) KLSTM_ERR("AffineTransform: dimension mismatch");
    SetParams(w, b);
  }
  void WriteData(std::ostream &os, bool binary) const override {
    std::vector<BaseFloat> w, b;
    HostParams(&w, &b);
    WriteToken(os, binary, "<LearnRateCoef>"); WriteBasicType(os, binary, lr_coef_);
    WriteToken(os, binary, "<BiasLearnRateCoef>"); WriteBasicType(os, binary, bias_lr_coef_);
    WriteToken(os, binary, "<MaxNorm>"); WriteBasicType(os, binary, max_norm_);
    WriteMatrix(os, binary, w.data(), out_, in_, in_);
    WriteVector(os, binary, b.data(), out_);
  }
  // parameters keep a host shadow so model files can be read / converted without a GPU; the device copy is
  // created at the first PropagateFnc
  void SetParams(const std::vector<BaseFloat> &w, const std::vector<BaseFloat> &b) {
    KLSTM_ASSERT((int32)w.size() == out_ * in_ && (int32)b.size() == out_);
    hw_ = w; hb_ = b; host_fresh_ = true;
    if (W_) Upload();
  }
  void HostParams(std::vector<BaseFloat> *w, std::vector<BaseFloat> *b) const {
    if (W_ && !host_fresh_) {
      hw_.resize((size_t)out_ * in_); hb_.resize(out_);
      KCheck(klstm_memcpy_d2h(hw_.data(), W_, hw_.size() * sizeof(BaseFloat), nullptr));
      KCheck(klstm_memcpy_d2h(hb_.data(), b_, hb_.size() * sizeof(BaseFloat), nullptr));
      host_fresh_ = true;
    }
    *w = hw_; *b = hb_;
  }
  int32 NumParams() const override { return out_ * in_ + out_; }
  void GetParams(std::vector<BaseFloat> *p) const override {
    std::vector<BaseFloat> w, b; HostParams(&w, &b); *p = w; p->insert(p->end(), b.begin(), b.end());
  }
  void PropagateFnc(const MatrixView &in, MatrixView *out) override {
    Alloc();
    KCheck(klstm_affine_propagate(in.Data(), in.NumRows(), in_, in.Stride(), W_, b_, out->Data(), out_, out->Stride(), nullptr));
  }
  void BackpropagateFnc(const MatrixView &, const MatrixView &, const MatrixView &od, MatrixView *id) override {
    if (id) KCheck(klstm_affine_backpropagate(od.Data(), od.NumRows(), out_, od.Stride(), W_, in_, id->Data(), id->Stride(), nullptr));
  }
  void Update(const MatrixView &input, const MatrixView &diff) override {
    if (opts_.l2_penalty != 0.f || opts_.l1_penalty != 0.f) KLSTM_ERR("AffineTransform: l1/l2 penalties are not implemented");
    KCheck(klstm_affine_update(input.Data(), input.Stride(), diff.Data(), diff.Stride(), input.NumRows(), in_, out_, W_, b_, Wc_, bc_,
                               opts_.learn_rate * lr_coef_, opts_.learn_rate * bias_lr_coef_, opts_.momentum, nullptr));
    host_fresh_ = false;
  }
  void SetTrainOptions(const NnetTrainOptions &o) override { opts_ = o; }
 private:
  void Alloc() {
    if (W_) return;
    void *p;
    KCheck(klstm_malloc(&p, (size_t)out_ * in_ * 4)); W_ = (BaseFloat *)p;
    KCheck(klstm_malloc(&p, (size_t)out_ * 4)); b_ = (BaseFloat *)p;
    KCheck(klstm_malloc(&p, (size_t)out_ * in_ * 4)); Wc_ = (BaseFloat *)p;
    KCheck(klstm_malloc(&p, (size_t)out_ * 4)); bc_ = (BaseFloat *)p;
    KCheck(klstm_memset_zero(Wc_, (size_t)out_ * in_ * 4, nullptr));
    KCheck(klstm_memset_zero(bc_, (size_t)out_ * 4, nullptr));
    Upload();
  }
  void Upload() {
    KLSTM_ASSERT((int32)hw_.size() == out_ * in_ && (int32)hb_.size() == out_);
    KCheck(klstm_memcpy_h2d(W_, hw_.data(), hw_.size() * sizeof(BaseFloat), nullptr));
    KCheck(klstm_memcpy_h2d(b_, hb_.data(), hb_.size() * sizeof(BaseFloat), nullptr));
  }
  int32 in_, out_;
  BaseFloat lr_coef_, bias_lr_coef_, max_norm_;
  NnetTrainOptions opts_;
  BaseFloat *W_, *b_, *Wc_, *bc_;
  mutable std::vector<BaseFloat> hw_, hb_;
  mutable bool host_fresh_ = true;
};

// [UPSTREAM-unvendored nnet-activation.h] Softmax: row softmax forward; backward passes the diff through, because
// Xent's diff (y - t) is already the derivative w.r.t. the softmax input.
class SoftmaxLayer : public Layer {
 public:
  SoftmaxLayer(int32 i, int32 o) : in_(i), out_(o) {}
  const char *Marker() const override { return "<Softmax>"; }
  int32 InputDim() const override { return in_; }
  int32 OutputDim() const override { return out_; }
  void PropagateFnc(const MatrixView &in, MatrixView *out) override {
    KCheck(klstm_softmax(in.Data(), in.NumRows(), in.NumCols(), in.Stride(), out->Data(), out->Stride(), nullptr));
  }
  void BackpropagateFnc(const MatrixView &, const MatrixView &, const MatrixView &od, MatrixView *id) override {
    if (id) KCheck(klstm_time_shift(od.Data(), od.NumRows(), od.NumCols(), od.Stride(), id->Data(), id->Stride(), 0, nullptr));
  }
 private:
  int32 in_, out_;
};

class Nnet {                                  // google/nnet/nnet-nnet.h:36-150
 public:
  Nnet() {}
  int32 NumComponents() const { return (int32)layers_.size(); }
  Layer &GetComponent(int32 i) { return *layers_[i]; }
  int32 InputDim() const { KLSTM_ASSERT(!layers_.empty()); return layers_.front()->InputDim(); }
  int32 OutputDim() const { KLSTM_ASSERT(!layers_.empty()); return layers_.back()->OutputDim(); }
  void AppendComponent(Layer *l) {
    if (!layers_.empty() && layers_.back()->OutputDim() != l->InputDim()) KLSTM_ERR("Nnet: dimension mismatch between components");
    layers_.emplace_back(l);
  }

  void Read(std::istream &is, bool binary) {      // "<Nnet>" components "</Nnet>"
    layers_.clear();
    std::string token;
    ReadToken(is, binary, &token);
    if (token != "<Nnet>") KLSTM_ERR("Expected <Nnet>, got " << token);
    while (true) {
      ReadToken(is, binary, &token);
      if (token == "</Nnet>") break;
      int32 dim_out, dim_in;
      ReadBasicType(is, binary, &dim_out);
      ReadBasicType(is, binary, &dim_in);
      Layer *l = nullptr;
      if (token == "<LstmProjectedStreams>") l = new LstmLayer(new LstmProjectedStreams(dim_in, dim_out));
      else if (token == "<LstmProjected>") l = new LstmLayer(new LstmProjected(dim_in, dim_out));
      else if (token == "<TimeShift>") l = new TimeShiftLayer(dim_in, dim_out);
      else if (token == "<Transmit>") l = new TransmitLayer(dim_in, dim_out);
      else if (token == "<AffineTransform>") l = new AffineLayer(dim_in, dim_out);
      else if (token == "<Softmax>") l = new SoftmaxLayer(dim_in, dim_out);
      else KLSTM_ERR("Unknown component marker " << token);
      std::unique_ptr<Layer> guard(l);
      l->ReadData(is, binary);
      AppendComponent(guard.release());
    }
  }
  void Read(const std::string &file) {
    std::ifstream f(file, std::ios::binary);
    if (!f) KLSTM_ERR("cannot open " << file);
    Read(f, InitKaldiInputStream(f));
  }
  void Write(std::ostream &os, bool binary) const {
    WriteToken(os, binary, "<Nnet>");
    if (!binary) os << "\n";
    for (const auto &l : layers_) l->Write(os, binary);
    WriteToken(os, binary, "</Nnet>");
    if (!binary) os << "\n";
  }
  void Write(const std::string &file, bool binary) const {
    std::ofstream f(file, std::ios::binary);
    InitKaldiOutputStream(f, binary);
    Write(f, binary);
  }

  void SetTrainOptions(const NnetTrainOptions &o) { opts_ = o; for (auto &l : layers_) l->SetTrainOptions(o); }
  void Reset(std::vector<int> &stream_reset_flag) {          // nnet-nnet.h:132-138: fan out to EVERY component
    for (auto &l : layers_) l->Reset(stream_reset_flag);
  }

  // Nnet::Propagate [UPSTREAM]: each component's output buffer is (re)sized, then PropagateFnc.
  void Propagate(const MatrixView &in, DeviceMatrix *out) {
    const int32 n = NumComponents();
    if ((int32)prop_.size() != n + 1) { prop_.clear(); for (int32 i = 0; i <= n; i++) prop_.emplace_back(new DeviceMatrix()); }
    in0_ = in;
    for (int32 i = 0; i < n; i++) {
      prop_[i + 1]->Resize(in.NumRows(), layers_[i]->OutputDim(), false);
      MatrixView o = prop_[i + 1]->View();
      layers_[i]->PropagateFnc(i == 0 ? in : prop_[i]->View(), &o);
    }
    out->Resize(in.NumRows(), OutputDim(), false);
    MatrixView ov = out->View();
    KCheck(klstm_time_shift(prop_[n]->View().Data(), in.NumRows(), OutputDim(), prop_[n]->Stride(), ov.Data(), ov.Stride(), 0, nullptr));
  }
  void Feedforward(const MatrixView &in, DeviceMatrix *out) { Propagate(in, out); }

  // Nnet::Backpropagate(out_diff, NULL) [UPSTREAM]: components last -> first: Backpropagate, then Update if updatable.
  // The first component gets no in_diff (the stated reason for the dummy <Transmit>, README.md:49).
  void Backpropagate(const MatrixView &out_diff, MatrixView *in_diff) {
    const int32 n = NumComponents();
    KLSTM_ASSERT((int32)prop_.size() == n + 1);
    if ((int32)bprop_.size() != n + 1) { bprop_.clear(); for (int32 i = 0; i <= n; i++) bprop_.emplace_back(new DeviceMatrix()); }
    for (int32 i = n - 1; i >= 0; i--) {
      const MatrixView in = i == 0 ? in0_ : prop_[i]->View();
      const MatrixView out = prop_[i + 1]->View();
      const MatrixView od = i == n - 1 ? out_diff : bprop_[i + 1]->View();
      MatrixView idv, *idp = nullptr;
      if (i > 0) { bprop_[i]->Resize(in.NumRows(), layers_[i]->InputDim(), false); idv = bprop_[i]->View(); idp = &idv; }
      else if (in_diff) { idp = in_diff; }
      const bool upd = layers_[i]->IsUpdatable();
      layers_[i]->SetUpdateFollows(upd);
      layers_[i]->BackpropagateFnc(in, out, od, idp);
      if (upd) layers_[i]->Update(in, od);
      layers_[i]->SetUpdateFollows(false);
    }
  }
 private:
  std::vector<std::unique_ptr<Layer> > layers_;
  std::vector<std::unique_ptr<DeviceMatrix> > prop_, bprop_;
  MatrixView in0_;
  NnetTrainOptions opts_;
};

// Kaldi's Posterior (hmm/posterior.h [UPSTREAM-unvendored]; used as such in nnet-loss.cc:78): per frame a list of (pdf-id, weight)
typedef std::vector<std::vector<std::pair<int32, BaseFloat> > > Posterior;

// Xent with the overlay's EvalMasked (google/nnet/nnet-loss.h:33-80, nnet-loss.cc:76-164, Report :293-307).
class Xent {
 public:
  Xent() : frames_(0), correct_(0), loss_(0), entropy_(0), tgt_(nullptr), mask_(nullptr), rx_(nullptr), rc_(nullptr), cap_(0),
           poff_(nullptr), ppdf_(nullptr), pw_(nullptr), re_(nullptr), pcap_(0), ecap_(0) {}
  ~Xent() { klstm_free(tgt_); klstm_free(mask_); klstm_free(rx_); klstm_free(rc_); klstm_free(poff_); klstm_free(ppdf_); klstm_free(pw_); klstm_free(re_); }
  // EvalMasked with the reference's signature (nnet-loss.cc:76-79): general posteriors.  The (pdf, weight) lists go to the
  // device as CSR arrays (a few KB) instead of the reference's dense num_frames x num_pdf host matrix (:85-97).
  void EvalMasked(const std::vector<BaseFloat> &frame_mask, const DeviceMatrix &net_out, const Posterior &post, DeviceMatrix *diff) {
    const int32 n = net_out.NumRows(), d = net_out.NumCols();
    KLSTM_ASSERT(n == (int32)post.size() && n == (int32)frame_mask.size());              // :82
    std::vector<int32> off(1, 0), pdf;
    std::vector<BaseFloat> w;
    for (int32 t = 0; t < n; t++) {
      for (size_t i = 0; i < post[t].size(); i++) {
        const int32 id = post[t][i].first;
        if (id >= d || id < 0) KLSTM_ERR("Posterior pdf-id out of NN-output dimension, please check number of pdfs by 'hmm-info'." << " nn-outputs : " << d << ", posterior pdf-id : " << id);   // :89-92
        pdf.push_back(id); w.push_back(post[t][i].second);
      }
      off.push_back((int32)pdf.size());
    }
    void *p;
    if ((size_t)n > pcap_) {
      klstm_free(poff_); klstm_free(mask_); klstm_free(rx_); klstm_free(rc_); klstm_free(re_); klstm_free(tgt_);
      KCheck(klstm_malloc(&p, (size_t)(n + 1) * 4)); poff_ = (int32 *)p;
      KCheck(klstm_malloc(&p, (size_t)n * 4)); mask_ = (BaseFloat *)p;
      KCheck(klstm_malloc(&p, (size_t)n * 4)); rx_ = (BaseFloat *)p;
      KCheck(klstm_malloc(&p, (size_t)n * 4)); rc_ = (BaseFloat *)p;
      KCheck(klstm_malloc(&p, (size_t)n * 4)); re_ = (BaseFloat *)p;
      KCheck(klstm_malloc(&p, (size_t)n * 4)); tgt_ = (int32 *)p;
      pcap_ = n; cap_ = n;
    }
    if (pdf.size() + 1 > ecap_) {
      klstm_free(ppdf_); klstm_free(pw_);
      ecap_ = pdf.size() + 1;
      KCheck(klstm_malloc(&p, ecap_ * 4)); ppdf_ = (int32 *)p;
      KCheck(klstm_malloc(&p, ecap_ * 4)); pw_ = (BaseFloat *)p;
    }
    KCheck(klstm_memcpy_h2d(poff_, off.data(), off.size() * 4, nullptr));
    if (!pdf.empty()) { KCheck(klstm_memcpy_h2d(ppdf_, pdf.data(), pdf.size() * 4, nullptr)); KCheck(klstm_memcpy_h2d(pw_, w.data(), w.size() * 4, nullptr)); }
    KCheck(klstm_memcpy_h2d(mask_, frame_mask.data(), (size_t)n * 4, nullptr));
    diff->Resize(n, d, false);                                                            // :103
    MatrixView y = net_out.View(), dv = diff->View();
    KCheck(klstm_xent_eval_masked_post(y.Data(), n, d, y.Stride(), poff_, ppdf_, pw_, mask_, dv.Data(), dv.Stride(), rx_, re_, rc_, nullptr));
    std::vector<BaseFloat> rx(n), rc(n), re(n);
    KCheck(klstm_memcpy_d2h(rx.data(), rx_, (size_t)n * 4, nullptr));
    KCheck(klstm_memcpy_d2h(rc.data(), rc_, (size_t)n * 4, nullptr));
    KCheck(klstm_memcpy_d2h(re.data(), re_, (size_t)n * 4, nullptr));
    double xe = 0, ent = 0; int32 correct = 0, valid = 0;
    for (int32 i = 0; i < n; i++) { xe += rx[i]; ent += re[i]; correct += (rc[i] == 1.f); valid += (frame_mask[i] == 1.f); }
    loss_ += xe; entropy_ += ent; correct_ += correct; frames_ += valid;                  // :138-142
  }
  // frame_mask: 1 valid / 0 padded per row; target: pdf-id per row (one-hot posterior)
  void EvalMasked(const std::vector<BaseFloat> &frame_mask, const DeviceMatrix &net_out, const std::vector<int32> &target,
                  DeviceMatrix *diff) {
    const int32 n = net_out.NumRows(), d = net_out.NumCols();
    KLSTM_ASSERT(n == (int32)target.size() && n == (int32)frame_mask.size());          // :82
    for (int32 t : target) if (t >= d || t < 0) KLSTM_ERR("Posterior pdf-id out of NN-output dimension, please check number of pdfs by 'hmm-info'." << " nn-outputs : " << d << ", posterior pdf-id : " << t);   // :89-92
    if ((size_t)n > cap_) {
      klstm_free(tgt_); klstm_free(mask_); klstm_free(rx_); klstm_free(rc_);
      void *p;
      KCheck(klstm_malloc(&p, (size_t)n * 4)); tgt_ = (int32 *)p;
      KCheck(klstm_malloc(&p, (size_t)n * 4)); mask_ = (BaseFloat *)p;
      KCheck(klstm_malloc(&p, (size_t)n * 4)); rx_ = (BaseFloat *)p;
      KCheck(klstm_malloc(&p, (size_t)n * 4)); rc_ = (BaseFloat *)p;
      cap_ = n;
    }
    KCheck(klstm_memcpy_h2d(tgt_, target.data(), (size_t)n * 4, nullptr));
    KCheck(klstm_memcpy_h2d(mask_, frame_mask.data(), (size_t)n * 4, nullptr));
    diff->Resize(n, d, false);                                                            // :103
    MatrixView y = net_out.View(), dv = diff->View();
    KCheck(klstm_xent_eval_masked(y.Data(), n, d, y.Stride(), tgt_, mask_, dv.Data(), dv.Stride(), rx_, rc_, nullptr));
    std::vector<BaseFloat> rx(n), rc(n);
    KCheck(klstm_memcpy_d2h(rx.data(), rx_, (size_t)n * 4, nullptr));
    KCheck(klstm_memcpy_d2h(rc.data(), rc_, (size_t)n * 4, nullptr));
    double xe = 0; int32 correct = 0, valid = 0;
    for (int32 i = 0; i < n; i++) { xe += rx[i]; correct += (rc[i] == 1.f); valid += (frame_mask[i] == 1.f); }
    loss_ += xe; correct_ += correct; frames_ += valid;                                   // :138-142 (entropy of one-hot targets is 0)
  }
  std::string Report() const {                                                            // :293-307
    std::ostringstream oss;
    oss << "AvgLoss: " << (loss_ - entropy_) / frames_ << " (Xent), " << "[AvgXent: " << loss_ / frames_
        << ", AvgTargetEnt: " << entropy_ / frames_ << "]" << std::endl;
    oss << "\nFRAME_ACCURACY >> " << 100.0 * correct_ / frames_ << "% <<";
    return oss.str();
  }
  double AvgLoss() const { return (loss_ - entropy_) / frames_; }
  double FrameAccuracy() const { return (double)correct_ / frames_; }
  double Frames() const { return frames_; }
 private:
  double frames_, correct_, loss_, entropy_;
  int32 *tgt_; BaseFloat *mask_, *rx_, *rc_;
  size_t cap_;
  int32 *poff_, *ppdf_; BaseFloat *pw_, *re_;      // CSR posterior and per-row target entropy of the general EvalMasked
  size_t pcap_, ecap_;
};

struct TrainLstmStreamsOptions {              // bd-nnet-train-lstm-streams.cc:27-71 (the options that matter)
  NnetTrainOptions trn_opts;
  int32 targets_delay = 5, batch_size = 20, num_stream = 4;
  bool crossvalidate = false;
};
struct TrainLstmStreamsStats { int32 num_done = 0; double total_frames = 0, seconds = 0, avg_loss = 0, frame_accuracy = 0; int32 num_minibatches = 0; };

// The while(1) loop of bd-nnet-train-lstm-streams.cc:143-282 on in-memory utterances.
inline TrainLstmStreamsStats TrainLstmStreams(Nnet *nnet, const std::vector<Utterance> &utts, const TrainLstmStreamsOptions &o,
                                              std::string *report = nullptr) {
  nnet->SetTrainOptions(o.trn_opts);                                                  // :104
  MultiStreamBatcher batcher(&utts, o.num_stream, o.batch_size, o.targets_delay);
  Xent xent;
  StreamBatch b;
  DeviceMatrix feat_dev, nnet_out, obj_diff;
  TrainLstmStreamsStats st;
  const auto t0 = std::chrono::steady_clock::now();
  while (batcher.Next(&b)) {
    nnet->Reset(b.new_utt_flags);                                                     // :209
    feat_dev.CopyFromHost(b.feat.data(), o.batch_size * o.num_stream, b.dim);         // :212 CuMatrix(feat)  (no feature transform)
    nnet->Propagate(feat_dev.View(), &nnet_out);                                      // :215
    xent.EvalMasked(b.frame_mask, nnet_out, b.target, &obj_diff);                     // :219
    if (!o.crossvalidate) nnet->Backpropagate(obj_diff.View(), nullptr);              // :227-229
    st.total_frames += b.NumValidFrames();                                            // :241
    st.num_minibatches++;
  }
  KCheck(klstm_stream_synchronize(nullptr));
  st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  st.num_done = batcher.NumDone();
  st.avg_loss = xent.AvgLoss();
  st.frame_accuracy = xent.FrameAccuracy();
  if (report) *report = xent.Report();
  return st;
}

}  // namespace klstm_kaldi
