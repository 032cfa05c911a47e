// include/klstm_trainer.hpp -- host-side multi-stream BPTT batcher, the piece of
// google/nnetbin/bd-nnet-train-lstm-streams.cc (:128-206) that defines exactly which `in` rows, reset
// flags, padded targets and frame mask the component sees.  Kaldi table I/O is replaced by an in-memory
// utterance list (file plumbing is out of scope); the stream book-keeping is the reference's.
#pragma once
#include <cstring>
#include <vector>

#include "klstm_kaldi_io.hpp"

namespace klstm_kaldi {

struct Utterance {
  int32 num_frames = 0, dim = 0;
  std::vector<BaseFloat> feats;     // [num_frames x dim] row-major
  std::vector<int32> targets;       // one pdf-id per frame (a one-hot Posterior)
};

struct StreamBatch {
  int32 num_stream = 0, batch_size = 0, dim = 0;
  std::vector<BaseFloat> feat;          // [batch_size*num_stream x dim], time-major rows t*S+s  (:138)
  std::vector<int32> target;            // [batch_size*num_stream]                               (:139)
  std::vector<BaseFloat> frame_mask;    // 1 valid, 0 padded                                     (:137)
  std::vector<int> new_utt_flags;       // per stream, 1 = a new utterance starts in this batch   (:135)
  int32 NumValidFrames() const { int32 n = 0; for (BaseFloat m : frame_mask) n += (m == 1.f); return n; }
};

class MultiStreamBatcher {
 public:
  MultiStreamBatcher(const std::vector<Utterance> *utts, int32 num_stream, int32 batch_size, int32 targets_delay)
      : utts_(utts), pos_(0), S_(num_stream), T_(batch_size), delay_(targets_delay), cur_(num_stream, nullptr),
        curt_(num_stream, 0), lent_(num_stream, 0), flags_(num_stream, 0), num_done_(0), num_other_error_(0) {
    KLSTM_ASSERT(num_stream > 0 && batch_size > 0 && targets_delay >= 0);
  }

  // Fills `b` with the next minibatch; returns false when every stream is exhausted (:177-181).
  bool Next(StreamBatch *b) {
    for (int32 s = 0; s < S_; s++) {                                   // :146-174
      if (curt_[s] < lent_[s]) { flags_[s] = 0; continue; }
      while (pos_ < utts_->size()) {
        const Utterance &u = (*utts_)[pos_++];
        if (u.num_frames != (int32)u.targets.size()) { num_other_error_++; continue; }   // :160-164
        cur_[s] = &u; curt_[s] = 0; lent_[s] = u.num_frames; flags_[s] = 1; num_done_++;
        break;
      }
    }
    bool done = true;
    for (int32 s = 0; s < S_; s++) if (curt_[s] < lent_[s]) done = false;
    if (done) return false;
    for (int32 s = 0; s < S_; s++)
      if (lent_[s] == 0)   // the reference would read targets[s][-1] here (:195, :201)
        KLSTM_ERR("MultiStreamBatcher: fewer utterances than streams, stream " << s << " never received data");
    const int32 dim = cur_[0]->dim;
    b->num_stream = S_; b->batch_size = T_; b->dim = dim;
    b->feat.assign((size_t)T_ * S_ * dim, 0.f);
    b->target.assign((size_t)T_ * S_, 0);
    b->frame_mask.assign((size_t)T_ * S_, 0.f);
    for (int32 t = 0; t < T_; t++) {                                    // :187-206
      for (int32 s = 0; s < S_; s++) {
        const Utterance &u = *cur_[s];
        const int32 row = t * S_ + s, cur = curt_[s], len = lent_[s];
        if (cur < len) { b->frame_mask[row] = 1.f; b->target[row] = u.targets[cur]; }
        else { b->frame_mask[row] = 0.f; b->target[row] = u.targets[len - 1]; }
        const int32 src = (cur + delay_ < len) ? cur + delay_ : len - 1;
        std::memcpy(&b->feat[(size_t)row * dim], &u.feats[(size_t)src * dim], (size_t)dim * sizeof(BaseFloat));
        curt_[s]++;
      }
    }
    b->new_utt_flags = flags_;
    return true;
  }
  int32 NumDone() const { return num_done_; }
  int32 NumOtherError() const { return num_other_error_; }

 private:
  const std::vector<Utterance> *utts_;
  size_t pos_;
  int32 S_, T_, delay_;
  std::vector<const Utterance *> cur_;
  std::vector<int32> curt_, lent_;
  std::vector<int> flags_;
  int32 num_done_, num_other_error_;
};

}  // namespace klstm_kaldi
