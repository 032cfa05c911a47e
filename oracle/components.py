"""CPU restatements (numpy) of the small caller-side pieces around the hot path.
TEST INFRASTRUCTURE ONLY (see oracle/lstmp_oracle.c header).  PARITY UNPINNED: the reference has
no tests or fixtures for these either; each function cites the lines it follows.
"""
import numpy as np


def time_shift(x, shift):
    """TimeShift::PropagateFnc, /root/reference/standard/nnet/nnet-time-shift.h:42-51:
    out[dst] = in[clamp(dst + shift, 0, num_frames-1)]."""
    n = x.shape[0]
    src = np.clip(np.arange(n) + shift, 0, n - 1)
    return x[src].copy()


def transmit(x):
    """TransmitComponent, /root/reference/standard/nnet/nnet-transmit-component.h:26-33: identity both ways."""
    return x.copy()


class MultiStreamBatcher:
    """The multi-stream BPTT batcher of /root/reference/google/nnetbin/bd-nnet-train-lstm-streams.cc:128-206.

    utts: list of (feats [len x dim] float32, targets [len] int) consumed in order (the
    SequentialBaseFloatMatrixReader).  next() returns None when every stream is exhausted, else
    (feat [T*S x dim], target [T*S], frame_mask [T*S], new_utt_flags [S]) with time-major rows t*S+s.
    """

    def __init__(self, utts, num_stream, batch_size, targets_delay):
        self.utts = list(utts)
        self.pos = 0
        self.S, self.T, self.delay = num_stream, batch_size, targets_delay
        self.feats = [None] * num_stream          # :132-137 book-keeping
        self.targets = [None] * num_stream
        self.curt = [0] * num_stream
        self.lent = [0] * num_stream
        self.new_utt_flags = [0] * num_stream

    def next(self):
        S, T = self.S, self.T
        for s in range(S):                                            # :146-174
            if self.curt[s] < self.lent[s]:
                self.new_utt_flags[s] = 0
                continue
            while self.pos < len(self.utts):
                f, t = self.utts[self.pos]
                self.pos += 1
                if f.shape[0] != len(t):                              # :160-164 length mismatch: skip
                    continue
                self.feats[s], self.targets[s] = f, t
                self.curt[s], self.lent[s] = 0, f.shape[0]
                self.new_utt_flags[s] = 1
                break
        if all(self.curt[s] >= self.lent[s] for s in range(S)):       # :177-181
            return None
        if any(self.lent[s] == 0 for s in range(S)):
            # latent out-of-bounds read in the reference (targets[s][lent[s]-1] with lent == 0, :195,:201)
            raise ValueError("fewer utterances than streams: a stream never received data")
        dim = self.feats[0].shape[1]
        feat = np.zeros((T * S, dim), np.float32)
        target = np.zeros(T * S, np.int64)
        mask = np.zeros(T * S, np.float32)
        for t in range(T):                                            # :187-206
            for s in range(S):
                cur, ln = self.curt[s], self.lent[s]
                if cur < ln:
                    mask[t * S + s] = 1
                    target[t * S + s] = self.targets[s][cur]
                else:
                    mask[t * S + s] = 0
                    target[t * S + s] = self.targets[s][ln - 1]
                if cur + self.delay < ln:
                    feat[t * S + s] = self.feats[s][cur + self.delay]
                else:
                    feat[t * S + s] = self.feats[s][ln - 1]
                self.curt[s] += 1
        return feat, target, mask, list(self.new_utt_flags)


# ---- output tail: AffineTransform / Softmax are [UPSTREAM-unvendored] nnet1 components (only their use is in
# the reference: google/nnet.proto:4-5, README.md:27-28); Xent::EvalMasked is vendored. ----

def affine_propagate(x, W, b):
    """AffineTransform::PropagateFnc: out = in * linearity^T + bias (linearity is [out_dim x in_dim])."""
    return (x @ W.T + b).astype(x.dtype)


def affine_backpropagate(out_diff, W):
    """AffineTransform::BackpropagateFnc: in_diff = out_diff * linearity."""
    return (out_diff @ W).astype(out_diff.dtype)


def affine_update(x, out_diff, W, b, W_corr, b_corr, lr, lr_bias, momentum):
    """AffineTransform::Update without l1/l2 (the recipe sets none, train_lstm_streams.sh):
    corr = momentum*corr + grad ; param -= lr*corr.  Arrays are updated in place."""
    W_corr[...] = momentum * W_corr + out_diff.T @ x
    b_corr[...] = momentum * b_corr + out_diff.sum(0)
    W -= lr * W_corr
    b -= lr_bias * b_corr


def softmax(x):
    e = np.exp(x - x.max(1, keepdims=True))
    return (e / e.sum(1, keepdims=True)).astype(x.dtype)


def xent_eval_masked(net_out, target, frame_mask):
    """Xent::EvalMasked, /root/reference/google/nnet/nnet-loss.cc:76-142, for one-hot targets.
    Returns diff, cross_entropy (sum), entropy (sum), correct (count), valid_frames."""
    n, d = net_out.shape
    t = np.zeros((n, d), net_out.dtype)                      # :86-96 posterior -> dense matrix
    t[np.arange(n), target] += 1.0
    diff = (net_out - t) * frame_mask[:, None]               # :102-107
    correct = int(np.sum((frame_mask == 1) & (net_out.argmax(1) == t.argmax(1))))      # :109-120
    cross_entropy = float(-np.sum(np.log(net_out) * t * frame_mask[:, None]))           # :122-128
    entropy = float(-np.sum(np.log(t + 1e-20) * t * frame_mask[:, None]))               # :130-136
    return diff, cross_entropy, entropy, correct, int(frame_mask.sum())


def xent_eval_masked_post(net_out, post, frame_mask):
    """Xent::EvalMasked, /root/reference/google/nnet/nnet-loss.cc:76-142, for GENERAL posteriors: post[t] = list of
    (pdf, weight).  Dense restatement, statement by statement.  Returns diff, cross_entropy, entropy, correct, valid_frames."""
    n, d = net_out.shape
    assert n == len(post)                                    # :82
    t = np.zeros((n, d), net_out.dtype)                      # :85 zero-filled
    for fr in range(n):                                      # :86-96
        for pdf, w in post[fr]:
            if pdf >= d:
                raise ValueError("Posterior pdf-id out of NN-output dimension")
            t[fr, pdf] += net_out.dtype.type(w)
    diff = (net_out - t) * frame_mask[:, None]               # :102-107
    correct = int(np.sum((frame_mask == 1) & (net_out.argmax(1) == t.argmax(1))))      # :109-120 (FindRowMaxId: first maximum)
    with np.errstate(divide="ignore", invalid="ignore"):
        cross_entropy = float(-np.sum(np.log(net_out) * t * frame_mask[:, None], dtype=np.float64))          # :122-128
        entropy = float(-np.sum(np.log(t + net_out.dtype.type(1e-20)) * t * frame_mask[:, None], dtype=np.float64))   # :130-136
    return diff, cross_entropy, entropy, correct, int(frame_mask.sum())
