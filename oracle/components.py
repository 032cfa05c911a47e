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
