"""Host-side multi-stream BPTT batcher: the Python twin of include/klstm_trainer.hpp, i.e. the piece of
google/nnetbin/bd-nnet-train-lstm-streams.cc (:128-206) that defines which `in` rows, reset flags, padded targets and
frame mask the component sees.  Host bookkeeping only (no arithmetic on features); used by bench.py's ragged-length
run and by launch scripts.  Streams take a new utterance only at minibatch boundaries (:146-174); inside a
minibatch row t*S+s carries feats[min(cur+delay, len-1)] (:196-200), targets[min(cur, len-1)] and mask 1/0 (:191-195).
"""
import numpy as np


class MultiStreamBatcher:
    def __init__(self, utts, num_stream, batch_size, targets_delay):
        """utts: sequence of (feats [len, dim] float32, targets [len] int) consumed in order."""
        self.utts = list(utts)
        self.pos = 0
        self.S, self.T, self.delay = int(num_stream), int(batch_size), int(targets_delay)
        self.cur = [None] * self.S
        self.curt = [0] * self.S
        self.lent = [0] * self.S
        self.flags = [0] * self.S

    def next(self):
        """(feat [T*S, dim], target [T*S], frame_mask [T*S], new_utt_flags [S]) or None when every stream is exhausted."""
        S, T = self.S, self.T
        for s in range(S):
            if self.curt[s] < self.lent[s]:
                self.flags[s] = 0
                continue
            while self.pos < len(self.utts):
                f, t = self.utts[self.pos]
                self.pos += 1
                if f.shape[0] != len(t):                 # length mismatch: skipped (:160-164)
                    continue
                self.cur[s] = (f, t)
                self.curt[s], self.lent[s], self.flags[s] = 0, f.shape[0], 1
                break
        if all(self.curt[s] >= self.lent[s] for s in range(S)):
            return None
        if any(self.lent[s] == 0 for s in range(S)):
            raise ValueError("MultiStreamBatcher: fewer utterances than streams (the reference reads targets[-1] here, :195)")
        dim = self.cur[0][0].shape[1]
        feat = np.empty((T, S, dim), np.float32)
        target = np.empty((T, S), np.int32)
        mask = np.empty((T, S), np.float32)
        tt = np.arange(T)
        for s in range(S):
            f, tg = self.cur[s]
            cur, ln = self.curt[s], self.lent[s]
            idx = cur + tt
            mask[:, s] = idx < ln
            target[:, s] = np.asarray(tg)[np.minimum(idx, ln - 1)]
            feat[:, s] = f[np.minimum(idx + self.delay, ln - 1)]
            self.curt[s] += T
        return feat.reshape(T * S, dim), target.reshape(T * S), mask.reshape(T * S), list(self.flags)
