"""Data parallelism over utterance streams.

Streams are independent utterances; the only cross-stream coupling in the reference is the sum over
rows in the seven gradient accumulations (...streams.h:468-487).  Rank g owns streams
[g*S/G, (g+1)*S/G): its own carried state, activation planes and batcher slice, and a full replica of
the parameters and momentum buffers.  Per minibatch there is exactly ONE collective: an all-reduce
(sum, fp32) of the contiguous gradient blob (RCCL over xGMI through torch.distributed's "nccl"
backend; "gloo" in the CPU tests).

Semantics preserved: the reference folds momentum into the gradient GEMM's beta (:465-487); here each
rank produces the PURE local gradient (beta = 0), the blob is summed over ranks, and only then
corr = momentum*corr + sum(grad) and theta -= lr*corr run, identically on every rank.  The result
equals the single-GPU S_total run up to fp32 summation order.  Gradients are summed, not averaged
(the reference's lr = 1e-5 is tuned for sums).
"""
import torch


def shard_time_major(mat, num_stream_total, rank, world):
    """Rows of a time-major minibatch matrix [T*S_total, D] that belong to this rank's streams,
    as a contiguous time-major [T*S_local, D] tensor/array."""
    assert num_stream_total % world == 0, "streams must divide evenly over ranks"
    s_local = num_stream_total // world
    T = mat.shape[0] // num_stream_total
    v = mat.reshape(T, num_stream_total, mat.shape[1])[:, rank * s_local:(rank + 1) * s_local, :]
    return v.reshape(T * s_local, mat.shape[1])


class DataParallelLstm:
    """Drives one engine per rank.  `engine` needs: reset, propagate, backpropagate(.., flags),
    grad_blob_tensor(), apply_momentum, update (kaldi_lstm_amd.Engine provides exactly these)."""

    DEFER_MOMENTUM = 1

    def __init__(self, engine, group=None, force_collective=False):
        import torch.distributed as dist
        self.engine = engine
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collective: take the all-reduce code path even with one rank (tests)
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self._blob = None

    def broadcast_params(self, src=0):
        """Make every replica start from rank `src`'s parameters."""
        if self.world == 1:
            return
        p = torch.from_numpy(self.engine.get_params())
        if self.dist.get_backend(self.group) == "nccl":
            p = p.cuda()
        self.dist.broadcast(p, src=src, group=self.group)
        self.engine.set_params(p.cpu().numpy())

    def train_step(self, x, out, out_diff, in_diff, momentum, learn_rate, reset_flags=None):
        e = self.engine
        if reset_flags is not None:
            e.reset(reset_flags)
        e.propagate(x, out)
        if not self.collective:
            e.backpropagate(x, out_diff, in_diff, momentum, 0)
        else:
            e.backpropagate(x, out_diff, in_diff, momentum, self.DEFER_MOMENTUM)
            if self._blob is None:
                self._blob = e.grad_blob_tensor()
            self.dist.all_reduce(self._blob, op=self.dist.ReduceOp.SUM, group=self.group)
            e.apply_momentum(momentum)
        e.update(learn_rate)
