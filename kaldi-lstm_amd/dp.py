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


def native_comm(dist, group=None, device=None):
    """RCCL communicator created through the C-ABI of libklstm.so (klstm_comm_*), its 128-byte id handed from rank 0 to
    the other ranks over the already initialised torch.distributed group (launcher plumbing only: the gradient all-reduce
    itself is then issued by klstm_allreduce_grads / klstm_allreduce_buffer on the engine's stream).  None when the
    process group is not on GPUs (gloo CPU tests keep torch.distributed's all_reduce)."""
    import sys
    import torch
    from .binding import RcclComm
    if not dist.is_initialized() or dist.get_backend(group) != "nccl":
        return None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.cuda.current_device() if device is None else device
    # Every rank takes the same decision: rank 0's id (or its failure to make one) is broadcast, and after the collective
    # communicator creation the ranks agree on whether ALL of them succeeded; otherwise everybody falls back to
    # torch.distributed's all_reduce (a rank that raised on its own would leave the others waiting in a collective).
    msg = torch.zeros(129, dtype=torch.uint8, device="cuda")
    if rank == 0:
        try:
            msg[1:] = torch.tensor(list(RcclComm.unique_id()), dtype=torch.uint8)
            msg[0] = 1
        except Exception as ex:
            print("kaldi_lstm_amd.dp: no RCCL unique id (%s)" % ex, file=sys.stderr)
    dist.broadcast(msg, src=0, group=group)
    host = msg.cpu().tolist()
    comm = None
    if host[0] == 1:
        try:
            comm = RcclComm(world, rank, device=dev, uid=bytes(host[1:]))
        except Exception as ex:
            print("kaldi_lstm_amd.dp: rank %d could not join the library-owned RCCL communicator (%s)" % (rank, ex), file=sys.stderr)
    ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        if comm is not None:
            comm.close()
        if rank == 0:
            print("kaldi_lstm_amd.dp: the all-reduce goes through torch.distributed instead of klstm_allreduce_*", file=sys.stderr)
        return None
    return comm


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
    FUSE_UPDATE = 2       # klstm.h: the Update follows immediately (it does, two lines below)

    def __init__(self, engine, group=None, force_collective=False, require_native=False, oneshot=False):
        """require_native: fail instead of falling back to torch.distributed's all_reduce when the library-owned RCCL
        communicator cannot be made on a GPU process group (bench.py --gpus N: the line must say what it measured)."""
        import torch.distributed as dist
        self.engine = engine
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collective: take the all-reduce code path even with one rank (tests)
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self._blob = None
        # on GPUs the collective is libklstm.so's own (klstm_allreduce_grads: RCCL, in place, on the engine's stream)
        self.comm = native_comm(dist, group) if self.collective and hasattr(engine, "allreduce_grads") else None
        on_gpu = dist.is_initialized() and dist.get_backend(group) == "nccl"
        if self.collective and require_native and on_gpu and self.comm is None:
            raise RuntimeError("data-parallel run on GPUs without the library-owned RCCL communicator (klstm_comm_init_rank failed on "
                               "some rank, see stderr): refusing to fall back to torch.distributed.all_reduce silently")
        # what the gradient all-reduce of train_step goes through, and how many ranks it really spans
        self.collective_name = ("none (single rank)" if not self.collective else
                                "klstm_allreduce_grads (RCCL ncclAllReduce, in place, on the engine's stream)" if self.comm is not None else
                                "torch.distributed.all_reduce (%s)" % (dist.get_backend(group) if dist.is_initialized() else "?"))
        self.ranks_seen = self.comm.count() if self.comm is not None else self.world
        if self.collective and self.comm is None and on_gpu and hasattr(engine, "set_option"):
            # torch.distributed's all_reduce over the gradient alone: the library cannot put its validity word through it, so a rank
            # whose persistent chain gave up has to be found BEFORE the collective -- every persistent call waits for its launch
            engine.set_option("persist_verify", 1)
        # oneshot=True (default OFF; klstm_oneshot.hip: prepared, never run across devices): the gradient blob of every rank is
        # mapped into every other rank (hipIpc handles handed around over the process group) and ONE kernel per rank reduces its
        # 1/N slice from all peers and writes it back to all of them, instead of ncclAllReduce
        self.oneshot = None
        self.use_oneshot = False
        self._rccl_name = self.collective_name
        self.oneshot_note = None
        if oneshot and self.collective and hasattr(engine, "grad_blob_tensor"):
            self._setup_oneshot(engine, dist, group)

    def _setup_oneshot(self, engine, dist, group):
        """Every rank takes the same decision (as in native_comm): a rank that cannot export or open a handle says so IN the
        collectives everybody takes part in -- raising on its own would leave the others waiting -- and then nobody uses the
        exchange (`oneshot_note` says why)."""
        import sys
        from .binding import OneshotAllreduce
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        one, handles, err = None, None, None
        try:
            # (the whole blob: the validity word of data-parallel runs rides behind the gradient, klstm.h klstm_grad_blob_len)
            self._blob_full = engine.grad_blob_tensor(full=True)
            self._blob = self._blob_full[:engine.num_params]
            one = OneshotAllreduce(self._blob_full, device=self._blob_full.device.index or 0)
            handles = one.export()
        except Exception as ex:
            err = "rank %d: %s" % (rank, ex)
        hs = [None] * self.world
        dist.all_gather_object(hs, (handles, err), group=group)
        errs = [h[1] for h in hs if h[1]]
        if not errs:
            try:
                one.connect(rank, self.world, [h[0] for h in hs])
            except Exception as ex:
                err = "rank %d: %s" % (rank, ex)
            es = [None] * self.world
            dist.all_gather_object(es, err, group=group)
            errs = [x for x in es if x]
        if errs:
            if one is not None:
                one.close()
            self.oneshot_note = "one-shot exchange not available (%s)" % "; ".join(errs)
            if rank == 0:
                print("kaldi_lstm_amd.dp: " + self.oneshot_note + ": the all-reduce stays on " + self._rccl_name, file=sys.stderr)
            return
        self.oneshot = one
        self.use_oneshot = True              # (a caller that wants both at hand -- bench.py's A/B -- switches this flag)
        self.collective_name = "klstm_allreduce_grads_oneshot (peer-mapped blobs, one kernel per rank; EXPERIMENTAL)"

    def collective_in_use(self):
        return self.collective_name if (self.oneshot is not None and self.use_oneshot) or self.oneshot is None else self._rccl_name

    def broadcast_params(self, src=0):
        """Make every replica start from rank `src`'s parameters (device to device on GPUs)."""
        if self.world == 1:
            return
        if self.dist.get_backend(self.group) == "nccl" and hasattr(self.engine, "param_blob_tensor"):
            p = self.engine.param_blob_tensor().clone()
            self.dist.broadcast(p, src=src, group=self.group)
            self.engine.set_params_device(p)
            return
        p = torch.from_numpy(self.engine.get_params())
        self.dist.broadcast(p, src=src, group=self.group)
        self.engine.set_params(p.numpy())

    def train_step(self, x, out, out_diff, in_diff, momentum, learn_rate, reset_flags=None):
        e = self.engine
        if reset_flags is not None:
            e.reset(reset_flags)
        e.propagate(x, out)
        if not self.collective:
            e.backpropagate(x, out_diff, in_diff, momentum, self.FUSE_UPDATE)
        else:
            e.backpropagate(x, out_diff, in_diff, momentum, self.DEFER_MOMENTUM)
            if self.oneshot is not None and self.use_oneshot:
                self.oneshot.allreduce_engine(e)
            elif self.comm is not None:
                e.allreduce_grads(self.comm)
            else:
                if self._blob is None:
                    self._blob = e.grad_blob_tensor()
                self.dist.all_reduce(self._blob, op=self.dist.ReduceOp.SUM, group=self.group)
            e.apply_momentum(momentum)
        e.update(learn_rate)


# ------------------------------------------------------------------------------------------------
# Stacked nets (BASELINE.json configs[3]: LSTM x2 + AffineTransform + Softmax + masked Xent): the gradient blobs of
# ALL layers live back to back in one buffer and a minibatch still needs exactly ONE all-reduce.
#
# Layer protocol (duck-typed; the device classes below wrap the C-ABI, the CPU tests plug in oracle-backed twins):
#   num_params                      int
#   bind_grad(view)                 this layer's slice of the fused blob (flat tensor)
#   reset(flags)                    new-utterance flags (layers without state ignore it)
#   propagate(x) -> out
#   backpropagate(x, out_diff, want_in_diff) -> in_diff | None     pure LOCAL gradient into the bound slice
#   apply(momentum, lr)             corr = momentum*corr + grad ; theta -= lr*corr   (after the all-reduce)
# Loss protocol:  eval(net_out, targets, mask) -> (diff, xent_sum, correct, valid)
# ------------------------------------------------------------------------------------------------
class LstmDP:
    """One LstmProjectedStreams engine as a layer of DataParallelNnet."""

    def __init__(self, engine):
        self.e = engine
        self.num_params = engine.num_params
        self._out = self._ind = None

    def bind_grad(self, view):
        self.e.bind_grad_blob(view)

    def reset(self, flags):
        self.e.reset(flags)

    def propagate(self, x):
        if self._out is None or self._out.shape[0] != x.shape[0]:
            self._out = torch.empty(x.shape[0], self.e.R, device=x.device)
        self.e.propagate(x, self._out)
        return self._out

    def backpropagate(self, x, out_diff, want_in_diff, fused_momentum=None):
        """fused_momentum (single rank, no collective in between): Kaldi's Component::Backpropagate order -- the gradient
        products run inside the following update() as ONE pass with momentum and the step (KLSTM_BPTT_FUSE_UPDATE)."""
        if want_in_diff and (self._ind is None or self._ind.shape[0] != x.shape[0]):
            self._ind = torch.empty(x.shape[0], self.e.I, device=x.device)
        # (round 5: the bf16 gradient tiles carry the fused momentum + Update epilogue too, klstm_kernels.hip gemm_tile_bf16_tn)
        self._fused = fused_momentum is not None
        if self._fused:
            self.e.backpropagate(x, out_diff, self._ind if want_in_diff else None, fused_momentum, DataParallelLstm.FUSE_UPDATE)
        else:
            self.e.backpropagate(x, out_diff, self._ind if want_in_diff else None, 0.0, DataParallelLstm.DEFER_MOMENTUM)
        return self._ind if want_in_diff else None

    def apply(self, momentum, lr):
        if not getattr(self, "_fused", False):
            self.e.apply_momentum(momentum)
        self.e.update(lr)


class AffineDP:
    """AffineTransform (W [out, in], bias [out]) on the device ops of the C-ABI (klstm_affine_*)."""

    def __init__(self, W, bias, ops, stream=None):
        """stream: the torch stream the device ops are issued on (None: the default stream, like an engine created without
        one); a trainer that replays whole minibatches from one hipGraph gives every layer the same explicit stream."""
        self.W, self.bias, self.ops, self.stream = W.contiguous(), bias.contiguous(), ops, stream
        self.num_params = W.numel() + bias.numel()
        self.W_corr, self.b_corr = torch.zeros_like(self.W), torch.zeros_like(self.bias)
        self._out = self._ind = None

    def bind_grad(self, view):
        n = self.W.numel()
        self.gW, self.gb = view[:n].view_as(self.W), view[n:]

    def reset(self, flags):
        pass

    def propagate(self, x):
        if self._out is None or self._out.shape[0] != x.shape[0]:
            self._out = torch.empty(x.shape[0], self.W.shape[0], device=x.device)
        self.ops.affine_propagate(x, self.W, self.bias, self._out, self.stream)
        return self._out

    def backpropagate(self, x, out_diff, want_in_diff, fused_momentum=None):
        """fused_momentum (single rank): the gradient is not materialised -- apply() runs klstm_affine_update (gradient +
        momentum + step in one pass over W) on the x / out_diff of this call, which stay valid until then."""
        self._fused = fused_momentum is not None
        if self._fused:
            self._x, self._od = x, out_diff
        else:
            self.ops.affine_gradient(x, out_diff, self.gW, self.gb, self.stream)
        if not want_in_diff:
            return None
        if self._ind is None or self._ind.shape[0] != x.shape[0]:
            self._ind = torch.empty(x.shape[0], self.W.shape[1], device=x.device)
        self.ops.affine_backpropagate(out_diff, self.W, self._ind, self.stream)
        return self._ind

    def apply(self, momentum, lr):
        if getattr(self, "_fused", False):
            self.ops.affine_update(self._x, self._od, self.W, self.bias, self.W_corr, self.b_corr, lr, lr, momentum, self.stream)
            return
        self.ops.sgd_momentum_update(self.W.view(-1), self.W_corr.view(-1), self.gW.reshape(-1), momentum, lr, self.stream)
        self.ops.sgd_momentum_update(self.bias, self.b_corr, self.gb, momentum, lr, self.stream)


class SoftmaxXentDP:
    """Softmax + Xent::EvalMasked (google/nnet/nnet-loss.cc:76-142) on the device ops of the C-ABI.
    lazy: the statistics stay 0-d device tensors (no host synchronisation per minibatch; the reference's trainer prints
    them every few thousand frames, bd-nnet-train-lstm-streams.cc:240-257).
    accumulate: the statistics are added to `self.totals` (float64[3] on the device: cross entropy, correct, frames --
    Xent's loss_, correct_, frames_, nnet-loss.cc:138-142) by one small launch per minibatch; eval returns None for them."""

    def __init__(self, ops, lazy=False, stream=None, accumulate=False, one_pass=True):
        self.ops, self.lazy, self.stream, self.accumulate = ops, lazy, stream, accumulate
        self.one_pass = one_pass and hasattr(ops, "softmax_xent_masked")   # (the CPU twins of the tests have the two ops only)
        self.totals = None
        self._post = self._diff = self._rows = None

    def eval(self, net_out, targets, mask):
        if self._post is None or self._post.shape != net_out.shape:
            self._post, self._diff = torch.empty_like(net_out), torch.empty_like(net_out)
            self._rows = (torch.empty(net_out.shape[0], device=net_out.device), torch.empty(net_out.shape[0], device=net_out.device))
        if self.accumulate and self.totals is None:
            self.totals = torch.zeros(3, dtype=torch.float64, device=net_out.device)
        if self.one_pass:
            # rows the one-pass kernel serves do not need the posterior matrix at all (klstm.h); the others go through it
            c = net_out.shape[1]
            wide = c % 4 == 0 and 2048 <= c <= 32768 and net_out.stride(0) % 4 == 0 and net_out.data_ptr() % 16 == 0
            xe, correct, valid = self.ops.softmax_xent_masked(net_out, targets, mask, self._diff, post=None if wide else self._post,
                                                              stream=self.stream, lazy=self.lazy, rows_out=self._rows,
                                                              totals=self.totals if self.accumulate else None)
            return self._diff, xe, correct, valid
        self.ops.softmax(net_out, self._post, self.stream)
        xe, correct, valid = self.ops.xent_eval_masked(self._post, targets, mask, self._diff, stream=self.stream, lazy=self.lazy, rows_out=self._rows,
                                                       totals=self.totals if self.accumulate else None)
        return self._diff, xe, correct, valid


class DataParallelNnet:
    """Reset -> Propagate -> loss -> Backpropagate of a stack of layers, then ONE all-reduce (sum) of the fused
    gradient blob and the momentum/update step on every rank (bd-nnet-train-lstm-streams.cc:209-228 per rank, on this
    rank's streams).  `alloc(n)` returns the flat blob storage (torch CUDA float32 for the device layers)."""

    def __init__(self, layers, loss, alloc, group=None, force_collective=False, overlap=False, fuse_single_rank=False):
        """overlap=False: ONE all-reduce of the whole blob after the last Backpropagate (fewest, largest collective).
        overlap=True: one asynchronous all-reduce per layer slice, issued as soon as that layer's gradient exists, so
        the output layer's 34 MB (configs[3]) travel over xGMI while the LSTM layers below still run their BPTT chains;
        all of them are waited for before the first apply().  Same sums either way (disjoint slices)."""
        import torch.distributed as dist
        self.layers, self.loss, self.dist, self.group = layers, loss, dist, group
        self.overlap = overlap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        # single rank, nothing to reduce: layers that can do so run gradient + momentum + step as one pass (what the C++ mirror
        # does with KLSTM_BPTT_FUSE_UPDATE / klstm_affine_update); the fused gradient blob then stays unused
        self.fused = fuse_single_rank and not self.collective
        if self.collective:
            # A persistent launch that gives up leaves this rank's slice of the blob stale; in a stack, the layers above have
            # consumed its output by then.  With a collective in the step the engines wait for their persistent launches and answer
            # a give-up inside the call (klstm.h "persist_verify"), so that what is reduced is always a real gradient.
            for l in layers:
                if isinstance(l, LstmDP) and hasattr(l.e, "set_option"):
                    l.e.set_option("persist_verify", 1)
        pad4 = lambda n: (n + 3) // 4 * 4                  # every slice starts 16-byte aligned (float4 stores)
        self.blob = alloc(sum(pad4(l.num_params) for l in layers))
        # device layers: the all-reduce is libklstm.so's klstm_allreduce_buffer (RCCL) on the layers' stream
        self.comm = native_comm(dist, group) if self.collective and self.blob.is_cuda and not overlap else None
        off = 0
        self.slices = []
        for l in layers:
            l.bind_grad(self.blob[off:off + l.num_params])
            self.slices.append(self.blob[off:off + pad4(l.num_params)])
            off += pad4(l.num_params)

    def train_step(self, x, targets, mask, momentum, learn_rate, reset_flags=None):
        if reset_flags is not None:
            for l in self.layers:
                l.reset(reset_flags)
        acts = [x]
        for l in self.layers:
            acts.append(l.propagate(acts[-1]))
        diff, xent, correct, valid = self.loss.eval(acts[-1], targets, mask)
        pending = []
        for i in range(len(self.layers) - 1, -1, -1):
            if self.fused:
                diff = self.layers[i].backpropagate(acts[i], diff, i > 0, fused_momentum=momentum)
            else:
                diff = self.layers[i].backpropagate(acts[i], diff, i > 0)     # the first layer's in_diff is never used (:228)
            if self.collective and self.overlap:
                pending.append(self.dist.all_reduce(self.slices[i], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.collective and not self.overlap:
            if self.comm is not None:
                self.comm.allreduce(self.blob)
            else:
                self.dist.all_reduce(self.blob, op=self.dist.ReduceOp.SUM, group=self.group)
        for w in pending:
            w.wait()
        for l in self.layers:
            l.apply(momentum, learn_rate)
        return xent, correct, valid
