"""N>1 data-parallel path on CPU: world_size=2, gloo.  The host logic under test is
kaldi_lstm_amd.DataParallelLstm + shard_time_major; the per-rank engine is an oracle-backed stand-in
(the HIP engine needs a GPU; the GPU variant of this test runs under -m gpu in test_dp_gpu below)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.oracle import Oracle, make_params

I, C, R, S_TOTAL, T, NSTEP = 6, 8, 5, 4, 5, 4
LR, MMT = 1e-2, 0.9


class OracleEngine:
    """CPU stand-in exposing the subset of kaldi_lstm_amd.Engine that DataParallelLstm uses."""

    def __init__(self, S, params, dtype=np.float64):
        self.o = Oracle(I, C, R, S, dtype)
        self.o.set_params(params)
        self.dtype = dtype
        self._grad = torch.zeros(self.o.num_params, dtype=torch.float64)

    def reset(self, flags):
        self.o.reset(flags)

    def propagate(self, x, out):
        out.copy_(torch.from_numpy(self.o.propagate(x.numpy())))

    def backpropagate(self, x, out_diff, in_diff, momentum, flags):
        if flags & 1:                     # DEFER_MOMENTUM: pure local gradient, momentum buffers untouched
            saved = self.o.get_corr()
            self.o.set_corr(np.zeros_like(saved))
            d = self.o.backpropagate(x.numpy(), out_diff.numpy(), momentum=0.0)
            self._grad.copy_(torch.from_numpy(self.o.get_corr()))
            self.o.set_corr(saved)
        else:
            d = self.o.backpropagate(x.numpy(), out_diff.numpy(), momentum=momentum)
        if in_diff is not None:
            in_diff.copy_(torch.from_numpy(d))

    def grad_blob_tensor(self):
        return self._grad

    def apply_momentum(self, momentum):
        self.o.set_corr(momentum * self.o.get_corr() + self._grad.numpy())

    def update(self, lr):
        self.o.update(lr)

    def get_params(self):
        return self.o.get_params()

    def set_params(self, p):
        self.o.set_params(p)


def data():
    rng = np.random.RandomState(0)
    xs = [rng.randn(T * S_TOTAL, I) for _ in range(NSTEP)]
    ods = [rng.randn(T * S_TOTAL, R) for _ in range(NSTEP)]
    return make_params(I, C, R, scale=0.3, seed=1, dtype=np.float64), xs, ods


def worker(rank, world, port, q):
    import kaldi_lstm_amd as k
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, xs, ods = data()
    s_local = S_TOTAL // world
    # replicas deliberately start different: broadcast_params must fix that
    eng = OracleEngine(s_local, params if rank == 0 else params * 0.0)
    dp = k.DataParallelLstm(eng)
    dp.broadcast_params(src=0)
    outs = []
    for i in range(NSTEP):
        x = torch.from_numpy(np.ascontiguousarray(k.shard_time_major(xs[i], S_TOTAL, rank, world)))
        od = torch.from_numpy(np.ascontiguousarray(k.shard_time_major(ods[i], S_TOTAL, rank, world)))
        out = torch.empty(T * s_local, R, dtype=torch.float64)
        ind = torch.empty(T * s_local, I, dtype=torch.float64)
        dp.train_step(x, out, od, ind, MMT, LR, reset_flags=[1] * s_local if i == 0 else None)
        outs.append((out.numpy().copy(), ind.numpy().copy()))
    q.put((rank, eng.get_params(), outs))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, all streams, reference momentum folding (...streams.h:465-487)
    params, xs, ods = data()
    o = Oracle(I, C, R, S_TOTAL, np.float64)
    o.set_params(params)
    full = []
    for i in range(NSTEP):
        out = o.propagate(xs[i]); ind = o.backpropagate(xs[i], ods[i], momentum=MMT); o.update(LR)
        full.append((out, ind))
    import kaldi_lstm_amd as k
    for rank, p_rank, outs in res:
        np.testing.assert_allclose(p_rank, o.get_params(), rtol=1e-10, atol=1e-12)      # replicas identical to the S_total run
        for i in range(NSTEP):
            np.testing.assert_allclose(outs[i][0], k.shard_time_major(full[i][0], S_TOTAL, rank, 2), rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(outs[i][1], k.shard_time_major(full[i][1], S_TOTAL, rank, 2), rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(res[0][1], res[1][1])                                  # and bit-identical to each other


def oneshot_refusal_worker(rank, world, port, q):
    import kaldi_lstm_amd as k
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, xs, ods = data()

    class Eng(OracleEngine):
        def grad_blob_tensor(self, full=False):
            if full and rank == 1:
                raise RuntimeError("no peer mapping on this rank")
            return self._grad                 # (a CPU tensor: rank 0's export is refused too, for a different reason)
    eng = Eng(S_TOTAL // world, params)
    dp = k.DataParallelLstm(eng, oneshot=True)       # must return on BOTH ranks, with the same decision
    q.put((rank, dp.oneshot is None, dp.oneshot_note, dp.collective_in_use()))
    dist.barrier()
    dist.destroy_process_group()


def test_oneshot_exchange_that_one_rank_cannot_set_up_is_dropped_by_all_ranks():
    """bench.py --gpus N asks for the one-shot exchange; a rank that cannot export or open a handle must not leave the others
    waiting in a collective: the ranks agree and everybody stays on the ordinary all-reduce (dp.py _setup_oneshot)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=oneshot_refusal_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, none, note, name in res:
        assert none and "rank 1: no peer mapping on this rank" in note and "torch.distributed" in name
    assert res[0][2] == res[1][2]


def test_shard_time_major_layout():
    import kaldi_lstm_amd as k
    m = np.arange(3 * 4 * 2).reshape(12, 2)          # T=3, S=4
    a = k.shard_time_major(m, 4, 1, 2)               # streams 2,3
    assert a.shape == (6, 2)
    assert np.array_equal(a[:, 0] // 2, [2, 3, 6, 7, 10, 11])      # rows t*4+s for s in {2,3}
    with pytest.raises(AssertionError):
        k.shard_time_major(m, 4, 0, 3)


# ---- stacked net (LSTM x2 + Affine + Softmax + masked Xent), ONE all-reduce of the fused blob per minibatch ----
DIMS = (6, 8, 5, 2, 11)      # I, C, R, n_lstm, n_out


def stack_data():
    rng = np.random.RandomState(3)
    xs = [rng.randn(T * S_TOTAL, DIMS[0]) for _ in range(NSTEP)]
    tg = [rng.randint(0, DIMS[4], T * S_TOTAL).astype(np.int32) for _ in range(NSTEP)]
    mk = [(rng.rand(T * S_TOTAL) > 0.2).astype(np.float32) for _ in range(NSTEP)]     # padded tail frames
    return xs, tg, mk


def stack_worker(rank, world, port, q, overlap=False):
    import kaldi_lstm_amd as k
    from tests import nnet_twins as tw
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s_local = S_TOTAL // world
    lstm, W, b = tw.make_stack(DIMS, s_local, seed=5)
    layers = tw.cpu_layers(DIMS, s_local, lstm, W, b)
    calls = {"n": 0}
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(*a, **kw):
        calls["n"] += 1
        return real_all_reduce(*a, **kw)
    dist.all_reduce = counting_all_reduce
    net = k.DataParallelNnet(layers, tw.NumpyLoss(), alloc=lambda n: torch.zeros(n, dtype=torch.float64), overlap=overlap)
    xs, tg, mk = stack_data()
    stats = []
    for i in range(NSTEP):
        x = torch.from_numpy(np.ascontiguousarray(k.shard_time_major(xs[i], S_TOTAL, rank, world)))
        t = torch.from_numpy(np.ascontiguousarray(k.shard_time_major(tg[i][:, None], S_TOTAL, rank, world)[:, 0]))
        m = torch.from_numpy(np.ascontiguousarray(k.shard_time_major(mk[i][:, None], S_TOTAL, rank, world)[:, 0]))
        stats.append(net.train_step(x, t, m, MMT, LR, reset_flags=[1] * s_local if i == 0 else None))
    q.put((rank, [l.params() for l in layers], stats, calls["n"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_two_rank_stacked_net_one_allreduce_equals_single_process(overlap):
    """overlap=False: exactly ONE collective per minibatch for the whole model; overlap=True: one asynchronous collective
    per layer, issued as that layer's gradient appears (3 per minibatch here).  Both equal the single-process run."""
    from tests import nnet_twins as tw
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=stack_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, all streams, no collective
    import kaldi_lstm_amd as k
    lstm, W, b = tw.make_stack(DIMS, S_TOTAL, seed=5)
    layers = tw.cpu_layers(DIMS, S_TOTAL, lstm, W, b)
    net = k.DataParallelNnet(layers, tw.NumpyLoss(), alloc=lambda n: torch.zeros(n, dtype=torch.float64))
    xs, tg, mk = stack_data()
    full = [net.train_step(torch.from_numpy(xs[i]), torch.from_numpy(tg[i]), torch.from_numpy(mk[i]), MMT, LR,
                           reset_flags=[1] * S_TOTAL if i == 0 else None) for i in range(NSTEP)]
    for rank, params, stats, ncalls in res:
        assert ncalls == NSTEP * (len(layers) if overlap else 1)
        for pr, l in zip(params, layers):
            np.testing.assert_allclose(pr, l.params(), rtol=1e-9, atol=1e-12)
    for i in range(NSTEP):                                       # loss statistics add up over ranks
        assert abs(res[0][2][i][0] + res[1][2][i][0] - full[i][0]) <= 1e-9 * abs(full[i][0])
        assert res[0][2][i][1] + res[1][2][i][1] == full[i][1]
        assert res[0][2][i][2] + res[1][2][i][2] == full[i][2]
