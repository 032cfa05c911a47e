"""klstm_oneshot_* (kaldi-lstm_amd/csrc/klstm_oneshot.hip): the one-shot all-reduce over peer-mapped gradient blobs that VERDICT r02
asked to be PREPARED behind an option.  It is off by default and has never run across devices (one-GPU lease).  What can run here:
  * a 1-rank group (self-loop): both flag phases against the rank's own flags, the sum of one blob is the blob, repeated calls;
  * two PROCESSES on the one GPU: real hipIpc handles (an offset inside a framework allocation included), both kernels resident at
    once, arrival / departure flags through the peer mapping, sums bit-identical on both ranks and equal to the fp32 sum in rank
    order, several minibatches in a row; and the bounded wait (a rank whose peer never arrives reports phase 0 instead of hanging).
The cross-DEVICE memory ordering it assumes (klstm_oneshot.hip header) stays unverified."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N = 2_181_600          # the 40/800/512 gradient blob


def test_one_rank_self_loop():
    import kaldi_lstm_amd as k
    blob = torch.randn(N, device="cuda")
    ref = blob.clone()
    g = k.OneshotAllreduce(blob)
    g.connect(0, 1, [g.export()])
    s = torch.cuda.Stream()
    for _ in range(3):
        g.allreduce(s)
    s.synchronize()
    assert g.status() == 0
    assert torch.equal(blob, ref)
    g.close()


def _worker(rank, world, port, n, q, absent):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import kaldi_lstm_amd as k
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    pad = torch.empty(1000 + 4 * rank, device="cuda")              # (the blob does not start its allocation: the offset must travel)
    store = torch.empty(n + 64, device="cuda")
    blob = store[32:32 + n]
    g = k.OneshotAllreduce(blob)
    hs = [None] * world
    dist.all_gather_object(hs, g.export())
    g.connect(rank, world, hs)
    s = torch.cuda.Stream()
    out = []
    for it in range(3):
        rng = np.random.RandomState(100 * it + rank)
        blob.copy_(torch.from_numpy(rng.randn(n).astype(np.float32)))
        torch.cuda.synchronize()
        dist.barrier()
        if absent and rank == 1 and it == 2:
            break                                                     # rank 1 never arrives at the third minibatch
        g.allreduce(s, timeout_ms=300 if absent else 5000)
        s.synchronize()
        out.append((g.status(), blob.cpu().numpy().copy()))
    q.put((rank, out))
    dist.barrier()
    g.close()
    dist.destroy_process_group()
    del pad


@pytest.mark.parametrize("absent", [False, True])
def test_two_processes_on_one_gpu(absent):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n = N + 3                                                        # (n % 4 != 0: the tail loop)
    ps = [ctx.Process(target=_worker, args=(r, 2, 29547 + int(absent), n, q, absent)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    nit = 2 if absent else 3
    for it in range(nit):
        want = np.zeros(n, np.float32)
        for r in range(2):
            want = want + np.random.RandomState(100 * it + r).randn(n).astype(np.float32)    # rank order, fp32
        for r in range(2):
            st, got = res[r][it]
            assert st == 0
            assert np.array_equal(got, want), f"minibatch {it}, rank {r}"
    if absent:                                                       # rank 0 waited 300 ms for rank 1 and says which phase expired
        st, _ = res[0][2]
        assert st == 0x80000000
        assert len(res[1]) == 2


def _dp_worker(rank, world, port, q):
    """DataParallelLstm(oneshot=True) at 40/800/512, 2 ranks x 2 streams on ONE GPU, 3 minibatches.  Launch-per-step chain: the
    persistent launches of two PROCESSES sharing one GPU could start interleaved (2 x 200 workgroups do not fit), which is not what
    is under test here."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import kaldi_lstm_amd as k
    from oracle.oracle import make_params
    from kaldi_lstm_amd import shard_time_major
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    I, C, R, S, T = 40, 800, 512, 4, 20
    e = k.Engine(I, C, R, S // world, stream=torch.cuda.Stream())
    e.set_params(make_params(I, C, R, scale=0.01, seed=5))
    e.set_option("persist", 0)
    dp = k.DataParallelLstm(e, oneshot=True)
    assert dp.oneshot is not None and "oneshot" in dp.collective_name
    rng = np.random.RandomState(6)
    out = torch.empty(T * S // world, R, device="cuda"); ind = torch.empty(T * S // world, I, device="cuda")
    for it in range(3):
        x = rng.randn(T * S, I).astype(np.float32); od = (0.1 * rng.randn(T * S, R)).astype(np.float32)
        xs = shard_time_major(torch.from_numpy(x), S, rank, world).contiguous().cuda()
        ods = shard_time_major(torch.from_numpy(od), S, rank, world).contiguous().cuda()
        dist.barrier()
        dp.train_step(xs, out, ods, ind, 0.9, 1e-3, reset_flags=[1] * (S // world) if it == 0 else None)
        e.synchronize()
    assert dp.oneshot.status() == 0
    q.put((rank, e.get_params(), e.get_corr()))
    dist.barrier()
    e.close()
    dist.destroy_process_group()


def test_data_parallel_step_through_the_oneshot_exchange():
    """Two ranks (processes) x 2 streams on one GPU through DataParallelLstm(oneshot=True) against ONE engine with all 4 streams:
    same parameters and momentum after three minibatches up to fp32 summation order (the gradient of 4 streams is the sum of two
    2-stream gradients), and bit-identical replicas."""
    import torch.multiprocessing as mp
    import kaldi_lstm_amd as k
    from oracle.oracle import make_params
    from tests.test_engine_gpu import check_blob, dev
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, 29551, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict((r, (pp, cc)) for r, pp, cc in (q.get(timeout=240) for _ in range(2)))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    I, C, R, S, T = 40, 800, 512, 4, 20
    e = k.Engine(I, C, R, S); e.set_params(make_params(I, C, R, scale=0.01, seed=5))
    rng = np.random.RandomState(6)
    out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
    for it in range(3):
        x = rng.randn(T * S, I).astype(np.float32); od = (0.1 * rng.randn(T * S, R)).astype(np.float32)
        if it == 0:
            e.reset([1] * S)
        e.propagate(dev(x), out); e.backpropagate(dev(x), dev(od), ind, momentum=0.9); e.update(1e-3)
    e.synchronize()
    check_blob(res[0][0], e.get_params(), 2e-5, C, R, "params: 2 ranks x 2 streams vs 4 streams")
    check_blob(res[0][1], e.get_corr(), 2e-4, C, R, "momentum: 2 ranks x 2 streams vs 4 streams")
    e.close()
