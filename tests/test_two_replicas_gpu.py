"""Two HIP replicas before the first 8-GPU run (VERDICT r05 missing #3): two PROCESSES on the one GPU of the box, each with its own
klstm engine over 4 of 8 streams, exchange their gradient blobs once per minibatch and must stay BIT-identical to each other and within
2e-5 of ONE engine that sees all 8 streams -- the data-parallel order of DESIGN.md 8 (gradient with KLSTM_BPTT_DEFER_MOMENTUM -> sum over
the ranks -> klstm_apply_momentum -> klstm_update) on the real kernels, not on the CPU twins of tests/test_dp_gloo.py.

Exchange: (a) "gloo": the caller's own all-reduce of klstm_grad_blob() through a host copy (what an MPI-based Kaldi trainer would do;
klstm.h: such a caller sets "persist_verify" = 1, a give-up is answered inside the call that launched); (b) "oneshot": the two-process
one-shot exchange over hipIpc-mapped blobs with the validity word riding behind the gradient (klstm_allreduce_grads_oneshot).
Chains: "launch" = "persist" 0; "persist" = the default weights-resident launches -- two processes' 225..250-workgroup launches on one
256-CU chip collide, so give-ups are ALLOWED and answered (run again one launch per step / left out by both replicas).
The last case forces a give-up on replica 1 only (persist_test_stall_bwd): its validity word goes out as 1, BOTH replicas leave that
Update out (dp_updates_left_out = 1 on both, parameters unchanged across that minibatch, still bit-identical).

One pair of processes runs all cases (a fresh `import torch` per process costs more than the cases themselves)."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = {"small": (40, 64, 32, 4, 8), "c2": (40, 800, 512, 4, 20)}
CASES = [("small", "launch", "gloo", False), ("small", "launch", "oneshot", False), ("c2", "launch", "gloo", False),
         ("c2", "launch", "oneshot", False), ("c2", "persist", "gloo", False), ("c2", "persist", "oneshot", False),
         ("small", "persist", "oneshot", False), ("c2", "stall", "oneshot", True)]
NMB, LR, MMT = 3, 1e-3, 0.9


def _minibatches(I, R, S2, T, seed):
    rng = np.random.RandomState(seed)
    return [(rng.randn(T * S2, I).astype(np.float32), (0.1 * rng.randn(T * S2, R)).astype(np.float32)) for _ in range(NMB)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import kaldi_lstm_amd as k
    from oracle.oracle import make_params
    from kaldi_lstm_amd import shard_time_major
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    results = []
    for ci, (shape, chain, exchange, stall) in enumerate(CASES):
        I, C, R, S, T = SHAPES[shape]
        stream = torch.cuda.Stream()
        e = k.Engine(I, C, R, S, stream=stream)
        e.set_params(make_params(I, C, R, scale=0.02, seed=50 + ci))
        if chain == "launch":
            e.set_option("persist", 0)
        else:
            e.set_option("persist", 2 if chain == "persist" else (2 if rank == 1 else 0))   # "stall": replica 0 never gives up by itself
            e.set_option("persist_spin_us", 3000); e.set_option("persist_cooldown", 1)
        if exchange == "gloo":
            e.set_option("persist_verify", 1)         # klstm.h: a caller with its own all-reduce over num_params floats
        one = None
        if exchange == "oneshot":
            full = e.grad_blob_tensor(full=True)       # gradient + validity word
            one = k.OneshotAllreduce(full)
            hs = [None] * world
            dist.all_gather_object(hs, one.export())
            one.connect(rank, world, hs)
        out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
        snaps, gus = [], []
        for it, (x, od) in enumerate(_minibatches(I, R, 2 * S, T, 60 + ci)):
            xs = shard_time_major(torch.from_numpy(x), 2 * S, rank, world).contiguous().cuda()
            ods = shard_time_major(torch.from_numpy(od), 2 * S, rank, world).contiguous().cuda()
            if stall:
                e.set_option("persist_test_stall_bwd", 5 if (rank == 1 and it == 1) else 0)
                # (the other minibatches: a wait long enough to sit out the other replica's kernels -- a give-up of its OWN in minibatch 0
                #  would put minibatch 1 on the launch-per-step chain (cool-down), where the forced stall has nothing to stall)
                e.set_option("persist_spin_us", 3000 if it == 1 else 300000)
            torch.cuda.synchronize()
            dist.barrier()
            with torch.cuda.stream(stream):
                if it == 0:
                    e.reset([1] * S)
                e.propagate(xs, out)
                e.backpropagate(xs, ods, ind, MMT, 1)              # KLSTM_BPTT_DEFER_MOMENTUM: the pure gradient into the blob
                if one is not None:
                    one.allreduce_engine(e, timeout_ms=10000)
                else:
                    blob = e.grad_blob_tensor()
                    host = blob.cpu()                              # (synchronises the engine's stream: torch's current stream here)
                    dist.all_reduce(host)
                    blob.copy_(host)
                e.apply_momentum(MMT)
                e.update(LR)
            e.synchronize()
            snaps.append(e.get_params())
            gus.append(e.profile_query("persist_giveups")[1])          # (this replica's give-ups so far: which minibatch each one belongs to)
        results.append(dict(params=snaps, corr=e.get_corr(), status=(one.status() if one is not None else 0), gus=gus,
                            giveups=e.profile_query("persist_giveups")[1], left_out=e.profile_query("dp_updates_left_out")[1],
                            dropped=e.profile_query("persist_dropped")[1]))
        dist.barrier()
        if one is not None:
            one.close()
        e.close()
    q.put((rank, results))
    dist.barrier()
    dist.destroy_process_group()


def test_two_hip_replicas_stay_bit_identical_and_match_one_engine_with_all_streams():
    import torch.multiprocessing as mp
    import kaldi_lstm_amd as k
    from oracle.oracle import make_params
    from tests.test_engine_gpu import check_blob, dev
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, 29561, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    for ci, (shape, chain, exchange, stall) in enumerate(CASES):
        tag = f"case {ci} ({shape}, {chain}, {exchange}{', forced give-up on replica 1' if stall else ''})"
        a, b = res[0][ci], res[1][ci]
        assert a["status"] == 0 and b["status"] == 0, tag
        for it in range(NMB):                         # the replicas: bit-identical after every minibatch
            assert np.array_equal(a["params"][it], b["params"][it]), f"{tag}: parameters differ after minibatch {it}"
        assert np.array_equal(a["corr"], b["corr"]), f"{tag}: momentum differs"
        assert a["left_out"] == b["left_out"], f"{tag}: the replicas left out different numbers of Updates ({a['left_out']}, {b['left_out']})"
        assert a["dropped"] == 0 and b["dropped"] == 0 or exchange == "oneshot", tag
        if chain == "launch":
            assert a["giveups"] == 0 and b["giveups"] == 0 and a["left_out"] == 0
        if exchange == "gloo":
            assert a["left_out"] == 0                 # (persist_verify = 1: every give-up is answered inside the call that launched)
        if stall:
            assert b["giveups"] >= 1 and a["left_out"] >= 1
            # every left-out Update left the parameters of BOTH replicas where they were, every other one moved them (the two processes
            # share one GPU: replica 1 may give up on its own in another minibatch too -- then that one is left out as well)
            I_, C_, R_, _, _ = SHAPES[shape]
            prev, unchanged = make_params(I_, C_, R_, scale=0.02, seed=50 + ci).astype(np.float32), []
            for it in range(NMB):
                unchanged.append(bool(np.array_equal(a["params"][it], prev)))
                prev = a["params"][it]
            assert sum(unchanged) == a["left_out"], f"{tag}: {a['left_out']} Updates left out, parameters unchanged across {unchanged}"
            # ... and they are exactly the minibatches in which replica 1 gave up: the forced one (minibatch 1) -- unless a give-up of its
            # own in minibatch 0 put minibatch 1 on the launch-per-step chain (cool-down), where the forced stall has nothing to stall
            gave_up = [b["gus"][it] > (b["gus"][it - 1] if it else 0) for it in range(NMB)]
            assert unchanged == gave_up, f"{tag}: give-ups of replica 1 in minibatches {gave_up}, parameters unchanged across {unchanged}"
            if b["giveups"] == 1 and not gave_up[0]:  # (only the forced one: it was minibatch 1)
                assert unchanged == [False, True, False], f"{tag}: the Update of the failed minibatch was applied ({unchanged})"
        if a["left_out"]:
            continue                                  # (a left-out Update: the one-engine twin below sees every minibatch)
        I, C, R, S, T = SHAPES[shape]
        e = k.Engine(I, C, R, 2 * S); e.set_params(make_params(I, C, R, scale=0.02, seed=50 + ci))
        out = torch.empty(T * 2 * S, R, device="cuda"); ind = torch.empty(T * 2 * S, I, device="cuda")
        for it, (x, od) in enumerate(_minibatches(I, R, 2 * S, T, 60 + ci)):
            if it == 0:
                e.reset([1] * (2 * S))
            e.propagate(dev(x), out); e.backpropagate(dev(x), dev(od), ind, momentum=MMT); e.update(LR)
        e.synchronize()
        check_blob(a["params"][-1], e.get_params(), 2e-5, C, R, f"{tag}: params vs one 8-stream engine")
        check_blob(a["corr"], e.get_corr(), 2e-4, C, R, f"{tag}: momentum vs one 8-stream engine")
        e.close()
