"""The persistent (weights-resident) chain behind Kaldi: what happens when its co-residency assumption fails, and that it
does not care what else runs on the chip.  (VERDICT r02 "make the persistent chain safe to ship"; ADVICE r02 medium:
a give-up must not reach the parameters.)

* forced give-up (test hook: workgroup 0 withholds one publish, every sweep of that step expires): no error -- the minibatch is
  run again on the launch-per-step chain by the first call that looks (bit-identical to an engine that never used the
  persistent chain), a cool-down on that chain follows, then the persistent chain again; with "persist_verify" inside the very
  call; a minibatch whose buffers the caller has taken back is dropped and counted;
* uneven load: 40-56 compute units held by a foreign kernel while the chain runs -- results bit-identical to the idle run
  (MI355X_MICROARCH.md: "test every hand-off under uneven load");
* fewer compute units than workgroups: the engine keeps to one launch per step and says why;
* the chain next to asynchronous RCCL all-reduces (DataParallelNnet(overlap=True)).
"""
import numpy as np
import pytest
import torch

from oracle.oracle import Oracle, make_params
from tests.test_engine_gpu import check_blob, dev, relerr, _stacked_net_against_cpu_twins

pytestmark = pytest.mark.gpu

KLSTM_ERR_HIP = 4


def _minibatch(rng, I, R, S, T, od_scale):
    x = rng.randn(T * S, I).astype(np.float32)
    od = (od_scale * rng.randn(T * S, R)).astype(np.float32)
    return x, od


def _step(e, xd, odd, out, idf, lr):
    e.propagate(xd, out); e.backpropagate(xd, odd, idf, momentum=0.9, flags=2); e.update(lr)


def _snapshot(e, out, idf):
    e.synchronize()
    c, r = e.get_state()
    return dict(out=out.cpu().numpy(), in_diff=idf.cpu().numpy(), params=e.get_params(), corr=e.get_corr(), c=c, r=r)


def _same(a, b, what):
    for key in a:
        assert np.array_equal(a[key], b[key]), f"{what}: {key} differs from the engine that never used the persistent chain"


@pytest.mark.parametrize("direction", ["bwd", "fwd"])
@pytest.mark.parametrize("I,C,R,S,T", [(40, 64, 32, 4, 8), (40, 800, 512, 4, 20), (40, 800, 512, 8, 20)])
def test_forced_give_up_is_answered_by_running_the_minibatch_again(I, C, R, S, T, direction):
    """Workgroup 0 withholds one publish: every sweep of that step expires, the launch gives up, everything queued behind it does
    nothing (device-side guard), and the first call that looks -- here the synchronisation after the Update -- runs the minibatch
    again on the launch-per-step chain: no error, no lost Update.  forward give-up: out, in_diff, parameters, momentum and carried
    state BIT-IDENTICAL to a twin engine that never used the persistent chain (the whole minibatch is re-run from the untouched
    state); backward give-up: the forward launch was good and stays (its planes differ from the twin's in the last bits), BPTT +
    Update are re-run: oracle tolerances.  Then `persist_cooldown` minibatches on the launch-per-step chain (bit-identical to the
    twin in the forward case), then the persistent chain again."""
    import kaldi_lstm_amd as k
    big = C > 200
    scale, lr, od_scale = (0.01, 1e-5, 0.1) if big else (0.2, 1e-3, 1.0)
    p = make_params(I, C, R, scale=scale, seed=71)
    rng = np.random.RandomState(72)
    o = Oracle(I, C, R, S, np.float32); o.set_params(p)
    e = k.Engine(I, C, R, S); e.set_params(p)
    t = k.Engine(I, C, R, S); t.set_params(p); t.set_option("persist", 0)     # the twin: launch-per-step chain only
    e.set_option("persist", 2)
    e.set_option("persist_spin_us", 3000)             # a wait gives up after 3 ms instead of 50
    e.set_option("persist_cooldown", 2)
    e.set_option("profile", 1)                        # (counts launches per kernel)
    out = torch.empty(T * S, R, device="cuda"); idf = torch.empty(T * S, I, device="cuda")
    out_t = torch.empty(T * S, R, device="cuda"); idf_t = torch.empty(T * S, I, device="cuda")

    def vs_oracle(x, od, tol_out=3e-5, tol_grad=3e-4):
        out_o = o.propagate(x); id_o = o.backpropagate(x, od, momentum=0.9); o.update(lr)
        assert relerr(out.cpu().numpy(), out_o) <= tol_out and relerr(idf.cpu().numpy(), id_o) <= tol_grad
        check_blob(e.get_corr(), o.get_corr(), tol_grad, C, R, "corr")
        check_blob(e.get_params(), o.get_params(), tol_out, C, R, "params")

    # 1: the give-up, on the very first minibatch (both engines start from the same parameters, state and momentum)
    e.set_option("persist_test_stall_" + direction, 3)
    x, od = _minibatch(rng, I, R, S, T, od_scale)
    xd, odd = dev(x), dev(od)
    _step(e, xd, odd, out, idf, lr)                   # no exception, here or below
    got = _snapshot(e, out, idf)
    e.set_option("persist_test_stall_" + direction, 0)
    assert e.profile_query("persist_giveups")[1] == 1 and e.profile_query("persist_replayed")[1] == 1
    assert e.profile_query("persist_dropped")[1] == 0
    assert e.profile_query("k_fwd_persist")[1] == 1   # (it WAS the persistent path)
    assert b"run again" in k.load_library().klstm_last_error()
    _step(t, xd, odd, out_t, idf_t, lr)
    want = _snapshot(t, out_t, idf_t)
    if direction == "fwd":
        _same(got, want, "minibatch of the give-up")
    vs_oracle(x, od)
    # 2, 3: the cool-down, on the launch-per-step chain
    for i in range(2):
        x, od = _minibatch(rng, I, R, S, T, od_scale)
        xd, odd = dev(x), dev(od)
        _step(e, xd, odd, out, idf, lr); got = _snapshot(e, out, idf)
        _step(t, xd, odd, out_t, idf_t, lr); want = _snapshot(t, out_t, idf_t)
        if direction == "fwd":
            _same(got, want, f"cool-down minibatch {i}")
        vs_oracle(x, od)
    assert e.profile_query("k_fwd_persist")[1] == 1
    # 4, 5: back on the persistent chain
    for i in range(2):
        x, od = _minibatch(rng, I, R, S, T, od_scale)
        xd, odd = dev(x), dev(od)
        _step(e, xd, odd, out, idf, lr); e.synchronize()
        vs_oracle(x, od)
    assert e.profile_query("k_fwd_persist")[1] == 3 and e.profile_query("k_bwd_persist")[1] >= 2
    assert e.profile_query("persist_giveups")[1] == 1
    e.close(); t.close()


@pytest.mark.parametrize("flags", [0, 2])
@pytest.mark.parametrize("direction", ["bwd", "fwd"])
def test_persist_verify_answers_the_give_up_inside_the_call(direction, flags):
    """Option "persist_verify" = 1 (what the Kaldi adapter of INTEGRATION.md sets): the call waits for its persistent launch, so
    `out` is right when klstm_propagate returns and `in_diff` when klstm_backpropagate returns -- before any neighbour of the
    component has read them -- whatever the launch did.  With KLSTM_BPTT_FUSE_UPDATE (flags = 2: "klstm_update follows
    immediately", Kaldi's Component::Backpropagate) the second wait sits at the end of that klstm_update: `in_diff`, parameters
    and momentum are right when IT returns, the give-up of the BPTT launch answered there (both calls run again)."""
    import kaldi_lstm_amd as k
    I, C, R, S, T = 40, 800, 512, 4, 20
    p = make_params(I, C, R, scale=0.01, seed=75)
    rng = np.random.RandomState(76)
    o = Oracle(I, C, R, S, np.float32); o.set_params(p)
    e = k.Engine(I, C, R, S); e.set_params(p)
    e.set_option("persist", 2); e.set_option("persist_spin_us", 3000); e.set_option("persist_verify", 1)
    e.set_option("persist_test_stall_" + direction, 5)
    x, od = _minibatch(rng, I, R, S, T, 0.1)
    xd, odd = dev(x), dev(od)
    out = torch.empty(T * S, R, device="cuda"); idf = torch.empty(T * S, I, device="cuda")
    e.propagate(xd, out)
    torch.cuda.synchronize()                          # (no klstm call: what a neighbour reading `out` would see)
    assert relerr(out.cpu().numpy(), o.propagate(x)) <= 3e-5
    assert e.profile_query("persist_giveups")[1] == (1 if direction == "fwd" else 0)
    e.backpropagate(xd, odd, idf, momentum=0.9, flags=flags)
    if flags == 2:
        e.update(1e-5)
    torch.cuda.synchronize()                          # (no klstm call: what the neighbour reading `in_diff` would see)
    assert relerr(idf.cpu().numpy(), o.backpropagate(x, od, momentum=0.9)) <= 3e-4
    assert e.profile_query("persist_giveups")[1] == 1 and e.profile_query("persist_replayed")[1] == 1
    assert e.profile_query("persist_dropped")[1] == 0
    if flags != 2:
        e.update(1e-5)
    o.update(1e-5)
    check_blob(e.get_params(), o.get_params(), 3e-5, C, R, "params")
    check_blob(e.get_corr(), o.get_corr(), 3e-4, C, R, "corr")
    # the promise broken: backpropagate(FUSE_UPDATE) and no klstm_update -- the next propagate waits, answers, and the record it
    # answers with is still the old minibatch's (nothing dropped)
    if flags == 2 and direction == "bwd":
        e.set_option("persist_cooldown", 0)
        e.set_option("persist_test_stall_bwd", 7)
        x2, od2 = _minibatch(rng, I, R, S, T, 0.1)
        x2d, od2d = dev(x2), dev(od2)
        e.propagate(x2d, out); e.backpropagate(x2d, od2d, idf, momentum=0.9, flags=2)
        e.set_option("persist_test_stall_bwd", 0)
        out3 = torch.empty_like(out)
        e.propagate(x2d, out3)                        # (flushes the gradient products of minibatch 2 the ordinary way, after the answer)
        torch.cuda.synchronize()
        o.propagate(x2)
        assert relerr(idf.cpu().numpy(), o.backpropagate(x2, od2, momentum=0.9)) <= 3e-4
        assert e.profile_query("persist_giveups")[1] == 2 and e.profile_query("persist_dropped")[1] == 0
        check_blob(e.get_corr(), o.get_corr(), 3e-4, C, R, "corr after the broken promise")
    e.close()


def test_give_up_found_after_the_caller_moved_on_drops_that_minibatch_only():
    """No synchronisation between minibatches (a pipelined trainer): the forward launch of minibatch 1 gives up; by the time anybody
    looks, minibatch 2 has been queued behind it (and did nothing: guard) and the buffers of minibatch 1 are the caller's again.
    Minibatch 1 is dropped -- no Update, no state advance -- and counted; minibatch 2 is run again from the state minibatch 1
    started from: BIT-IDENTICAL to a twin that only ever saw minibatch 2."""
    import kaldi_lstm_amd as k
    I, C, R, S, T = 40, 800, 512, 4, 20
    p = make_params(I, C, R, scale=0.01, seed=77)
    rng = np.random.RandomState(78)
    e = k.Engine(I, C, R, S); e.set_params(p)
    t = k.Engine(I, C, R, S); t.set_params(p); t.set_option("persist", 0)
    e.set_option("persist", 2); e.set_option("persist_spin_us", 3000)
    e.set_option("persist_test_stall_fwd", 4)
    bufs = []
    for i in range(2):
        x, od = _minibatch(rng, I, R, S, T, 0.1)
        bufs.append((dev(x), dev(od), torch.empty(T * S, R, device="cuda"), torch.empty(T * S, I, device="cuda")))
    for xd, odd, out, idf in bufs:                    # both minibatches queued without anybody looking in between
        _step(e, xd, odd, out, idf, 1e-5)
    got = _snapshot(e, bufs[1][2], bufs[1][3])
    assert e.profile_query("persist_giveups")[1] == 1 and e.profile_query("persist_dropped")[1] == 1
    # (minibatch 2 was queued behind the give-up and run again -- or, if the host-mapped word was already up when its propagate
    #  began, it simply ran on the launch-per-step chain: the same bits either way)
    assert e.profile_query("persist_replayed")[1] in (0, 1)
    out_t = torch.empty(T * S, R, device="cuda"); idf_t = torch.empty(T * S, I, device="cuda")
    _step(t, bufs[1][0], bufs[1][1], out_t, idf_t, 1e-5)
    _same(got, _snapshot(t, out_t, idf_t), "the minibatch behind the dropped one")
    e.close(); t.close()


@pytest.mark.parametrize("I,C,R,S,T", [(40, 800, 512, 4, 20),
                                       (128, 520, 512, 7, 8)])     # R close to C: r(1..T) is a batched product BEHIND the forward launch
def test_give_up_two_minibatches_back_keeps_the_resets_issued_since(I, C, R, S, T):
    """ADVICE r04 (medium, twice): the host is TWO forward launches ahead when it hears of the give-up, and Reset() was called in
    between.  Minibatch 0 runs clean (the carried state is not zero any more); the forward launch of minibatch 1 gives up; Reset
    (streams 0 and 2 start new utterances) and minibatch 2 are queued behind it without anybody looking.  The state goes back to
    the buffer minibatch 1 started from -- decided by launch ordinal, two buffer flips back looks like "no change" to a parity
    test -- and the Reset issued since, which went into the abandoned buffer, is applied again: minibatch 2 must be BIT-IDENTICAL
    to a twin that ran minibatch 0, the same Reset and minibatch 2 and never saw minibatch 1.  Second shape: the product that
    writes `out` and the carried r(T) is a batched product behind the forward launch (R > C / tpw); queued behind a give-up it
    must not touch the other state buffer -- the one the state goes back to (the kernel looks at the status words)."""
    import kaldi_lstm_amd as k
    p = make_params(I, C, R, scale=0.01, seed=91)
    rng = np.random.RandomState(92)
    e = k.Engine(I, C, R, S); e.set_params(p); e.set_option("persist", 2); e.set_option("persist_spin_us", 3000)
    t = k.Engine(I, C, R, S); t.set_params(p); t.set_option("persist", 2)
    bufs = []
    for i in range(3):
        x, od = _minibatch(rng, I, R, S, T, 0.1)
        bufs.append((dev(x), dev(od), torch.empty(T * S, R, device="cuda"), torch.empty(T * S, I, device="cuda")))
    flags = [1 if s in (0, 2) else 0 for s in range(S)]
    for eng in (e, t):                                  # minibatch 0 on the persistent chain on both: identical, non-zero carried state
        eng.reset([1] * S)
        _step(eng, *bufs[0], 1e-5)
        eng.synchronize()
        assert eng.profile_query("persist_giveups")[1] == 0 and eng.profile_query("persist_launches")[1] == 2
    e.set_option("persist_test_stall_fwd", 4)
    _step(e, *bufs[1], 1e-5)                            # gives up (3 ms spin limit); nobody looks
    e.reset(flags)
    _step(e, *bufs[2], 1e-5)
    got = _snapshot(e, bufs[2][2], bufs[2][3])
    assert e.profile_query("persist_giveups")[1] == 1 and e.profile_query("persist_dropped")[1] == 1
    t.set_option("persist", 0)                          # the chain a minibatch is run again on
    t.reset(flags)
    out_t = torch.empty(T * S, R, device="cuda"); idf_t = torch.empty(T * S, I, device="cuda")
    _step(t, bufs[2][0], bufs[2][1], out_t, idf_t, 1e-5)
    want = _snapshot(t, out_t, idf_t)
    _same(got, want, "minibatch 2 behind the dropped minibatch 1, with a Reset in between")
    e.close(); t.close()


@pytest.mark.parametrize("S,hog", [(4, 40), (8, 48), (1, 48)])
def test_uneven_load_is_bit_identical_to_the_idle_chip(S, hog):
    """40/800/512: 200 workgroups exchange d_m / m through the fabric every step while `hog` compute units are held by a
    foreign kernel on another stream (each of its workgroups fills a CU: 1024 threads, 96 KB of LDS; confirmed resident,
    5-6 per XCD, before the first minibatch).  Same bits as on the idle chip, no expired wait.
    Measured limits of the co-residency assumption (tools/hog_probe.py, tools/cotenant_probe.py, profiles/r03_hog_probe.txt,
    profiles/r03_cotenant.txt): up to 48 held CUs the chain runs undisturbed, at 56 -- exactly 200 left -- a launch waits for
    the foreign kernel to end.  One erratic observation on top: while the foreign kernel's residency flags were polled with
    pageable D2H copies (tensor.cpu() in a loop), persistent launches that followed a later runtime copy started with ~30 of
    their 200 workgroups missing until the foreign kernel ended; with the flags in pinned host memory the same sequences
    (copies, syncs, gaps, allocations in between) all pass.  Not understood (runtime queue scheduling); it is what the
    bounded waits, the gated Update and the fallback (tests above) are for.  This test keeps the flags in pinned memory."""
    import time
    import kaldi_lstm_amd as k
    I, C, R, T = 40, 800, 512, 20
    p = make_params(I, C, R, scale=0.01, seed=81)
    rng = np.random.RandomState(82)
    data = [_minibatch(rng, I, R, S, T, 0.1) for _ in range(4)]
    lib = k.load_library()
    res = []
    for loaded in (False, True):
        e = k.Engine(I, C, R, S); e.set_params(p); e.set_option("persist", 2)
        keep = [(dev(x), dev(od), torch.empty(T * S, R, device="cuda"), torch.empty(T * S, I, device="cuda")) for x, od in data]
        e.set_option("profile", 1)
        e.propagate(keep[0][0], keep[0][2]); e.synchronize()   # (allocations done; and it IS the persistent path)
        assert e.profile_query("k_fwd_persist")[1] == 1
        e.set_option("profile", 0)
        e.reset([1] * S)
        where = torch.full((2 * hog,), -1, dtype=torch.int32).pin_memory()      # written by the foreign kernel, polled without a copy
        torch.cuda.synchronize()
        if loaded:
            assert lib.klstm_debug_occupy(0, hog, 60000, None, where.data_ptr()) == 0   # 60 ms: far longer than the four minibatches
            t0 = time.time()
            while (where.numpy() == -1).any():           # every workgroup of the foreign kernel is resident
                assert time.time() - t0 < 5.0
                time.sleep(0.0005)
        for xd, odd, out, idf in keep:
            e.propagate(xd, out); e.backpropagate(xd, odd, idf, momentum=0.9, flags=2); e.update(1e-5)
        e.synchronize()                                # (raises if a wait expired)
        res.append([(o_.cpu().numpy(), i_.cpu().numpy()) for _, _, o_, i_ in keep] + [(e.get_params(), e.get_corr())])
        e.close()
        torch.cuda.synchronize()
    for a, b in zip(*res):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_fewer_compute_units_than_workgroups_keeps_to_one_launch_per_step():
    import kaldi_lstm_amd as k
    I, C, R, S, T = 40, 800, 512, 4, 20
    p = make_params(I, C, R, scale=0.01, seed=91)
    rng = np.random.RandomState(92)
    o = Oracle(I, C, R, S, np.float32); o.set_params(p)
    e = k.Engine(I, C, R, S); e.set_params(p)
    e.set_option("persist_ncu", 128)                  # e.g. a partitioned MI355X: 200 workgroups cannot be co-resident
    e.set_option("persist", 2)
    e.set_option("profile", 1)
    x, od = _minibatch(rng, I, R, S, T, 0.1)
    xd, odd = dev(x), dev(od)
    out = torch.empty(T * S, R, device="cuda"); idf = torch.empty(T * S, I, device="cuda")
    e.propagate(xd, out); e.backpropagate(xd, odd, idf, momentum=0.0); e.synchronize()
    assert e.profile_query("k_fwd_persist")[1] == 0 and e.profile_query("k_bwd_persist")[1] == 0
    assert b"compute units" in k.load_library().klstm_last_error()
    assert relerr(out.cpu().numpy(), o.propagate(x)) <= 3e-5
    assert relerr(idf.cpu().numpy(), o.backpropagate(x, od, momentum=0.0)) <= 3e-4
    e.close()


def test_persistent_chain_next_to_asynchronous_allreduces():
    """DataParallelNnet(overlap=True): RCCL kernels of the upper layers' gradient slices share the chip with the 200-workgroup
    persistent launches of the layers below (two stacked 800/512 layers, T = 20: the persistent chain is the default)."""
    _stacked_net_against_cpu_twins((40, 800, 512, 2, 131), S=4, T=20, scale=0.01, lr=1e-3, nsteps=3, overlap=True, port=29538,
                                   tol_param=5e-5, tol_grad=3e-4)


@pytest.mark.parametrize("verify", [0, 1])
def test_data_parallel_step_with_a_give_up_keeps_the_replicas_together(verify):
    """The data-parallel order (gradient -> klstm_allreduce_grads -> momentum -> Update) on a 1-rank RCCL communicator, four
    minibatches queued WITHOUT a host synchronisation; the backward launch of minibatch 1 gives up.
    verify = 0 (the C-ABI's default): klstm_allreduce_grads does not wait for the device.  The gradient kernel of the failed
    minibatch writes 1 into the validity word behind the blob, the word goes through the all-reduce with the gradient, and the
    Update kernels -- of every rank -- leave that step out; so do the steps queued behind it until the host has heard of the
    give-up.  Nothing is applied twice, nothing half: the parameters equal a twin that saw minibatch 0 and the minibatches from
    the recovery on, and the left-out ones are counted ("dp_updates_left_out").
    verify = 1: every persistent call waits for its launch; the give-up is answered inside klstm_backpropagate, the all-reduce
    carries a real gradient, nothing is left out."""
    import os
    import torch.distributed as dist
    import kaldi_lstm_amd as k
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        I, C, R, S, T, N = 40, 800, 512, 4, 20, 4
        p = make_params(I, C, R, scale=0.01, seed=81)
        rng = np.random.RandomState(82)
        stream = torch.cuda.Stream()
        e = k.Engine(I, C, R, S, stream=stream); e.set_params(p)
        e.set_option("persist", 2); e.set_option("persist_spin_us", 3000); e.set_option("persist_verify", verify)
        dp = k.DataParallelLstm(e, force_collective=True)
        assert dp.collective and dp.comm is not None
        t = k.Engine(I, C, R, S); t.set_params(p); t.set_option("persist", 0)
        mbs = []
        for i in range(N):
            x, od = _minibatch(rng, I, R, S, T, 0.1)
            mbs.append((dev(x), dev(od), torch.empty(T * S, R, device="cuda"), torch.empty(T * S, I, device="cuda")))
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for i, (xd, odd, out, idf) in enumerate(mbs):
                e.set_option("persist_test_stall_bwd", 5 if i == 1 else 0)
                dp.train_step(xd, out, odd, idf, 0.9, 1e-5, reset_flags=[1] * S if i == 0 else None)
        e.synchronize()
        assert e.profile_query("persist_giveups")[1] == 1
        left_out = e.profile_query("dp_updates_left_out")[1]
        if verify:
            assert left_out == 0 and e.profile_query("persist_replayed")[1] == 1 and e.profile_query("persist_dropped")[1] == 0
        else:
            # (3 ms pass before the launch gives up; the host has queued all four minibatches long before: normally N - 1.  0 would
            #  mean the host heard of it before minibatch 1's all-reduce went out and ran its BPTT again.)
            assert 0 <= left_out <= N - 1
        # the twin: minibatch 0; the forward pass of minibatch 1 was good and advanced the state, its Update is left out; the
        # minibatches queued behind it did nothing at all (no Update, no state advance); then the rest
        out_t = torch.empty(T * S, R, device="cuda"); idf_t = torch.empty(T * S, I, device="cuda")
        for i, (xd, odd, _, _) in enumerate(mbs):
            if i == 0:
                t.reset([1] * S)
            if 1 <= i < 1 + left_out:
                if i == 1:
                    t.propagate(xd, out_t)
                continue
            t.propagate(xd, out_t); t.backpropagate(xd, odd, idf_t, momentum=0.9); t.update(1e-5)
        t.synchronize()
        assert relerr(e.get_params(), t.get_params()) <= 1e-6
        assert relerr(e.get_corr(), t.get_corr()) <= 2e-5
        ce, re_ = e.get_state(); ct, rt = t.get_state()
        assert relerr(ce, ct) <= 2e-5 and relerr(re_, rt) <= 2e-5
        if left_out < N - 1:                       # the last minibatch was applied: its outputs are the twin's
            assert relerr(mbs[-1][2].cpu().numpy(), out_t.cpu().numpy()) <= 2e-5
            assert relerr(mbs[-1][3].cpu().numpy(), idf_t.cpu().numpy()) <= 3e-4
        e.close(); t.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S", [5, 6, 8])
def test_interleaved_stream_groups_of_the_backward_launch_are_bit_identical_to_groups_in_sequence(S):
    """5..8 streams: the backward launch walks its two groups of 4 streams as two interleaved chains (k_bwd_persist2i: one group's
    d_m crosses the fabric while the other group's planes are pulled and contracted) instead of one chain after the other
    (k_bwd_persist2, option "persist_bwd_interleave" = 0).  Same instruction sequence per (cell, stream): every output, the
    parameters, the momentum and the carried state must agree to the bit, over chained minibatches."""
    import kaldi_lstm_amd as k
    I, C, R, T = 40, 800, 512, 20
    p = make_params(I, C, R, scale=0.01, seed=91)
    rng = np.random.RandomState(92)
    a = k.Engine(I, C, R, S); a.set_params(p); a.set_option("persist", 2)
    b = k.Engine(I, C, R, S); b.set_params(p); b.set_option("persist", 2); b.set_option("persist_bwd_interleave", 0)
    out_a = torch.empty(T * S, R, device="cuda"); idf_a = torch.empty(T * S, I, device="cuda")
    out_b = torch.empty(T * S, R, device="cuda"); idf_b = torch.empty(T * S, I, device="cuda")
    for i in range(3):
        x, od = _minibatch(rng, I, R, S, T, 0.1)
        xd, odd = dev(x), dev(od)
        _step(a, xd, odd, out_a, idf_a, 1e-5)
        _step(b, xd, odd, out_b, idf_b, 1e-5)
        _same(_snapshot(a, out_a, idf_a), _snapshot(b, out_b, idf_b), "minibatch %d" % i)
    a.close(); b.close()


@pytest.mark.parametrize("I,S", [(40, 5), (40, 6), (40, 8), (512, 8)])
def test_interleaved_stream_groups_of_the_forward_launch_are_bit_identical_to_groups_in_lock_step(I, S):
    """Round 6, 5..8 streams: the forward launch walks its two groups of 4 streams as two interleaved chains (one barrier per (frame,
    group): while one group's m(t) crosses the fabric the workgroup sweeps, contracts and updates the other) instead of both groups in
    lock-step ("persist_fwd_interleave" = 0, rounds 2-5).  Same instruction sequence per (cell, stream): every output, every plane, the
    parameters, the momentum and the carried state agree to the bit over chained minibatches (I = 512: the wide-input form, the x term
    from the batched product), and a Reset in between."""
    import kaldi_lstm_amd as k
    C, R, T = 800, 512, 20
    p = make_params(I, C, R, scale=0.01, seed=93)
    rng = np.random.RandomState(94)
    a = k.Engine(I, C, R, S); a.set_params(p); a.set_option("persist", 2)
    b = k.Engine(I, C, R, S); b.set_params(p); b.set_option("persist", 2); b.set_option("persist_fwd_interleave", 0)
    out_a = torch.empty(T * S, R, device="cuda"); idf_a = torch.empty(T * S, I, device="cuda")
    out_b = torch.empty(T * S, R, device="cuda"); idf_b = torch.empty(T * S, I, device="cuda")
    for i in range(3):
        x, od = _minibatch(rng, I, R, S, T, 0.1)
        xd, odd = dev(x), dev(od)
        if i == 2:
            fl = [1 if s % 3 == 0 else 0 for s in range(S)]
            a.reset(fl); b.reset(fl)
        _step(a, xd, odd, out_a, idf_a, 1e-5)
        _step(b, xd, odd, out_b, idf_b, 1e-5)
        _same(_snapshot(a, out_a, idf_a), _snapshot(b, out_b, idf_b), "minibatch %d" % i)
        assert np.array_equal(a.activations(0), b.activations(0)), "forward planes, minibatch %d" % i
    assert a.profile_query("persist_giveups")[1] == 0 and b.profile_query("persist_giveups")[1] == 0
    a.close(); b.close()


def test_cool_down_backs_off_while_the_give_ups_keep_coming():
    """A co-tenant that stays: every attempt to go back to the persistent chain gives up again (here: the test hook stays on).  The
    cool-down doubles with every give-up that follows a re-arm closely -- 2, 4, 8 minibatches with "persist_cooldown" = 2 -- instead of
    paying a spin limit and a re-run every 2 minibatches: 4 give-ups in 20 minibatches (at 0, 3, 8, 17), not 7; results are those of an
    engine that never used the persistent chain (every give-up was a forward one, answered inside the call: bit-identical)."""
    import kaldi_lstm_amd as k
    I, C, R, S, T, N = 40, 64, 32, 4, 8, 20
    p = make_params(I, C, R, scale=0.2, seed=95)
    rng = np.random.RandomState(96)
    e = k.Engine(I, C, R, S); e.set_params(p)
    t = k.Engine(I, C, R, S); t.set_params(p); t.set_option("persist", 0)
    e.set_option("persist", 2); e.set_option("persist_spin_us", 2000); e.set_option("persist_verify", 1)
    e.set_option("persist_cooldown", 2); e.set_option("persist_test_stall_fwd", 3)
    out = torch.empty(T * S, R, device="cuda"); idf = torch.empty(T * S, I, device="cuda")
    out_t = torch.empty(T * S, R, device="cuda"); idf_t = torch.empty(T * S, I, device="cuda")
    for i in range(N):
        x, od = _minibatch(rng, I, R, S, T, 1.0)
        xd, odd = dev(x), dev(od)
        _step(e, xd, odd, out, idf, 1e-3)
        _step(t, xd, odd, out_t, idf_t, 1e-3)
    _same(_snapshot(e, out, idf), _snapshot(t, out_t, idf_t), "after 20 minibatches with 4 give-ups")
    assert e.profile_query("persist_giveups")[1] == 4
    assert e.profile_query("persist_replayed")[1] == 4 and e.profile_query("persist_dropped")[1] == 0
    e.close(); t.close()
