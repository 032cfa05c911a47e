"""The persistent (weights-resident) chain behind Kaldi: what happens when its co-residency assumption fails, and that it
does not care what else runs on the chip.  (VERDICT r02 "make the persistent chain safe to ship"; ADVICE r02 medium:
a give-up must not reach the parameters.)

* forced give-up (test hook: workgroup 0 withholds one publish, every sweep of that step expires): KLSTM_ERR_HIP at the
  next call that looks, momentum and parameters UNTOUCHED by that minibatch's Update, the engine continues on the
  launch-per-step chain with oracle parity on the following minibatches;
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


@pytest.mark.parametrize("direction", ["bwd", "fwd"])
@pytest.mark.parametrize("I,C,R,S,T", [(40, 64, 32, 4, 8), (40, 800, 512, 4, 20), (40, 800, 512, 8, 20)])
def test_forced_give_up_reports_gates_the_update_and_falls_back(I, C, R, S, T, direction):
    import kaldi_lstm_amd as k
    big = C > 200
    scale, lr, od_scale = (0.01, 1e-5, 0.1) if big else (0.2, 1e-3, 1.0)
    p = make_params(I, C, R, scale=scale, seed=71)
    rng = np.random.RandomState(72)
    o = Oracle(I, C, R, S, np.float32); o.set_params(p)
    e = k.Engine(I, C, R, S); e.set_params(p)
    e.set_option("persist", 2)
    e.set_option("persist_spin_us", 3000)             # a wait gives up after 3 ms instead of 50
    out = torch.empty(T * S, R, device="cuda"); idf = torch.empty(T * S, I, device="cuda")

    def both(x, od):
        xd, odd = dev(x), dev(od)
        e.propagate(xd, out); e.backpropagate(xd, odd, idf, momentum=0.9, flags=2); e.update(lr); e.synchronize()
        out_o = o.propagate(x); id_o = o.backpropagate(x, od, momentum=0.9); o.update(lr)
        assert relerr(out.cpu().numpy(), out_o) <= 3e-5
        assert relerr(idf.cpu().numpy(), id_o) <= 3e-4
        check_blob(e.get_corr(), o.get_corr(), 3e-4, C, R, "corr")
        check_blob(e.get_params(), o.get_params(), 3e-5, C, R, "params")

    both(*_minibatch(rng, I, R, S, T, od_scale))      # 0: the persistent chain, healthy
    e.set_option("profile", 1)
    x, od = _minibatch(rng, I, R, S, T, od_scale)
    xd, odd = dev(x), dev(od)
    e.propagate(xd, out); e.synchronize()
    assert e.profile_query("k_fwd_persist")[1] == 1   # (it IS the persistent path that is about to be broken)
    e.set_option("profile", 0)
    params_before, corr_before = e.get_params(), e.get_corr()
    o.propagate(x)                                    # the oracle only advances its state over this minibatch

    # 1: the give-up.  Whichever call looks first reports it; none of them may touch momentum or parameters.
    e.set_option("persist_test_stall_" + direction, 3)
    x1, od1 = _minibatch(rng, I, R, S, T, od_scale)
    x1d, od1d = dev(x1), dev(od1)
    err = None
    try:
        e.propagate(x1d, out); e.backpropagate(x1d, od1d, idf, momentum=0.9, flags=2); e.update(lr); e.synchronize()
    except k.KlstmError as ex:
        err = ex
    assert err is not None and err.status == KLSTM_ERR_HIP and "timed out" in str(err)
    e.synchronize()                                   # (reported once; the engine is usable again)
    assert np.array_equal(e.get_params(), params_before), "a minibatch whose chain gave up reached the parameters"
    assert np.array_equal(e.get_corr(), corr_before), "a minibatch whose chain gave up reached the momentum buffers"
    e.set_option("persist_test_stall_" + direction, 0)

    # 2, 3: on from a Reset (the carried state of the broken minibatch is invalid), now on the launch-per-step chain
    e.reset([1] * S); o.reset([1] * S)
    e.set_option("profile", 1)
    for _ in range(2):
        both(*_minibatch(rng, I, R, S, T, od_scale))
    assert e.profile_query("k_fwd_persist")[1] == 0 and e.profile_query("k_bwd_persist")[1] == 0
    e.close()


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
