"""CPU tests pinning the oracle (oracle/lstmp_oracle.c) as far as it can be pinned without
reference golden vectors (the reference has none -- SURVEY.md section 4 / 8c):
  * fp64 autograd of the forward equations (tests/ref_torch.py) vs the oracle's BPTT
  * central finite differences
  * invariants the reference's structure implies (streams == per-stream runs, chunked ==
    unchunked forward, Reset semantics, masked frames give zero gradient)
"""
import numpy as np
import pytest

from oracle.oracle import Oracle, make_params, split_blob
from tests import ref_torch

DIMS = dict(I=5, C=7, R=4, S=3)


def _data(I, C, R, S, T, seed=0, scale=0.5):
    rng = np.random.RandomState(seed)
    p = make_params(I, C, R, scale=scale, seed=seed + 1, dtype=np.float64)
    x = rng.randn(T * S, I)
    od = rng.randn(T * S, R)
    return p, x, od


def test_forward_backward_vs_autograd_fp64():
    I, C, R, S = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"]
    T = 6
    p, x, od = _data(I, C, R, S, T)
    o = Oracle(I, C, R, S, np.float64)
    o.set_params(p)
    # non-zero carried state: run one warm-up chunk first
    o.propagate(x[::-1].copy())
    st = o.get_state()
    c0, r0 = st[:, 4 * C:5 * C], st[:, 7 * C:]
    out = o.propagate(x)
    in_diff = o.backpropagate(x, od, momentum=0.0)
    g = o.get_corr()
    out_t, g_t, xg_t, cT, rT = ref_torch.grads(p, x, od, c0, r0, I, C, R, S)
    np.testing.assert_allclose(out, out_t, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(in_diff, xg_t, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(g, g_t, rtol=1e-10, atol=1e-12)
    st2 = o.get_state()
    np.testing.assert_allclose(st2[:, 4 * C:5 * C], cT, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(st2[:, 7 * C:], rT, rtol=1e-12, atol=1e-13)


def test_fp32_oracle_close_to_fp64():
    I, C, R, S, T = 40, 64, 32, 4, 20
    p, x, od = _data(I, C, R, S, T, seed=3, scale=0.1)
    o32, o64 = Oracle(I, C, R, S, np.float32), Oracle(I, C, R, S, np.float64)
    o32.set_params(p); o64.set_params(p)
    y32, y64 = o32.propagate(x), o64.propagate(x)
    d32, d64 = o32.backpropagate(x, od), o64.backpropagate(x, od)
    np.testing.assert_allclose(y32, y64, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(d32, d64, rtol=2e-3, atol=2e-5)
    g32, g64 = o32.get_corr(), o64.get_corr()
    assert np.abs(g32 - g64).max() <= 1e-4 * max(1.0, np.abs(g64).max())


def test_clip_has_identity_gradient():
    """c is clipped to +-50 with NO gradient mask (reference :296-297 vs :424-428)."""
    I, C, R, S, T = 3, 4, 2, 2, 5
    rng = np.random.RandomState(5)
    p = make_params(I, C, R, scale=0.5, seed=2, dtype=np.float64)
    o = Oracle(I, C, R, S, np.float64)
    o.set_params(p)
    st = np.zeros((S, o.W))
    st[:, 4 * C:5 * C] = 49.9 * np.sign(rng.randn(S, C))   # |c| grows past 50 -> clip active
    st[:, 7 * C:] = rng.randn(S, R)
    parts = split_blob(p.copy(), I, C, R)
    parts["bias"][C:3 * C] = 6.0          # i,f gates ~1 so that c_{t-1}*f + g*i overshoots
    p2 = np.concatenate([v.ravel() for v in parts.values()])
    o.set_params(p2)
    o.set_state(st)
    x = rng.randn(T * S, I); od = rng.randn(T * S, R)
    out = o.propagate(x)
    Y = o.prop_buf()
    assert np.abs(Y[S:(T + 1) * S, 4 * C:5 * C]).max() == 50.0      # the clip fired
    in_diff = o.backpropagate(x, od)
    out_t, g_t, xg_t, _, _ = ref_torch.grads(p2, x, od, st[:, 4 * C:5 * C], st[:, 7 * C:], I, C, R, S)
    np.testing.assert_allclose(out, out_t, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(in_diff, xg_t, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(o.get_corr(), g_t, rtol=1e-9, atol=1e-11)


def test_finite_differences_fp64():
    I, C, R, S, T = 4, 5, 3, 2, 4
    p, x, od = _data(I, C, R, S, T, seed=11)
    o = Oracle(I, C, R, S, np.float64)

    def loss(pp, xx):
        o.set_params(pp)
        o.reset(np.ones(S, np.int32))
        return float((o.propagate(xx) * od).sum())

    o.set_params(p); o.reset(np.ones(S, np.int32))
    o.propagate(x)
    in_diff = o.backpropagate(x, od)
    g = o.get_corr()
    rng = np.random.RandomState(0)
    eps = 1e-6
    for idx in rng.choice(p.size, 25, replace=False):
        pp = p.copy(); pp[idx] += eps; lp = loss(pp, x)
        pp[idx] -= 2 * eps; lm = loss(pp, x)
        assert abs((lp - lm) / (2 * eps) - g[idx]) <= 1e-6 * max(1.0, abs(g[idx]))
    for idx in rng.choice(x.size, 10, replace=False):
        xx = x.copy().ravel(); xx[idx] += eps; lp = loss(p, xx.reshape(x.shape))
        xx[idx] -= 2 * eps; lm = loss(p, xx.reshape(x.shape))
        assert abs((lp - lm) / (2 * eps) - in_diff.ravel()[idx]) <= 1e-6


def test_streams_equal_independent_single_streams():
    """Streams never mix except in the gradient row-sums (reference :263-312, :468-487)."""
    I, C, R, S, T = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"], 6
    p, x, od = _data(I, C, R, S, T, seed=4)
    o = Oracle(I, C, R, S, np.float64); o.set_params(p)
    out = o.propagate(x); ind = o.backpropagate(x, od); g = o.get_corr()
    gsum = np.zeros_like(g)
    for s in range(S):
        o1 = Oracle(I, C, R, 1, np.float64); o1.set_params(p)
        xs, ods = x[s::S].copy(), od[s::S].copy()
        np.testing.assert_allclose(o1.propagate(xs), out[s::S], rtol=0, atol=1e-14)
        np.testing.assert_allclose(o1.backpropagate(xs, ods), ind[s::S], rtol=0, atol=1e-13)
        gsum += o1.get_corr()
    np.testing.assert_allclose(gsum, g, rtol=1e-11, atol=1e-13)


def test_chunked_forward_equals_unchunked():
    """State is bridged across BPTT batches (reference :231, :331)."""
    I, C, R, S = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"]
    p, x, _ = _data(I, C, R, S, 12, seed=6)
    a = Oracle(I, C, R, S, np.float32); a.set_params(p)
    b = Oracle(I, C, R, S, np.float32); b.set_params(p)
    full = a.propagate(x)
    parts = [b.propagate(x[k * 4 * S:(k + 1) * 4 * S]) for k in range(3)]
    np.testing.assert_array_equal(full, np.concatenate(parts, 0))      # bit-exact


def test_reset_zeroes_only_flagged_streams():
    I, C, R, S = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"]
    p, x, _ = _data(I, C, R, S, 4, seed=8)
    o = Oracle(I, C, R, S, np.float32); o.set_params(p)
    o.propagate(x)
    before = o.get_state()
    o.reset([0, 1, 0])
    after = o.get_state()
    assert np.all(after[1] == 0) and np.array_equal(after[0], before[0]) and np.array_equal(after[2], before[2])
    with pytest.raises(ValueError):
        o.reset([1, 0])
    with pytest.raises(ValueError):
        o.propagate(x[:S + 1])


def test_masked_tail_frames_contribute_zero_gradient():
    """Padded frames have out_diff rows = 0 (EvalMasked, nnet-loss.cc:104-107); if a stream's
    padding is a suffix of the batch, its padded frames add exactly nothing."""
    I, C, R, S, T = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"], 6
    p, x, od = _data(I, C, R, S, T, seed=9)
    od = od.copy()
    od.reshape(T, S, R)[3:, 1, :] = 0.0          # stream 1: frames 3.. are padding
    o = Oracle(I, C, R, S, np.float64); o.set_params(p)
    o.propagate(x); o.backpropagate(x, od); g_full = o.get_corr()
    # same thing with stream 1 truncated to 3 frames, run as its own S=1 layer
    gsum = np.zeros_like(g_full)
    for s in range(S):
        o1 = Oracle(I, C, R, 1, np.float64); o1.set_params(p)
        Ts = 3 if s == 1 else T
        xs, ods = x[s::S][:Ts].copy(), od[s::S][:Ts].copy()
        o1.propagate(xs); o1.backpropagate(xs, ods); gsum += o1.get_corr()
    np.testing.assert_allclose(gsum, g_full, rtol=1e-11, atol=1e-13)


def test_momentum_and_update():
    I, C, R, S, T = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"], 4
    p, x, od = _data(I, C, R, S, T, seed=10)
    o = Oracle(I, C, R, S, np.float64); o.set_params(p)
    o.propagate(x); o.backpropagate(x, od, momentum=0.0); g1 = o.get_corr()
    o.set_state(np.zeros((S, o.W)))
    o.propagate(x); o.backpropagate(x, od, momentum=0.9); g2 = o.get_corr()
    np.testing.assert_allclose(g2, 0.9 * g1 + g1, rtol=1e-12, atol=1e-14)   # corr = mmt*corr + grad (:465-487)
    o.update(1e-3)
    np.testing.assert_allclose(o.get_params(), p - 1e-3 * g2, rtol=1e-13, atol=1e-15)  # :501-512
    # standard/ variant: corr clipped in place to +-thres before the step (standard/...:480-505)
    o.set_corr(g2 * 1e4)
    o.update(1e-3, clip_grad=50.0)
    c = o.get_corr()
    assert c.max() <= 50.0 and c.min() >= -50.0


def test_eval_masked_posterior_restatement():
    """oracle/components.py: the general-posterior restatement of Xent::EvalMasked (nnet-loss.cc:76-142) reduces to the
    one-hot one, repeated pdfs add up (:95), the target entropy of a soft posterior is -sum t log t, and an empty frame has
    target arg-max 0 (FindRowMaxId on an all-zero row)."""
    from oracle import components as oc
    rng = np.random.RandomState(0)
    y = oc.softmax(rng.randn(6, 11).astype(np.float32))
    tg = rng.randint(0, 11, 6); m = (rng.rand(6) > 0.3).astype(np.float32)
    a = oc.xent_eval_masked(y, tg, m)
    b = oc.xent_eval_masked_post(y, [[(int(t), 1.0)] for t in tg], m)
    assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    post = [[(2, 0.25), (4, 0.5), (2, 0.25)]] + [[] for _ in range(5)]
    m1 = np.ones(6, np.float32)
    d, xe, ent, correct, valid = oc.xent_eval_masked_post(y, post, m1)
    assert np.isclose(d[0, 2], y[0, 2] - 0.5) and np.isclose(d[0, 4], y[0, 4] - 0.5)
    assert np.isclose(ent, -2 * 0.5 * np.log(0.5), rtol=1e-6)
    assert np.isclose(xe, -0.5 * (np.log(y[0, 2]) + np.log(y[0, 4])), rtol=1e-6)
    assert correct == int(y[0].argmax() == 2) + int(sum(y[r].argmax() == 0 for r in range(1, 6))) and valid == 6


def test_against_torch_nn_lstm_with_projection():
    """An INDEPENDENT implementation of the same cell: torch.nn.LSTM(proj_size=R) is the LSTMP of Sak et al. without
    peepholes -- c' = f*c + i*g, h = o*tanh(c'), r = W_hr h -- with gate rows ordered i,f,g,o and two bias vectors.  With the
    three peephole vectors at zero and activations far from the +-50 cell clip the reference layer (...streams.h:222-332,
    :334-499) computes exactly that; streams are the batch dimension, a fresh engine starts from zero state like
    torch's default (h0, c0).  Forward outputs, the carried state, the input gradient and all weight gradients agree in
    fp64.  (Not a pin of the peephole / clip / multi-stream-reset semantics: those have no counterpart in torch.)"""
    import torch
    I, C, R, S, T = 5, 12, 7, 3, 9
    rng = np.random.RandomState(5)
    p = make_params(I, C, R, scale=0.5, seed=9, dtype=np.float64)
    blob = split_blob(p, I, C, R)
    for name in ("peephole_i_c", "peephole_f_c", "peephole_o_c"):
        blob[name][:] = 0.0                       # (views into p)
    x = rng.randn(T * S, I)
    od = rng.randn(T * S, R)

    o = Oracle(I, C, R, S, np.float64)
    o.set_params(p)
    out = o.propagate(x)
    in_diff = o.backpropagate(x, od, momentum=0.0)
    g = split_blob(o.get_corr(), I, C, R)
    st = o.get_state()

    lstm = torch.nn.LSTM(input_size=I, hidden_size=C, num_layers=1, bias=True, batch_first=False, proj_size=R).double()
    perm = np.concatenate([np.arange(C, 2 * C), np.arange(2 * C, 3 * C), np.arange(0, C), np.arange(3 * C, 4 * C)])   # torch rows i,f,g,o <- ours g,i,f,o
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.from_numpy(blob["w_gifo_x"][perm]))
        lstm.weight_hh_l0.copy_(torch.from_numpy(blob["w_gifo_r"][perm]))
        lstm.bias_ih_l0.copy_(torch.from_numpy(blob["bias"][perm]))
        lstm.bias_hh_l0.zero_()
        lstm.weight_hr_l0.copy_(torch.from_numpy(blob["w_r_m"]))
    xt = torch.from_numpy(x.reshape(T, S, I)).clone().requires_grad_(True)      # time-major rows t*S + s = (seq, batch)
    yt, (hT, cT) = lstm(xt)
    (yt * torch.from_numpy(od.reshape(T, S, R))).sum().backward()

    np.testing.assert_allclose(out, yt.detach().numpy().reshape(T * S, R), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(st[:, 4 * C:5 * C], cT[0].detach().numpy(), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(st[:, 7 * C:], hT[0].detach().numpy(), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(in_diff, xt.grad.numpy().reshape(T * S, I), rtol=1e-9, atol=1e-11)
    inv = np.argsort(perm)
    np.testing.assert_allclose(g["w_gifo_x"], lstm.weight_ih_l0.grad.numpy()[inv], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(g["w_gifo_r"], lstm.weight_hh_l0.grad.numpy()[inv], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(g["bias"], lstm.bias_ih_l0.grad.numpy()[inv], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(g["w_r_m"], lstm.weight_hr_l0.grad.numpy(), rtol=1e-9, atol=1e-11)
