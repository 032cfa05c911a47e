"""The nnet1 container around the hot path (include/klstm_nnet.hpp): whole-model file I/O, the
bd-nnet-train-lstm-streams training-loop workalike and the nnet-forward workalike, end to end against
oracle restatements (C LSTM oracle + numpy batcher / Affine / Softmax / Xent::EvalMasked)."""
import numpy as np
import pytest

from oracle import components as oc
from oracle.oracle import Oracle, make_params
from tests import kaldi_fmt
from tests.test_component import run, raw, _pack_utts

I, C, R, NPDF = 8, 16, 8, 11


def make_net(S, seed=0, standard=False, shift=2):
    rng = np.random.RandomState(seed)
    flat = make_params(I, C, R, scale=0.3, seed=seed + 1)
    W = (0.3 * rng.randn(NPDF, R)).astype(np.float32)
    b = (0.1 * rng.randn(NPDF)).astype(np.float32)
    if standard:      # standard/nnet.proto: TimeShift -> LstmProjected -> Affine -> Softmax
        layers = [("timeshift", I, shift), ("lstm", flat, I, C, R), ("affine", W, b), ("softmax", NPDF)]
    else:             # google/nnet.proto: Transmit -> LstmProjectedStreams -> Affine -> Softmax
        layers = [("transmit", I), ("lstm_streams", flat, I, C, R, S), ("affine", W, b), ("softmax", NPDF)]
    return layers, flat, W, b


def test_whole_nnet_file_roundtrip_text_binary(tmp_path):
    layers, flat, W, b = make_net(3)
    (tmp_path / "n.txt").write_bytes(kaldi_fmt.nnet_text(layers))
    out = run("nnet_copy", tmp_path / "n.txt", 1, tmp_path / "n.bin").stdout.split()
    assert out == ["OK", "4", "<Transmit>", "<LstmProjectedStreams>", "<AffineTransform>", "<Softmax>"]
    assert (tmp_path / "n.bin").read_bytes() == kaldi_fmt.nnet_binary(layers)         # text -> binary: byte-identical
    run("nnet_copy", tmp_path / "n.bin", 1, tmp_path / "n2.bin")
    assert (tmp_path / "n2.bin").read_bytes() == kaldi_fmt.nnet_binary(layers)
    run("nnet_copy", tmp_path / "n.bin", 0, tmp_path / "n2.txt")
    txt = (tmp_path / "n2.txt").read_text()
    assert txt.startswith("<Nnet> \n<Transmit> %d %d \n<LstmProjectedStreams> %d %d <CellDim> %d <NumStream> 3  [" % (I, I, R, I, C))
    assert "<AffineTransform> %d %d <LearnRateCoef> 1 <BiasLearnRateCoef> 1 <MaxNorm> 0  [" % (NPDF, R) in txt   # README.md:27
    assert txt.rstrip().endswith("</Nnet>")
    layers_s, *_ = make_net(1, standard=True)
    (tmp_path / "s.bin").write_bytes(kaldi_fmt.nnet_binary(layers_s))
    out = run("nnet_copy", tmp_path / "s.bin", 0, tmp_path / "s.txt").stdout.split()
    assert out[2:] == ["<TimeShift>", "<LstmProjected>", "<AffineTransform>", "<Softmax>"]
    assert "<TimeShift> %d %d <Shift> 2" % (I, I) in (tmp_path / "s.txt").read_text()


def oracle_train(flat, W, b, utts, S, T, delay, lr, mmt, crossvalidate=False):
    """bd-nnet-train-lstm-streams.cc:143-304 with the oracle pieces."""
    lstm = Oracle(I, C, R, S, np.float32); lstm.set_params(flat)
    W, b = W.copy(), b.copy(); Wc, bc = np.zeros_like(W), np.zeros_like(b)
    batcher = oc.MultiStreamBatcher(utts, S, T, delay)
    loss = correct = frames = 0.0
    nb = 0
    while True:
        nxt = batcher.next()
        if nxt is None:
            break
        feat, target, mask, flags = nxt
        lstm.reset(flags)                                              # :209
        h = lstm.propagate(oc.transmit(feat))                          # :215
        a = oc.affine_propagate(h, W, b); y = oc.softmax(a)
        diff, xe, ent, cor, valid = oc.xent_eval_masked(y, target, mask)        # :219
        loss += xe - ent; correct += cor; frames += valid
        if not crossvalidate:                                          # :227-229: last component first, Backpropagate then Update
            hd = oc.affine_backpropagate(diff, W)
            oc.affine_update(h, diff, W, b, Wc, bc, lr, lr, mmt)
            lstm.backpropagate(feat, hd, momentum=mmt); lstm.update(lr)
        nb += 1
    return dict(loss=loss / frames, acc=correct / frames, frames=frames, nb=nb, lstm=lstm.get_params(), W=W, b=b)


@pytest.mark.gpu
@pytest.mark.parametrize("crossvalidate", [0, 1])
def test_train_lstm_streams_workalike_end_to_end(tmp_path, crossvalidate):
    S, T, delay, lr, mmt = 4, 10, 3, 2e-3, 0.9
    layers, flat, W, b = make_net(S, seed=3)
    rng = np.random.RandomState(5)
    utts = [(rng.randn(n, I).astype(np.float32), rng.randint(0, NPDF, n)) for n in (37, 10, 52, 25, 8, 61, 30)]
    (tmp_path / "n.bin").write_bytes(kaldi_fmt.nnet_binary(layers))
    _pack_utts(utts).tofile(tmp_path / "u.raw")
    r = run("nnet_train", tmp_path / "n.bin", tmp_path / "u.raw", S, T, delay, lr, mmt, crossvalidate, tmp_path / "out.bin")
    head = r.stdout.splitlines()[0].split()
    exp = oracle_train(flat, W, b, utts, S, T, delay, lr, mmt, bool(crossvalidate))
    assert head[0] == "OK" and int(head[1]) == len(utts) and int(head[2]) == exp["nb"] and float(head[3]) == exp["frames"]
    assert abs(float(head[4]) - exp["loss"]) <= 2e-4 * abs(exp["loss"])               # AvgLoss (Xent)
    assert abs(float(head[5]) - exp["acc"]) <= 1.5 / exp["frames"]                    # frame accuracy (argmax ties)
    assert "FRAME_ACCURACY >>" in r.stdout and "AvgLoss:" in r.stdout                 # Xent::Report format (:293-307)
    if not crossvalidate:
        trained = [("transmit", I), ("lstm_streams", exp["lstm"], I, C, R, S), ("affine", exp["W"], exp["b"]), ("softmax", NPDF)]
        (tmp_path / "exp.bin").write_bytes(kaldi_fmt.nnet_binary(trained))
        run("nnet_copy", tmp_path / "out.bin", 0, tmp_path / "out.txt")
        run("nnet_copy", tmp_path / "exp.bin", 0, tmp_path / "exp.txt")
        got = np.array([float(v) for v in (tmp_path / "out.txt").read_text().replace("[", " ").replace("]", " ").split() if v[0] in "-0123456789."])
        want = np.array([float(v) for v in (tmp_path / "exp.txt").read_text().replace("[", " ").replace("]", " ").split() if v[0] in "-0123456789."])
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max()                   # 6-digit text + fp32 training drift
    else:
        assert not (tmp_path / "out.bin").exists()                                    # CV never writes a model (:294)


@pytest.mark.gpu
def test_nnet_forward_workalike_standard_topology(tmp_path):
    """BASELINE.json configs[0] shape of thing: TimeShift -> LstmProjected -> Affine -> Softmax, one utterance,
    Feedforward (README.md:18-30; standard/nnet.proto)."""
    layers, flat, W, b = make_net(1, seed=7, standard=True, shift=2)
    rng = np.random.RandomState(2)
    n = 57
    x = rng.randn(n, I).astype(np.float32)
    (tmp_path / "n.bin").write_bytes(kaldi_fmt.nnet_binary(layers))
    x.tofile(tmp_path / "x.raw")
    out = run("nnet_forward", tmp_path / "n.bin", tmp_path / "x.raw", n, tmp_path / "y.raw").stdout.split()
    assert out == ["OK", str(n), str(NPDF)]
    lstm = Oracle(I, C, R, 1, np.float32); lstm.set_params(flat)
    y = oc.softmax(oc.affine_propagate(lstm.propagate(oc.time_shift(x, 2)), W, b))
    got = raw(tmp_path / "y.raw").reshape(n, NPDF)
    assert np.abs(got - y).max() <= 2e-5
    np.testing.assert_allclose(got.sum(1), 1.0, rtol=1e-5)


@pytest.mark.gpu
def test_two_stacked_lstm_layers_train_end_to_end(tmp_path):
    """README.md:32-45 / BASELINE.json configs[3] topology in miniature: Transmit -> LSTM -> LSTM -> Affine -> Softmax.
    in_diff of the upper LSTM feeds the lower one; Reset fans out to both (nnet-nnet.h:132-138)."""
    S, T, delay, lr, mmt = 2, 6, 1, 5e-3, 0.5
    rng = np.random.RandomState(11)
    f1 = make_params(I, C, R, scale=0.3, seed=21); f2 = make_params(R, C, R, scale=0.3, seed=22)
    W = (0.3 * rng.randn(NPDF, R)).astype(np.float32); b = (0.1 * rng.randn(NPDF)).astype(np.float32)
    layers = [("transmit", I), ("lstm_streams", f1, I, C, R, S), ("lstm_streams", f2, R, C, R, S), ("affine", W, b), ("softmax", NPDF)]
    utts = [(rng.randn(n, I).astype(np.float32), rng.randint(0, NPDF, n)) for n in (19, 31, 12, 7)]
    (tmp_path / "n.bin").write_bytes(kaldi_fmt.nnet_binary(layers))
    _pack_utts(utts).tofile(tmp_path / "u.raw")
    r = run("nnet_train", tmp_path / "n.bin", tmp_path / "u.raw", S, T, delay, lr, mmt, 0, tmp_path / "out.bin")
    head = r.stdout.splitlines()[0].split()
    l1 = Oracle(I, C, R, S, np.float32); l1.set_params(f1)
    l2 = Oracle(R, C, R, S, np.float32); l2.set_params(f2)
    Wo, bo = W.copy(), b.copy(); Wc, bc = np.zeros_like(W), np.zeros_like(b)
    batcher = oc.MultiStreamBatcher(utts, S, T, delay)
    loss = frames = 0.0
    while True:
        nxt = batcher.next()
        if nxt is None:
            break
        feat, target, mask, flags = nxt
        l1.reset(flags); l2.reset(flags)
        h1 = l1.propagate(feat); h2 = l2.propagate(h1)
        y = oc.softmax(oc.affine_propagate(h2, Wo, bo))
        diff, xe, ent, cor, valid = oc.xent_eval_masked(y, target, mask)
        loss += xe - ent; frames += valid
        d2 = oc.affine_backpropagate(diff, Wo); oc.affine_update(h2, diff, Wo, bo, Wc, bc, lr, lr, mmt)
        d1 = l2.backpropagate(h1, d2, momentum=mmt); l2.update(lr)
        l1.backpropagate(feat, d1, momentum=mmt); l1.update(lr)
    assert head[0] == "OK" and float(head[3]) == frames
    assert abs(float(head[4]) - loss / frames) <= 3e-4 * abs(loss / frames)
    trained = [("transmit", I), ("lstm_streams", l1.get_params(), I, C, R, S), ("lstm_streams", l2.get_params(), R, C, R, S),
               ("affine", Wo, bo), ("softmax", NPDF)]
    (tmp_path / "exp.bin").write_bytes(kaldi_fmt.nnet_binary(trained))
    run("nnet_copy", tmp_path / "out.bin", 0, tmp_path / "out.txt")
    run("nnet_copy", tmp_path / "exp.bin", 0, tmp_path / "exp.txt")
    num = lambda p: np.array([float(v) for v in p.read_text().replace("[", " ").replace("]", " ").split() if v[0] in "-0123456789."])
    got, want = num(tmp_path / "out.txt"), num(tmp_path / "exp.txt")
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-3 * np.abs(want).max()
