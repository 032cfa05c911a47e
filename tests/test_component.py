"""The C++ mirror of the reference component (include/klstm_component.hpp + klstm_kaldi_io.hpp),
driven through tests/cpp/component_test.  Model-file tests are CPU-only; the *_gpu tests run the
component's PropagateFnc/BackpropagateFnc/Update on the device and compare with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import Oracle, make_params
from tests import kaldi_fmt
from tests.margins import bound

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "component_test")
SRC = EXE + ".cpp"
HDRS = [os.path.join(ROOT, "include", h) for h in ("klstm.h", "klstm_component.hpp", "klstm_kaldi_io.hpp", "klstm_trainer.hpp", "klstm_nnet.hpp")]


def build_driver():
    import kaldi_lstm_amd as k
    lib = k.lib_path()
    assert os.path.exists(lib), "libklstm.so missing: run __graft_entry__.build()"
    stale = (not os.path.exists(EXE)) or any(os.path.getmtime(f) > os.path.getmtime(EXE) for f in [SRC] + HDRS)
    if stale:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), SRC,
                               "-L" + os.path.dirname(lib), "-lklstm", "-Wl,-rpath,$ORIGIN/../../kaldi-lstm_amd", "-o", EXE])
    return EXE


def run(*args, ok=True):
    r = subprocess.run([build_driver()] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    if ok:
        assert r.returncode == 0, r.stdout + r.stderr
    return r


def raw(path):
    return np.fromfile(path, dtype=np.float32)


DIMS = dict(I=5, C=7, R=4, S=3)


def test_binary_model_bytes_match_independent_assembly(tmp_path):
    I, C, R, S = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"]
    flat = make_params(I, C, R, scale=0.5, seed=3)
    ref = kaldi_fmt.binary_model(flat, I, C, R, S)
    (tmp_path / "ref.nnet").write_bytes(ref)
    out = run("dump_params", tmp_path / "ref.nnet", tmp_path / "p.raw").stdout.split()
    assert out == ["<LstmProjectedStreams>", str(I), str(R), str(C), str(S)]
    assert np.array_equal(raw(tmp_path / "p.raw"), flat)                       # reader: bit-exact
    run("convert", tmp_path / "ref.nnet", 1, tmp_path / "again.nnet")
    assert (tmp_path / "again.nnet").read_bytes() == ref                       # writer: byte-identical


@pytest.mark.parametrize("rowsep", ["\n", ";"])
def test_text_model_and_roundtrips(tmp_path, rowsep):
    I, C, R, S = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"]
    flat = make_params(I, C, R, scale=0.5, seed=4)
    (tmp_path / "m.txt").write_bytes(kaldi_fmt.text_model(flat, I, C, R, S, rowsep=rowsep))
    run("dump_params", tmp_path / "m.txt", tmp_path / "p.raw")
    assert np.array_equal(raw(tmp_path / "p.raw"), flat)        # %.9g text is lossless for float32
    # text -> binary -> text(6 digits, Kaldi's default ostream precision) -> binary
    run("convert", tmp_path / "m.txt", 1, tmp_path / "m.bin")
    assert (tmp_path / "m.bin").read_bytes() == kaldi_fmt.binary_model(flat, I, C, R, S)
    run("convert", tmp_path / "m.bin", 0, tmp_path / "m2.txt")
    txt = (tmp_path / "m2.txt").read_text()
    assert txt.startswith("<LstmProjectedStreams> %d %d <CellDim> %d <NumStream> %d  [\n  " % (R, I, C, S))   # README.md:40
    run("dump_params", tmp_path / "m2.txt", tmp_path / "p2.raw")
    np.testing.assert_allclose(raw(tmp_path / "p2.raw"), flat, rtol=1e-5)


def test_init_from_proto_and_param_range(tmp_path):
    proto = "<CellDim> 7 <ParamScale> 0.25 <NumStream> 3"        # google/nnet.proto:3 token set
    r = run("init_write", "<LstmProjectedStreams>", 5, 4, proto, 1, tmp_path / "m.bin", tmp_path / "p.raw")
    assert r.stdout.split() == ["OK", str(4 * 7 * 5 + 4 * 7 * 4 + 7 * 7 + 4 * 7)]
    p = raw(tmp_path / "p.raw")
    assert np.abs(p).max() <= 0.25 and np.abs(p).max() > 0.2 and abs(p.mean()) < 0.05     # U[-scale, scale] (:41-53)
    run("dump_params", tmp_path / "m.bin", tmp_path / "q.raw")
    assert np.array_equal(raw(tmp_path / "q.raw"), p)
    r = run("bad_proto", ok=False)
    assert r.returncode == 3 and "Unknown token <Bogus>, a typo in config?" in r.stdout       # KALDI_ERR :70


def test_google_to_standard_conversion_by_text_edit(tmp_path):
    """README.md:18-30: rename LstmProjectedStreams -> LstmProjected and drop the NumStream tag."""
    I, C, R, S = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"]
    flat = make_params(I, C, R, scale=0.5, seed=5)
    txt = kaldi_fmt.text_model(flat, I, C, R, S).decode()
    txt = txt.replace("<LstmProjectedStreams>", "<LstmProjected>").replace("<NumStream> %d " % S, "")
    (tmp_path / "std.txt").write_text(txt)
    out = run("dump_params", tmp_path / "std.txt", tmp_path / "p.raw").stdout.split()
    assert out == ["<LstmProjected>", str(I), str(R), str(C), "1"]
    assert np.array_equal(raw(tmp_path / "p.raw"), flat)
    assert (tmp_path / "std.txt").read_bytes() == kaldi_fmt.text_model(flat, I, C, R, 1, marker="<LstmProjected>")


def test_truncated_and_malformed_models_raise(tmp_path):
    I, C, R, S = DIMS["I"], DIMS["C"], DIMS["R"], DIMS["S"]
    ref = kaldi_fmt.binary_model(make_params(I, C, R, seed=6), I, C, R, S)
    (tmp_path / "trunc.nnet").write_bytes(ref[:len(ref) - 10])
    r = run("dump_params", tmp_path / "trunc.nnet", tmp_path / "p.raw", ok=False)
    assert r.returncode == 3 and "cannot read matrix" in r.stdout and "ended inside the payload" in r.stdout
    (tmp_path / "bad.nnet").write_bytes(ref.replace(b"<CellDim>", b"<CellDum>"))
    r = run("dump_params", tmp_path / "bad.nnet", tmp_path / "p.raw", ok=False)
    assert r.returncode == 3 and "found token" in r.stdout and "'<CellDim>' belongs" in r.stdout
    wrong = kaldi_fmt.binary_model(make_params(I, C + 1, R, seed=6), I, C + 1, R, S).replace(
        b"<CellDim> \x04" + bytes([C + 1, 0, 0, 0]), b"<CellDim> \x04" + bytes([C, 0, 0, 0]))
    (tmp_path / "dims.nnet").write_bytes(wrong)
    r = run("dump_params", tmp_path / "dims.nnet", tmp_path / "p.raw", ok=False)
    assert r.returncode == 3 and "expected" in r.stdout


def test_text_and_binary_matrix_reader_edge_cases(tmp_path):
    """The matrix reader of include/klstm_kaldi_io.hpp against the on-disk forms kaldi-matrix.cc:1172-1212 / :1243-1406 define:
    rows split at newlines or ';', blank rows ignored, "[]" and "[ ]" = empty, inf / nan / infinity in any case, literals
    beyond the float range become +-inf, one line end ("\n" or "\r\n") behind the closing bracket is consumed (objects
    follow each other back to back), DM (double) payloads are converted; ragged rows, non-numbers and truncated objects raise."""
    import struct
    txt = (" [\n  1 2 3 \n  4 5 6 ]\n"            # the writer's own layout
           "[ 7 8 ; 9 10 ;\n ]\r\n"                # ';' rows, an empty trailing row, CRLF
           " []\n [ ]\n"                           # two empty matrices
           "[ inf -INF NaN 1e+50 -1e+50 Infinity ]\n"
           "[ -0.5 ]")                              # no newline at the very end
    (tmp_path / "m.txt").write_text(txt)
    r = run("read_matrices", tmp_path / "m.txt", tmp_path / "m.raw")
    assert r.stdout.split("\n")[:6] == ["2 3", "2 2", "0 0", "0 0", "1 6", "1 1"]
    v = raw(tmp_path / "m.raw")
    assert np.array_equal(v[:10], np.arange(1, 11, dtype=np.float32))
    assert v[10] == np.inf and v[11] == -np.inf and np.isnan(v[12]) and v[13] == np.inf and v[14] == -np.inf and v[15] == np.inf
    assert v[16] == -0.5 and v.size == 17
    a = np.arange(6, dtype=np.float32).reshape(2, 3) / 7
    b = np.arange(4, dtype=np.float64).reshape(1, 4) / 3
    blob = (b"\0B" + b"FM " + b"\x04" + struct.pack("<i", 2) + b"\x04" + struct.pack("<i", 3) + a.tobytes() +
            b"DM " + b"\x04" + struct.pack("<i", 1) + b"\x04" + struct.pack("<i", 4) + b.tobytes())
    (tmp_path / "m.bin").write_bytes(blob)
    r = run("read_matrices", tmp_path / "m.bin", tmp_path / "b.raw")
    assert r.stdout.split("\n")[:2] == ["2 3", "1 4"]
    assert np.array_equal(raw(tmp_path / "b.raw"), np.concatenate([a.ravel(), b.astype(np.float32).ravel()]))
    for bad, msg in (("[ 1 2 ; 3 ]", "row 1 has 1 entries"), ("[ 1 two ]", "'two' is not a number"), ("[ 1 2", "no closing ']'"),
                     ("1 2 ]", "where '[' should be")):
        (tmp_path / "bad.txt").write_text(bad)
        r = run("read_matrices", tmp_path / "bad.txt", tmp_path / "x.raw", ok=False)
        assert r.returncode == 3 and "cannot read matrix" in r.stdout and msg in r.stdout, (bad, r.stdout)
    (tmp_path / "bad.bin").write_bytes(blob[:len(blob) - 5])
    r = run("read_matrices", tmp_path / "bad.bin", tmp_path / "x.raw", ok=False)
    assert r.returncode == 3 and "ended inside the payload" in r.stdout


def _check_moment_report(text, flat, I, C, R, suffix):
    import re
    names = ["w_gifo_x", "w_gifo_r", "bias", "peephole_i_c", "peephole_f_c", "peephole_o_c", "w_r_m"]
    lens = [4 * C * I, 4 * C * R, 4 * C, C, C, C, R * C]
    rows = re.findall(r"\n  (\w+)   \( min (\S+), max (\S+), mean (\S+), variance (\S+), skewness (\S+), kurtosis (\S+) \)", text)
    assert [r[0] for r in rows] == [n + suffix for n in names], text
    off = 0
    for (name, *vals), n in zip(rows, lens):
        v = np.asarray(flat[off:off + n], np.float64); off += n
        d = v - v.mean()
        var = (d ** 2).mean()
        want = [v.min(), v.max(), v.mean(), var, (d ** 3).mean() / var ** 1.5, (d ** 4).mean() / var ** 2 - 3.0]
        got = [float(x) for x in vals]
        scale = np.abs(v).max()
        for g, w, what in zip(got[:3], want[:3], ("min", "max", "mean")):
            assert abs(g - w) <= 2e-3 * scale, (name, what, g, w)
        assert abs(got[3] - want[3]) <= 5e-3 * want[3], (name, "variance", got[3], want[3])
        assert abs(got[4] - want[4]) <= 2e-2 and abs(got[5] - want[5]) <= 5e-2 * max(1.0, abs(want[5])), (name, got[4:], want[4:])


def _relerr(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("marker,S,mode", [("<LstmProjectedStreams>", 4, "run_gpu"), ("<LstmProjected>", 1, "run_gpu"),
                                           ("<LstmProjectedStreams>", 4, "run_gpu_host"),
                                           ("<LstmProjectedStreams>", 4, "run_gpu_fold"), ("<LstmProjected>", 1, "run_gpu_fold"),
                                           ("<LstmProjectedStreams>", 4, "run_gpu_dp"),      # C++ data-parallel step over RCCL (1 rank)
                                           ("<LstmProjectedStreams>", 4, "run_gpu_copy")])   # Copy() after the first minibatch
def test_component_train_steps_gpu(tmp_path, marker, S, mode, monkeypatch):
    """Reset -> (PropagateFnc, BackpropagateFnc, Update) x3 through the C++ mirror, pitched device
    matrices (run_gpu_host: pitched HOST matrices, staged by the adapter); <LstmProjected> = standard/
    semantics (zero history per call, +-50 clip in Update)."""
    I, C, R, T = 40, 64, 32, 6
    flat = make_params(I, C, R, scale=0.2, seed=7)
    (tmp_path / "m.nnet").write_bytes(kaldi_fmt.binary_model(flat, I, C, R, S, marker=marker))
    rng = np.random.RandomState(0)
    x = rng.randn(T * S, I).astype(np.float32)
    od = (300.0 * rng.randn(T * S, R)).astype(np.float32) if S == 1 else rng.randn(T * S, R).astype(np.float32)
    x.tofile(tmp_path / "x.raw"); od.tofile(tmp_path / "od.raw")
    lr, mmt, nsteps = 1e-4, 0.9, 3
    if mode == "run_gpu_fold":                 # SetEngineOption("fold", 1): the folded chain behind the same Component calls
        monkeypatch.setenv("KLSTM_TEST_FOLD", "1")
        mode = "run_gpu"
    dp = mode == "run_gpu_dp"
    if mode == "run_gpu_dp":                   # SetDataParallel(comm): deferred momentum -> klstm_allreduce_grads -> momentum -> update,
        monkeypatch.setenv("KLSTM_TEST_DP", "1:0:%s" % (tmp_path / "rccl.id"))     # all issued by the C++ mirror through the C-ABI
        mode = "run_gpu"
    if mode == "run_gpu_copy":                 # the rest of the run happens on a Copy(): state, momentum, options carried over
        monkeypatch.setenv("KLSTM_TEST_COPY", "1")
        monkeypatch.setenv("KLSTM_TEST_FOLD", "1")
        mode = "run_gpu"
    res = run(mode, tmp_path / "m.nnet", tmp_path / "x.raw", tmp_path / "od.raw", T * S, lr, mmt, nsteps, tmp_path / "res")
    o = Oracle(I, C, R, S, np.float32); o.set_params(flat)
    std = marker == "<LstmProjected>"
    for _ in range(nsteps):
        if std:
            o.reset([1])
        out_o = o.propagate(x)
        corr_before_backprop = o.get_corr().copy()
        id_o = o.backpropagate(x, od, momentum=mmt)
        # (data-parallel order: BackpropagateFnc leaves the pure gradient in the blob, the momentum buffers move inside Update)
        corr_before_update = corr_before_backprop if dp else o.get_corr().copy()
        o.update(lr, clip_grad=50.0 if std else 0.0)
    if std:
        assert np.abs(o.get_corr()).max() == 50.0            # the clip is exercised
    assert _relerr(raw(tmp_path / "res.out").reshape(T * S, R), out_o) <= 5e-5
    assert _relerr(raw(tmp_path / "res.in_diff").reshape(T * S, I), id_o) <= 2e-4
    assert _relerr(raw(tmp_path / "res.params"), o.get_params()) <= 5e-5
    # the trained model written by the component is a valid Kaldi binary model of the same numbers
    expect = kaldi_fmt.binary_model(raw(tmp_path / "res.params"), I, C, R, S, marker=marker)
    assert (tmp_path / "res.model").read_bytes() == expect
    # Info() / InfoGradient() (...streams.h:190-210): the seven tensors under the reference's names, in its order, each with the six
    # moments of MomentStatistics -- InfoGradient was taken between the last BackpropagateFnc and its Update, Info after it
    _check_moment_report((tmp_path / "res.gradinfo").read_text(), corr_before_update, I, C, R, "_corr_")
    _check_moment_report(res.stdout.split("OK", 1)[1], o.get_params(), I, C, R, "_")


@pytest.mark.gpu
@pytest.mark.parametrize("marker,S,T,nmb", [("<LstmProjectedStreams>", 4, 20, 5), ("<LstmProjected>", 1, 1000, 2),
                                             ("<LstmProjectedStreams>", 8, 20, 3),      # configs[2]'s per-GPU shard: interleaved chains + tail workgroups
                                             ("<LstmProjectedStreams>", 12, 20, 3)])    # round 6: three interleaved chains per launch
def test_component_at_the_benchmarked_shape_gpu(tmp_path, marker, S, T, nmb):
    """The Kaldi-side component at the shape bench.py measures (BASELINE.json configs[1]: 40 -> 800 / 512, NumStream 4, T = 20;
    and configs[0]: <LstmProjected> over 1000-frame utterances), constructed and driven exactly as the shim of INTEGRATION.md 2
    does it: SetUpdateFollows(true), persist_verify at the mirror's default (1), pitched device matrices, and per minibatch the
    trainer's call order -- Reset(new_utt_flags), PropagateFnc, BackpropagateFnc, Update
    (bd-nnet-train-lstm-streams.cc:209-228; ...streams.h:222, :334-335, :501).  5 chained minibatches (state carried, streams 1
    and 3 start new utterances at minibatch 2, all of them at minibatch 0), every minibatch's out / in_diff and the final
    parameters, momentum buffers and carried state against the oracle at the ENGINE tests' bars; klstm_profile_query must show
    that the persistent launches (the kernels bench.py times) were the ones that ran, and that none gave up."""
    I, C, R = 40, 800, 512
    flat = make_params(I, C, R, scale=0.01, seed=31)
    (tmp_path / "m.nnet").write_bytes(kaldi_fmt.binary_model(flat, I, C, R, S, marker=marker))
    rng = np.random.RandomState(5)
    rows = T * S
    x = rng.randn(nmb, rows, I).astype(np.float32)
    od = (0.1 * rng.randn(nmb, rows, R)).astype(np.float32)
    flags = np.zeros((nmb, S), np.float32)
    flags[0, :] = 1
    if S > 1:
        flags[2, 1] = flags[2, 3] = 1
    x.tofile(tmp_path / "x.raw"); od.tofile(tmp_path / "od.raw"); flags.tofile(tmp_path / "fl.raw")
    lr, mmt = 1e-5, 0.9
    res = run("run_gpu_full", tmp_path / "m.nnet", tmp_path / "x.raw", tmp_path / "od.raw", tmp_path / "fl.raw", rows, nmb, lr, mmt,
              tmp_path / "res")
    counters = dict(kv.split("=") for kv in res.stdout.split("\n")[0].split()[1:])
    std = marker == "<LstmProjected>"
    o = Oracle(I, C, R, S, np.float32); o.set_params(flat)
    out_g = raw(tmp_path / "res.out").reshape(nmb, rows, R)
    id_g = raw(tmp_path / "res.in_diff").reshape(nmb, rows, I)
    for mb in range(nmb):
        o.reset([1] if std else flags[mb].astype(np.int32))
        out_o = o.propagate(x[mb])
        id_o = o.backpropagate(x[mb], od[mb], momentum=mmt)
        o.update(lr, clip_grad=50.0 if std else 0.0)
        bound(_relerr(out_g[mb], out_o), 2e-5, "out")
        bound(_relerr(id_g[mb], id_o), 5e-5, "in_diff")
    from oracle.oracle import split_blob
    gc, oc = split_blob(raw(tmp_path / "res.corr"), I, C, R), split_blob(o.get_corr(), I, C, R)
    gp, op = split_blob(raw(tmp_path / "res.params"), I, C, R), split_blob(o.get_params(), I, C, R)
    for name in gc:                                  # each of the seven tensors against its own maximum
        bound(_relerr(gc[name], oc[name]), 5e-5, "corr." + name)
        bound(_relerr(gp[name], op[name]), 2e-5, "params." + name)
    st = o.get_state()
    bound(_relerr(raw(tmp_path / "res.state_c").reshape(S, C), st[:, 4 * C:5 * C]), 2e-5, "state_c")
    bound(_relerr(raw(tmp_path / "res.state_r").reshape(S, R), st[:, 7 * C:]), 2e-5, "state_r")
    # the benchmarked kernels ran: one persistent launch per direction and minibatch, none gave up, nothing was run again
    assert int(counters["persist_launches"]) == 2 * nmb, counters
    assert counters["persist_giveups"] == "0" and counters["persist_replayed"] == "0" and counters["persist_dropped"] == "0", counters


def _pack_utts(utts):
    parts = [np.float32([len(utts)])]
    for f, t in utts:
        parts += [np.float32([f.shape[0], f.shape[1], len(t)]), f.astype(np.float32).ravel(), np.asarray(t, np.float32)]
    return np.concatenate(parts)


@pytest.mark.parametrize("S,T,delay,lens", [
    (4, 20, 5, [53, 20, 7, 41, 100, 3, 64]),      # ragged, refills at batch boundaries only, shorter-than-delay utterance
    (2, 5, 0, [5, 10, 1]),                        # exact multiples, no delay
    (3, 4, 2, [9, 9, 9]),                         # exactly S utterances
])
def test_multistream_batcher_matches_reference_bookkeeping(tmp_path, S, T, delay, lens):
    """klstm_kaldi::MultiStreamBatcher vs the numpy restatement of bd-nnet-train-lstm-streams.cc:128-206."""
    from oracle.components import MultiStreamBatcher
    rng = np.random.RandomState(0)
    dim = 3
    utts = [(rng.randn(n, dim).astype(np.float32), rng.randint(0, 50, n)) for n in lens]
    utts.insert(2, (rng.randn(6, dim).astype(np.float32), rng.randint(0, 50, 4)))      # length mismatch: skipped (:160-164)
    _pack_utts(utts).tofile(tmp_path / "u.raw")
    r = run("batcher", tmp_path / "u.raw", S, T, delay, tmp_path / "b.raw").stdout.split()
    ob = MultiStreamBatcher(utts, S, T, delay)
    exp, nb = [], 0
    while True:
        b = ob.next()
        if b is None:
            break
        feat, target, mask, flags = b
        exp += [feat.ravel(), target.astype(np.float32), mask, np.float32(flags)]
        nb += 1
    assert r == ["OK", str(nb), str(len(lens)), "1"]
    assert np.array_equal(raw(tmp_path / "b.raw"), np.concatenate(exp))
    assert sum(e.sum() for e in exp[2::4]) == sum(lens)            # every real frame is a valid frame exactly once


def test_batcher_refuses_fewer_utterances_than_streams(tmp_path):
    rng = np.random.RandomState(1)
    utts = [(rng.randn(5, 2).astype(np.float32), rng.randint(0, 9, 5))]
    _pack_utts(utts).tofile(tmp_path / "u.raw")
    r = run("batcher", tmp_path / "u.raw", 2, 4, 1, tmp_path / "b.raw", ok=False)
    assert r.returncode == 3 and "fewer utterances than streams" in r.stdout


def test_format_against_the_reference_data_files(tmp_path):
    """tests/golden/ holds the reference's own prototype file (google/nnet.proto) and the component header lines of its
    README's example models (tests/golden/make_fixtures.py).  InitData must accept the prototype line verbatim, and the
    text models written here must start with exactly the README's header lines (marker, output-dim, input-dim, then the
    WriteData tokens) -- the format side of the drop-in boundary pinned to reference-provided data."""
    gold = os.path.join(os.path.dirname(__file__), "golden")
    proto = [l.split() for l in open(os.path.join(gold, "nnet.proto")) if l.startswith("<LstmProjectedStreams>")][0]
    headers = open(os.path.join(gold, "model_headers.txt")).read().splitlines()
    assert proto[1] == "<InputDim>" and proto[3] == "<OutputDim>"
    I, O, rest = int(proto[2]), int(proto[4]), " ".join(proto[5:])
    r = run("init_write", "<LstmProjectedStreams>", I, O, rest, 0, tmp_path / "m.txt", tmp_path / "p.raw")
    assert r.stdout.split() == ["OK", "2181600"]                     # 2 181 600 parameters at 40/800/512
    p = raw(tmp_path / "p.raw")
    assert p.size == 2181600 and np.abs(p).max() <= 0.01 and np.abs(p).max() > 0.0099 and abs(p.mean()) < 1e-4   # <ParamScale> 0.01
    first = (tmp_path / "m.txt").read_text().split("[", 1)[0].split()
    assert " ".join(first) == "<LstmProjectedStreams> 512 40 <CellDim> 800 <NumStream> 4" and " ".join(first) in headers
    # the standard/ component: same prototype without <NumStream>; header as in the README's converted model
    rest_std = " ".join(t for i, t in enumerate(proto[5:]) if t != "<NumStream>" and proto[5:][i - 1] != "<NumStream>")
    run("init_write", "<LstmProjected>", I, O, rest_std, 0, tmp_path / "s.txt", tmp_path / "ps.raw")
    first = (tmp_path / "s.txt").read_text().split("[", 1)[0].split()
    assert " ".join(first) == "<LstmProjected> 512 40 <CellDim> 800" and " ".join(first) in headers
    # TimeShift / Transmit header lines survive an nnet-copy round trip byte for byte
    ts = [h for h in headers if h.startswith("<TimeShift>")][0]
    tr = [h for h in headers if h.startswith("<Transmit>")][0]
    (tmp_path / "n.txt").write_text("<Nnet>\n%s\n%s\n</Nnet>\n" % (ts, tr))
    run("nnet_copy", tmp_path / "n.txt", 0, tmp_path / "n2.txt")
    lines = [l.strip() for l in (tmp_path / "n2.txt").read_text().splitlines() if l.strip()]
    assert lines == ["<Nnet>", ts, tr, "</Nnet>"]


def test_reader_on_a_kaldi_written_file(tmp_path):
    """tests/golden/feature_transform.nnet.txt is the reference's own (Kaldi-written, text mode) feature transform:
    <Nnet> <AddShift> 40 40 [ v ] <Rescale> 40 40 [ v ] </Nnet>.  The token / integer / vector readers of
    include/klstm_kaldi_io.hpp must consume it and return the numbers printed in the file."""
    path = os.path.join(os.path.dirname(__file__), "golden", "feature_transform.nnet.txt")
    r = run("read_vectors", path, tmp_path / "v.raw")
    assert r.stdout.split("\n")[:2] == ["<AddShift> 40 40 40", "<Rescale> 40 40 40"]
    txt = open(path).read()
    expect = [float(t) for blk in txt.split("[")[1:] for t in blk.split("]")[0].split()]
    assert len(expect) == 80
    assert np.array_equal(raw(tmp_path / "v.raw"), np.float32(expect))
