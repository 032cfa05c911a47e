"""Independent (Python) assembly of Kaldi nnet1 model bytes for one LSTM component -- used to pin
include/klstm_kaldi_io.hpp from the outside.  Encoding facts:
  binary file header "\\0B"; token = ascii + ' '; int32/float = size byte (4) + little-endian payload;
  matrix = "FM " + int32 rows + int32 cols + packed row-major float32 (kaldi-matrix.cc:1176-1198);
  vector = "FV " + int32 dim + packed float32;  text matrix " [\\n  a b \\n  c d ]\\n" (kaldi-matrix.cc:1199-1211),
  text vector " [ a b c ]\\n" (cf. /root/reference/google/feature_transform.nnet.txt:3).
"""
import struct

import numpy as np


def _tok(t):
    return t.encode() + b" "


def _i32(v):
    return b"\x04" + struct.pack("<i", v)


def _mat(a):
    a = np.ascontiguousarray(a, dtype="<f4")
    return _tok("FM") + _i32(a.shape[0]) + _i32(a.shape[1]) + a.tobytes()


def _vec(a):
    a = np.ascontiguousarray(a, dtype="<f4")
    return _tok("FV") + _i32(a.shape[0]) + a.tobytes()


def parts(flat, I, C, R):
    o, out = 0, []
    for shp in [(4 * C, I), (4 * C, R), (4 * C,), (C,), (C,), (C,), (R, C)]:
        n = int(np.prod(shp))
        out.append(np.asarray(flat[o:o + n], np.float32).reshape(shp))
        o += n
    return out


def binary_model(flat, I, C, R, S, marker="<LstmProjectedStreams>"):
    wx, wr, b, pi, pf, po, wm = parts(flat, I, C, R)
    body = _tok(marker) + _i32(R) + _i32(I) + _tok("<CellDim>") + _i32(C)
    if marker == "<LstmProjectedStreams>":
        body += _tok("<NumStream>") + _i32(S)
    body += _mat(wx) + _mat(wr) + _vec(b) + _vec(pi) + _vec(pf) + _vec(po) + _mat(wm)
    return b"\x00B" + body


def _tmat(a, rowsep="\n"):
    if rowsep == ";":                 # the one-line form Kaldi also accepts: " [ 1 2 ; 3 4 ]"
        return " [ " + "; ".join("".join("%.9g " % v for v in row) for row in a) + "]\n"
    s = " ["
    for row in a:
        s += rowsep + "  " + "".join("%.9g " % v for v in row)
    return s + "]\n"


def _tvec(a):
    return " [ " + "".join("%.9g " % v for v in a) + "]\n"


def text_model(flat, I, C, R, S, marker="<LstmProjectedStreams>", rowsep="\n"):
    wx, wr, b, pi, pf, po, wm = parts(flat, I, C, R)
    s = "%s %d %d <CellDim> %d " % (marker, R, I, C)
    if marker == "<LstmProjectedStreams>":
        s += "<NumStream> %d " % S
    s += _tmat(wx, rowsep) + _tmat(wr, rowsep) + _tvec(b) + _tvec(pi) + _tvec(pf) + _tvec(po) + _tmat(wm, rowsep)
    return s.encode()


# ---- whole nets: "<Nnet>" components "</Nnet>" (README.md:24-45) ----
def _f32(v):
    return b"\x04" + struct.pack("<f", v)


def nnet_binary(layers):
    """layers: list of tuples ("transmit", dim) | ("timeshift", dim, shift) | ("lstm_streams", flat, I, C, R, S) |
    ("lstm", flat, I, C, R) | ("affine", W [out x in], b) | ("softmax", dim)"""
    out = b"\x00B" + _tok("<Nnet>")
    for l in layers:
        kind = l[0]
        if kind == "transmit":
            out += _tok("<Transmit>") + _i32(l[1]) + _i32(l[1])
        elif kind == "timeshift":
            out += _tok("<TimeShift>") + _i32(l[1]) + _i32(l[1]) + _tok("<Shift>") + _i32(l[2]) + b"\n"
        elif kind in ("lstm_streams", "lstm"):
            marker = "<LstmProjectedStreams>" if kind == "lstm_streams" else "<LstmProjected>"
            S = l[5] if kind == "lstm_streams" else 1
            out += binary_model(l[1], l[2], l[3], l[4], S, marker=marker)[2:]
        elif kind == "affine":
            W, b = l[1], l[2]
            out += (_tok("<AffineTransform>") + _i32(W.shape[0]) + _i32(W.shape[1]) + _tok("<LearnRateCoef>") + _f32(1.0) +
                    _tok("<BiasLearnRateCoef>") + _f32(1.0) + _tok("<MaxNorm>") + _f32(0.0) + _mat(W) + _vec(b))
        elif kind == "softmax":
            out += _tok("<Softmax>") + _i32(l[1]) + _i32(l[1])
    return out + _tok("</Nnet>")


def nnet_text(layers):
    s = "<Nnet> \n"
    for l in layers:
        kind = l[0]
        if kind == "transmit":
            s += "<Transmit> %d %d \n" % (l[1], l[1])
        elif kind == "timeshift":
            s += "<TimeShift> %d %d <Shift> %d \n" % (l[1], l[1], l[2])
        elif kind in ("lstm_streams", "lstm"):
            marker = "<LstmProjectedStreams>" if kind == "lstm_streams" else "<LstmProjected>"
            S = l[5] if kind == "lstm_streams" else 1
            s += text_model(l[1], l[2], l[3], l[4], S, marker=marker).decode()
        elif kind == "affine":
            W, b = l[1], l[2]
            s += "<AffineTransform> %d %d <LearnRateCoef> 1 <BiasLearnRateCoef> 1 <MaxNorm> 0 " % W.shape + _tmat(W) + _tvec(b)
        elif kind == "softmax":
            s += "<Softmax> %d %d \n" % (l[1], l[1])
    return (s + "</Nnet> \n").encode()
