"""ctypes loader for the CPU oracle (oracle/lstmp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (kaldi-lstm_amd/) never imports this module.

PARITY UNPINNED -- see the header of lstmp_oracle.c: the reference has no golden vectors
and cannot be built in this image.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile the C restatement (gcc, oracle/Makefile)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, f"liblstmp_oracle_{s}.so")) for s in ("f32", "f64"))
    if not need:
        src = os.path.getmtime(os.path.join(_HERE, "lstmp_oracle.c"))
        need = any(os.path.getmtime(os.path.join(_HERE, f"liblstmp_oracle_{s}.so")) < src
                   for s in ("f32", "f64"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def _lib(dtype):
    key = np.dtype(dtype).name
    if key in _LIBS:
        return _LIBS[key]
    build()
    suffix = {"float32": "f32", "float64": "f64"}[key]
    lib = ctypes.CDLL(os.path.join(_HERE, f"liblstmp_oracle_{suffix}.so"))
    real = ctypes.c_float if key == "float32" else ctypes.c_double
    P = ctypes.c_void_p
    lib.lstmp_oracle_create.restype = P
    lib.lstmp_oracle_create.argtypes = [ctypes.c_int] * 4
    lib.lstmp_oracle_destroy.argtypes = [P]
    lib.lstmp_oracle_set_threads.argtypes = [P, ctypes.c_int]
    lib.lstmp_oracle_num_params.restype = ctypes.c_long
    lib.lstmp_oracle_num_params.argtypes = [P]
    for name in ("set_params", "get_params", "set_corr", "get_corr", "get_state", "set_state"):
        getattr(lib, f"lstmp_oracle_{name}").argtypes = [P, P]
    lib.lstmp_oracle_prop_buf.restype = P
    lib.lstmp_oracle_prop_buf.argtypes = [P]
    lib.lstmp_oracle_bprop_buf.restype = P
    lib.lstmp_oracle_bprop_buf.argtypes = [P]
    lib.lstmp_oracle_reset.argtypes = [P, P, ctypes.c_int]
    lib.lstmp_oracle_propagate.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P, ctypes.c_int]
    lib.lstmp_oracle_backpropagate.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P, ctypes.c_int,
                                               P, ctypes.c_int, real]
    lib.lstmp_oracle_update.argtypes = [P, real, real]
    _LIBS[key] = lib
    return lib


def use_openblas(threads):
    """Route every GEMM of the fp32 oracle through cblas_sgemm of the OpenBLAS bundled with numpy (ILP64 build, symbol
    suffix 64_) on `threads` threads -- what a Kaldi CPU build linked against OpenBLAS executes (kaldi-matrix.cc:160-175).
    threads = 0 switches back to the plain loops.  Returns a description string, or None when the library is not found.
    Used by bench.py's cpu_baseline only; the parity tests keep the plain loops (fixed summation order)."""
    import glob
    lib = _lib(np.float32)
    lib.lstmp_oracle_set_sgemm.argtypes = [ctypes.c_void_p]
    if not threads:
        lib.lstmp_oracle_set_sgemm(None)
        return "plain loops"
    cands = glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs", "libscipy_openblas*.so*"))
    for path in cands:
        try:
            ob = ctypes.CDLL(path)
            for sfx in ("64_", ""):
                try:
                    fn = getattr(ob, "scipy_cblas_sgemm" + sfx)
                    setn = getattr(ob, "scipy_openblas_set_num_threads" + sfx)
                except AttributeError:
                    continue
                if sfx != "64_":
                    continue                       # the oracle's hook is declared for the ILP64 interface only
                setn.argtypes = [ctypes.c_int]
                setn(int(threads))
                lib.lstmp_oracle_set_sgemm(ctypes.cast(fn, ctypes.c_void_p))
                _LIBS["openblas"] = ob             # keep the handle alive
                return f"OpenBLAS cblas_sgemm ({os.path.basename(path)}, {threads} threads)"
        except OSError:
            continue
    return None


def param_sizes(I, C, R):
    """Flat blob layout = GetParams order (reference ...streams.h:162-189)."""
    return [("w_gifo_x", (4 * C, I)), ("w_gifo_r", (4 * C, R)), ("bias", (4 * C,)),
            ("peephole_i_c", (C,)), ("peephole_f_c", (C,)), ("peephole_o_c", (C,)),
            ("w_r_m", (R, C))]


def split_blob(flat, I, C, R):
    out, off = {}, 0
    for name, shp in param_sizes(I, C, R):
        n = int(np.prod(shp))
        out[name] = flat[off:off + n].reshape(shp)
        off += n
    assert off == flat.size
    return out


class Oracle:
    """One LstmProjectedStreams layer on the CPU, reference op order, fp32 or fp64."""

    def __init__(self, I, C, R, S, dtype=np.float32, threads=1):
        self.I, self.C, self.R, self.S = I, C, R, S
        self.W = 7 * C + R
        self.dtype = np.dtype(dtype)
        self.lib = _lib(dtype)
        self.h = self.lib.lstmp_oracle_create(I, C, R, S)
        self.lib.lstmp_oracle_set_threads(self.h, threads)
        self.T = 0

    def __del__(self):
        try:
            self.lib.lstmp_oracle_destroy(self.h)
        except Exception:
            pass

    @property
    def num_params(self):
        return int(self.lib.lstmp_oracle_num_params(self.h))

    def _arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    def set_params(self, flat):
        flat = self._arr(flat)
        assert flat.size == self.num_params
        self.lib.lstmp_oracle_set_params(self.h, flat.ctypes.data)

    def get_params(self):
        out = np.empty(self.num_params, self.dtype)
        self.lib.lstmp_oracle_get_params(self.h, out.ctypes.data)
        return out

    def set_corr(self, flat):
        flat = self._arr(flat)
        self.lib.lstmp_oracle_set_corr(self.h, flat.ctypes.data)

    def get_corr(self):
        out = np.empty(self.num_params, self.dtype)
        self.lib.lstmp_oracle_get_corr(self.h, out.ctypes.data)
        return out

    def get_state(self):
        out = np.empty((self.S, self.W), self.dtype)
        self.lib.lstmp_oracle_get_state(self.h, out.ctypes.data)
        return out

    def set_state(self, st):
        st = self._arr(st)
        assert st.shape == (self.S, self.W)
        self.lib.lstmp_oracle_set_state(self.h, st.ctypes.data)

    def reset(self, flags):
        f = np.ascontiguousarray(flags, dtype=np.int32)
        rc = self.lib.lstmp_oracle_reset(self.h, f.ctypes.data, int(f.size))
        if rc != 0:
            raise ValueError("reset: flags.size != num_stream")

    def propagate(self, x):
        x = self._arr(x)
        rows = x.shape[0]
        out = np.empty((rows, self.R), self.dtype)
        rc = self.lib.lstmp_oracle_propagate(self.h, x.ctypes.data, rows, x.shape[1], out.ctypes.data, self.R)
        if rc != 0:
            raise ValueError("propagate: rows % num_stream != 0")
        self.T = rows // self.S
        return out

    def backpropagate(self, x, out_diff, momentum=0.0, want_in_diff=True):
        x = self._arr(x)
        od = self._arr(out_diff)
        rows = x.shape[0]
        in_diff = np.empty((rows, self.I), self.dtype) if want_in_diff else None
        rc = self.lib.lstmp_oracle_backpropagate(
            self.h, x.ctypes.data, rows, x.shape[1], od.ctypes.data, od.shape[1],
            in_diff.ctypes.data if want_in_diff else None, self.I, float(momentum))
        if rc != 0:
            raise ValueError("backpropagate: shape does not match the preceding propagate")
        return in_diff

    def update(self, lr, clip_grad=0.0):
        self.lib.lstmp_oracle_update(self.h, float(lr), float(clip_grad))

    def _buf(self, ptr):
        n = (self.T + 2) * self.S * self.W
        ct = ctypes.c_float if self.dtype == np.float32 else ctypes.c_double
        arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(n,))
        return arr.reshape((self.T + 2) * self.S, self.W).copy()

    def prop_buf(self):
        """Activation slab [(T+2)S x (7C+R)], column groups G|I|F|O|C|H|M|R."""
        return self._buf(self.lib.lstmp_oracle_prop_buf(self.h))

    def bprop_buf(self):
        return self._buf(self.lib.lstmp_oracle_bprop_buf(self.h))


def make_params(I, C, R, scale=0.01, seed=7, dtype=np.float32):
    """U[-scale, +scale] parameters from a fixed-seed host RNG (InitMatParam/InitVecParam
    semantics, ...streams.h:41-53; the reference RNG itself is never part of parity)."""
    rng = np.random.RandomState(seed)
    n = 4 * C * I + 4 * C * R + 4 * C + 3 * C + R * C
    return ((rng.rand(n) - 0.5) * 2 * scale).astype(dtype)
