"""ctypes binding of include/klstm.h.  Mirrors the reference component's method names
(google/nnet/bd-nnet-lstm-projected-streams.h): Reset / PropagateFnc / BackpropagateFnc /
Update / NumParams / GetParams, with torch CUDA tensors standing in for CuMatrixBase
(data pointer + row stride are passed through untouched)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFER_MOMENTUM = 1
_LIB = None


class KlstmError(RuntimeError):
    """Non-zero klstm_status (the adapter's KALDI_ERR equivalent)."""

    def __init__(self, status, msg):
        super().__init__(f"klstm status {status}: {msg}")
        self.status = status


def lib_path():
    return os.path.join(_HERE, "libklstm.so")


def load_library():
    """dlopen the in-tree libklstm.so.  If it has not been built yet, or its sources changed since, it is compiled in place with hipcc (gfx950);
    there is no other implementation to fall back to -- without the HIP library every call raises."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    alt = os.environ.get("KLSTM_LIB_PATH")             # A-B experiments (tools/): another build of the same sources, loaded as is
    try:                               # no-op unless the sources changed since the library was built (content hash)
        if alt:
            path = alt
        import importlib.util
        spec = importlib.util.spec_from_file_location("klstm_build", os.path.join(_HERE, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if not alt:
            mod.build()
    except Exception as exc:       # noqa: BLE001
        raise KlstmError(-1, f"{path} missing or stale and could not be built with hipcc: {exc}") from exc
    lib = ctypes.CDLL(path)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.klstm_last_error.restype = ctypes.c_char_p
    lib.klstm_version.restype = ctypes.c_char_p
    lib.klstm_create.argtypes = [I, I, I, I, I, P, ctypes.POINTER(P)]
    lib.klstm_destroy.argtypes = [P]
    lib.klstm_destroy.restype = None
    for n in ("input_dim", "cell_dim", "recur_dim", "num_stream"):
        getattr(lib, f"klstm_{n}").argtypes = [P]
    lib.klstm_num_params.argtypes = [P]
    lib.klstm_num_params.restype = ctypes.c_long
    lib.klstm_grad_blob_len.argtypes = [P]
    lib.klstm_grad_blob_len.restype = ctypes.c_long
    for n in ("set_params_host", "get_params_host", "set_params_device", "get_corr_host",
              "set_corr_host", "get_grads_host"):
        getattr(lib, f"klstm_{n}").argtypes = [P, P]
    lib.klstm_grad_blob.argtypes = [P]
    lib.klstm_grad_blob.restype = P
    lib.klstm_param_blob.argtypes = [P]
    lib.klstm_param_blob.restype = P
    lib.klstm_reset.argtypes = [P, P, I]
    lib.klstm_get_state_host.argtypes = [P, P, P]
    lib.klstm_set_state_host.argtypes = [P, P, P]
    lib.klstm_propagate.argtypes = [P, P, I, I, P, I]
    lib.klstm_backpropagate.argtypes = [P, P, I, P, I, P, I, I, F, I]
    lib.klstm_propagate_host.argtypes = [P, P, I, I, P, I]
    lib.klstm_backpropagate_host.argtypes = [P, P, I, P, I, P, I, I, F, I]
    lib.klstm_pointer_on_device.argtypes = [P, P]
    lib.klstm_pointer_on_device.restype = I
    lib.klstm_apply_momentum.argtypes = [P, F]
    lib.klstm_update.argtypes = [P, F, F]
    lib.klstm_synchronize.argtypes = [P]
    lib.klstm_get_activations_host.argtypes = [P, I, P]
    lib.klstm_set_option.argtypes = [P, ctypes.c_char_p, I]
    lib.klstm_debug_occupy.argtypes = [I, I, I, P, P]
    lib.klstm_debug_gemm_bf16_nt2.argtypes = [I, P, P, P, I, I, P, P]
    lib.klstm_debug_gemm_bf16_nt2h.argtypes = [I, P, P, P, P, I, I, P, P]
    lib.klstm_profile_query.argtypes = [P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_long)]
    lib.klstm_time_shift.argtypes = [P, I, I, I, P, I, I, P]
    lib.klstm_affine_propagate.argtypes = [P, I, I, I, P, P, P, I, I, P]
    lib.klstm_affine_backpropagate.argtypes = [P, I, I, I, P, I, P, I, P]
    lib.klstm_affine_update.argtypes = [P, I, P, I, I, I, I, P, P, P, P, F, F, F, P]
    lib.klstm_softmax.argtypes = [P, I, I, I, P, I, P]
    lib.klstm_bind_grad_blob.argtypes = [P, P]
    lib.klstm_affine_gradient.argtypes = [P, I, P, I, I, I, I, P, P, P]
    lib.klstm_sgd_momentum_update.argtypes = [P, P, P, ctypes.c_long, F, F, P]
    lib.klstm_xent_eval_masked.argtypes = [P, I, I, I, P, P, P, I, P, P, P]
    lib.klstm_xent_eval_masked_post.argtypes = [P, I, I, I, P, P, P, P, P, I, P, P, P, P]
    lib.klstm_xent_accumulate.argtypes = [P, P, P, I, P, P]
    lib.klstm_softmax_xent_masked.argtypes = [P, I, I, I, P, I, P, P, P, I, P, P, P, P]
    lib.klstm_oneshot_create.argtypes = [I, P, ctypes.c_long, ctypes.POINTER(P)]
    lib.klstm_oneshot_export.argtypes = [P, P, P]
    lib.klstm_oneshot_connect.argtypes = [P, I, I, P, P]
    lib.klstm_oneshot_allreduce.argtypes = [P, P, I]
    lib.klstm_oneshot_status.argtypes = [P, ctypes.POINTER(ctypes.c_uint)]
    lib.klstm_oneshot_destroy.argtypes = [P]
    lib.klstm_oneshot_last_error.restype = ctypes.c_char_p
    lib.klstm_allreduce_grads_oneshot.argtypes = [P, P, I]
    lib.klstm_comm_get_unique_id.argtypes = [P]
    lib.klstm_comm_init_rank.argtypes = [I, I, I, P, ctypes.POINTER(P)]
    lib.klstm_comm_destroy.argtypes = [P]
    lib.klstm_comm_count.argtypes = [P, ctypes.POINTER(I)]
    lib.klstm_allreduce_grads.argtypes = [P, P]
    lib.klstm_allreduce_buffer.argtypes = [P, ctypes.c_size_t, P, P]
    _LIB = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


BPTT_DEFER_MOMENTUM = 1
BPTT_FUSE_UPDATE = 2      # klstm.h: "klstm_update follows immediately, input rows unchanged"


class Engine:
    """One LstmProjectedStreams layer on one MI355X."""

    def __init__(self, input_dim, cell_dim, recur_dim, num_stream, device=0, stream=None):
        self.lib = load_library()
        self.I, self.C, self.R, self.S = input_dim, cell_dim, recur_dim, num_stream
        self.W = 7 * cell_dim + recur_dim
        self._stream_obj = stream                      # keep a torch stream alive if given
        handle = ctypes.c_void_p()
        sptr = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
        self._chk(self.lib.klstm_create(input_dim, cell_dim, recur_dim, num_stream, device, sptr,
                                        ctypes.byref(handle)))
        self.h = handle
        self.T = 0
        self._keep = []          # tensors referenced by queued (asynchronous) work

    def close(self):
        if getattr(self, "h", None):
            self.lib.klstm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, status):
        if status != 0:
            raise KlstmError(status, self.lib.klstm_last_error().decode())

    # ---- NumParams / GetParams (reference :152-189) ----
    @property
    def num_params(self):
        return int(self.lib.klstm_num_params(self.h))

    def set_params(self, flat):
        flat = _f32(flat)
        assert flat.size == self.num_params
        self._chk(self.lib.klstm_set_params_host(self.h, flat.ctypes.data))

    def get_params(self):
        out = np.empty(self.num_params, np.float32)
        self._chk(self.lib.klstm_get_params_host(self.h, out.ctypes.data))
        return out

    def get_corr(self):
        out = np.empty(self.num_params, np.float32)
        self._chk(self.lib.klstm_get_corr_host(self.h, out.ctypes.data))
        return out

    def set_corr(self, flat):
        flat = _f32(flat)
        assert flat.size == self.num_params
        self._chk(self.lib.klstm_set_corr_host(self.h, flat.ctypes.data))

    def get_grads(self):
        out = np.empty(self.num_params, np.float32)
        self._chk(self.lib.klstm_get_grads_host(self.h, out.ctypes.data))
        return out

    def grad_blob_ptr(self):
        return int(self.lib.klstm_grad_blob(self.h))

    def grad_blob_tensor(self, full=False):
        """Zero-copy torch view of the device gradient blob (for dist.all_reduce).  full=True: klstm_grad_blob_len() floats -- the
        gradient plus the validity word of data-parallel runs (klstm.h), what the library's own collectives cover."""
        import torch

        class _Blob:
            pass
        b = _Blob()
        n = int(self.lib.klstm_grad_blob_len(self.h)) if full else self.num_params
        b.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4",
                                      "data": (self.grad_blob_ptr(), False), "version": 2}
        return torch.as_tensor(b, device="cuda")

    def param_blob_tensor(self):
        """Zero-copy torch view of the device parameter blob (read it; write through set_params_device)."""
        import torch

        class _Blob:
            pass
        b = _Blob()
        b.__cuda_array_interface__ = {"shape": (self.num_params,), "typestr": "<f4",
                                      "data": (int(self.lib.klstm_param_blob(self.h)), False), "version": 2}
        return torch.as_tensor(b, device="cuda")

    def set_params_device(self, t):
        """Load parameters from a contiguous CUDA float32 tensor (device-to-device, stream-ordered)."""
        assert t.is_cuda and t.is_contiguous() and t.numel() == self.num_params
        self._keep.append(t)
        self._chk(self.lib.klstm_set_params_device(self.h, t.data_ptr()))

    # ---- Reset (reference :212-220) ----
    def reset(self, flags):
        f = np.ascontiguousarray(flags, dtype=np.int32)
        self._chk(self.lib.klstm_reset(self.h, f.ctypes.data, int(f.size)))

    def get_state(self):
        c = np.empty((self.S, self.C), np.float32)
        r = np.empty((self.S, self.R), np.float32)
        self._chk(self.lib.klstm_get_state_host(self.h, c.ctypes.data, r.ctypes.data))
        return c, r

    def set_state(self, c, r):
        c, r = _f32(c), _f32(r)
        assert c.shape == (self.S, self.C) and r.shape == (self.S, self.R)
        self._chk(self.lib.klstm_set_state_host(self.h, c.ctypes.data, r.ctypes.data))

    # ---- PropagateFnc / BackpropagateFnc / Update (reference :222, :334, :501) ----
    def propagate(self, x, out):
        """x [rows, I], out [rows, R]: torch CUDA float32, last dim contiguous."""
        assert x.is_cuda and out.is_cuda and (x.shape[0] == 0 or (x.stride(1) == 1 and out.stride(1) == 1))
        rows = x.shape[0]
        self._keep = [x, out]
        self._chk(self.lib.klstm_propagate(self.h, x.data_ptr(), rows, x.stride(0), out.data_ptr(), out.stride(0)))
        self.T = rows // self.S

    def backpropagate(self, x, out_diff, in_diff=None, momentum=0.0, flags=0):
        assert x.is_cuda and out_diff.is_cuda and (x.shape[0] == 0 or (x.stride(1) == 1 and out_diff.stride(1) == 1))
        idp, ids = (in_diff.data_ptr(), in_diff.stride(0)) if in_diff is not None else (None, 0)
        self._keep += [x, out_diff, in_diff]
        self._chk(self.lib.klstm_backpropagate(self.h, x.data_ptr(), x.stride(0), out_diff.data_ptr(),
                                               out_diff.stride(0), idp, ids, x.shape[0],
                                               float(momentum), int(flags)))

    # ---- host-matrix variants (numpy float32, row stride = ld elements): staged through the device path ----
    def propagate_host(self, x, out, x_ld=None, out_ld=None):
        rows = x.shape[0]
        self._chk(self.lib.klstm_propagate_host(self.h, x.ctypes.data, rows, x_ld or x.strides[0] // 4,
                                                out.ctypes.data, out_ld or out.strides[0] // 4))
        self.T = rows // self.S

    def backpropagate_host(self, x, out_diff, in_diff=None, momentum=0.0, flags=0):
        idp, ids = (in_diff.ctypes.data, in_diff.strides[0] // 4) if in_diff is not None else (None, 0)
        self._chk(self.lib.klstm_backpropagate_host(self.h, x.ctypes.data, x.strides[0] // 4, out_diff.ctypes.data,
                                                    out_diff.strides[0] // 4, idp, ids, x.shape[0],
                                                    float(momentum), int(flags)))

    def pointer_on_device(self, ptr):
        return int(self.lib.klstm_pointer_on_device(self.h, ptr))

    def bind_grad_blob(self, t):
        """Use the torch CUDA float32 tensor `t` (num_params elements, contiguous) as the gradient blob."""
        if t is None:
            self._chk(self.lib.klstm_bind_grad_blob(self.h, None))
            self._bound = None
            return
        assert t.is_cuda and t.is_contiguous() and t.numel() == self.num_params
        self._chk(self.lib.klstm_bind_grad_blob(self.h, t.data_ptr()))
        self._bound = t

    def allreduce_grads(self, comm):
        """In-place fp32 sum of the gradient blob over the ranks of `comm` (a RcclComm), on the engine's stream."""
        self._chk(self.lib.klstm_allreduce_grads(self.h, comm.handle))

    def apply_momentum(self, momentum):
        self._chk(self.lib.klstm_apply_momentum(self.h, float(momentum)))

    def update(self, learn_rate, clip_grad=0.0):
        self._chk(self.lib.klstm_update(self.h, float(learn_rate), float(clip_grad)))

    def synchronize(self):
        self._chk(self.lib.klstm_synchronize(self.h))

    def activations(self, which=0):
        """Reference-layout slab [(T+2)S, 7C+R] of the last propagate (0) / backpropagate (1)."""
        out = np.empty(((self.T + 2) * self.S, self.W), np.float32)
        self._chk(self.lib.klstm_get_activations_host(self.h, which, out.ctypes.data))
        return out

    def set_option(self, key, value):
        self._chk(self.lib.klstm_set_option(self.h, key.encode(), int(value)))
        self.__dict__.setdefault("options", {})[key] = int(value)       # (what the caller asked for: dp.py looks at "bf16")

    def profile_query(self, kernel):
        tot, n = ctypes.c_double(), ctypes.c_long()
        self._chk(self.lib.klstm_profile_query(self.h, kernel.encode(), ctypes.byref(tot), ctypes.byref(n)))
        return tot.value, n.value


class RcclComm:
    """An RCCL communicator created through the C-ABI (klstm_comm_*): the data-path collective of data-parallel training
    is issued by libklstm.so itself (klstm_allreduce_grads / klstm_allreduce_buffer), not by the launcher.
    `exchange(id_bytes_or_None) -> id_bytes` hands rank 0's 128-byte id to every rank (e.g. a torch.distributed / MPI
    broadcast, or a file); with nranks == 1 no exchange is needed."""

    @staticmethod
    def unique_id():
        """128-byte id for a new communicator (call on one rank, hand the bytes to all)."""
        buf = ctypes.create_string_buffer(128)
        _chk(load_library().klstm_comm_get_unique_id(buf))
        return buf.raw

    def __init__(self, nranks, rank, device=0, exchange=None, uid=None):
        self.lib = load_library()
        if uid is not None:
            buf = ctypes.create_string_buffer(bytes(uid), 128)
        else:
            buf = ctypes.create_string_buffer(128)
            if rank == 0:
                _chk(self.lib.klstm_comm_get_unique_id(buf))
            if nranks > 1:
                raw = exchange(buf.raw if rank == 0 else None)
                buf = ctypes.create_string_buffer(bytes(raw), 128)
        h = ctypes.c_void_p()
        _chk(self.lib.klstm_comm_init_rank(int(device), int(nranks), int(rank), buf, ctypes.byref(h)))
        self.handle, self.nranks, self.rank = h, nranks, rank

    def count(self):
        """ncclCommCount: the number of ranks this communicator spans."""
        n = ctypes.c_int(0)
        _chk(self.lib.klstm_comm_count(self.handle, ctypes.byref(n)))
        return n.value

    def allreduce(self, t, stream=None):
        """In-place fp32 sum of a contiguous CUDA float32 tensor over the ranks (torch's current stream by default)."""
        import torch
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        sp = ctypes.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        _chk(self.lib.klstm_allreduce_buffer(t.data_ptr(), t.numel(), self.handle, sp))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.klstm_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OneshotAllreduce:
    """klstm_oneshot_*: the one-shot all-reduce over peer-mapped blobs (klstm_oneshot.hip).  Prepared and OFF by default: never run
    across devices.  `blob` is a contiguous CUDA float32 tensor (kept alive here); exchange() is the launcher's job: it takes a
    function that all-gathers a bytes object over the ranks (torch.distributed.all_gather_object, MPI ...)."""
    HANDLE = 80

    def __init__(self, blob, device=0):
        import torch
        assert blob.is_cuda and blob.dtype == torch.float32 and blob.is_contiguous()
        self.lib, self.blob = load_library(), blob
        self.h = ctypes.c_void_p()
        self._chk(self.lib.klstm_oneshot_create(device, blob.data_ptr(), blob.numel(), ctypes.byref(self.h)))

    def _chk(self, st):
        if st != 0:
            raise KlstmError(st, (self.lib.klstm_oneshot_last_error() or b"").decode())

    def export(self):
        hb, hf = ctypes.create_string_buffer(self.HANDLE), ctypes.create_string_buffer(self.HANDLE)
        self._chk(self.lib.klstm_oneshot_export(self.h, hb, hf))
        return hb.raw + hf.raw

    def connect(self, rank, nranks, all_handles):
        """all_handles: the export() bytes of every rank, in rank order (this rank's own entry is not opened)"""
        hb = ctypes.create_string_buffer(b"".join(x[:self.HANDLE] for x in all_handles), self.HANDLE * nranks)
        hf = ctypes.create_string_buffer(b"".join(x[self.HANDLE:] for x in all_handles), self.HANDLE * nranks)
        self._chk(self.lib.klstm_oneshot_connect(self.h, rank, nranks, hb, hf))

    def allreduce(self, stream=None, timeout_ms=2000):
        self._chk(self.lib.klstm_oneshot_allreduce(self.h, _sp(stream), timeout_ms))

    def allreduce_engine(self, engine, timeout_ms=2000):
        """on the engine's own stream, behind its gradient products (the group must sit on engine.grad_blob_tensor())"""
        engine._chk(self.lib.klstm_allreduce_grads_oneshot(engine.h, self.h, timeout_ms))

    def status(self):
        v = ctypes.c_uint(0)
        self._chk(self.lib.klstm_oneshot_status(self.h, ctypes.byref(v)))
        return v.value

    def close(self):
        if self.h:
            self.lib.klstm_oneshot_destroy(self.h)
            self.h = ctypes.c_void_p()


def time_shift(x, out, shift, stream=None):
    """TimeShift::PropagateFnc / TransmitComponent (shift = 0) on torch CUDA tensors (rows = frames)."""
    lib = load_library()
    assert x.is_cuda and out.is_cuda and x.shape == out.shape and x.stride(1) == 1 and out.stride(1) == 1
    sp = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
    st = lib.klstm_time_shift(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), out.stride(0), int(shift), sp)
    if st != 0:
        raise KlstmError(st, lib.klstm_last_error().decode())


def _sp(stream):
    return ctypes.c_void_p(stream.cuda_stream) if stream is not None else None


def _chk(st):
    if st != 0:
        raise KlstmError(st, load_library().klstm_last_error().decode())


def affine_propagate(x, W, bias, out, stream=None):
    """out = x @ W.T + bias on torch CUDA tensors (W [out_dim, in_dim] contiguous)."""
    lib = load_library()
    assert W.is_contiguous() and bias.is_contiguous() and x.stride(1) == 1 and out.stride(1) == 1
    _chk(lib.klstm_affine_propagate(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), W.data_ptr(), bias.data_ptr(),
                                    out.data_ptr(), W.shape[0], out.stride(0), _sp(stream)))


def affine_backpropagate(out_diff, W, in_diff, stream=None):
    lib = load_library()
    _chk(lib.klstm_affine_backpropagate(out_diff.data_ptr(), out_diff.shape[0], W.shape[0], out_diff.stride(0), W.data_ptr(),
                                        W.shape[1], in_diff.data_ptr(), in_diff.stride(0), _sp(stream)))


def affine_update(x, out_diff, W, bias, W_corr, bias_corr, lr, lr_bias, momentum, stream=None):
    lib = load_library()
    _chk(lib.klstm_affine_update(x.data_ptr(), x.stride(0), out_diff.data_ptr(), out_diff.stride(0), x.shape[0], W.shape[1],
                                 W.shape[0], W.data_ptr(), bias.data_ptr(), W_corr.data_ptr(), bias_corr.data_ptr(),
                                 float(lr), float(lr_bias), float(momentum), _sp(stream)))


def affine_gradient(x, out_diff, W_grad, bias_grad, stream=None):
    """Pure local gradient of an AffineTransform: W_grad = out_diff^T x, bias_grad = colsum(out_diff)."""
    lib = load_library()
    _chk(lib.klstm_affine_gradient(x.data_ptr(), x.stride(0), out_diff.data_ptr(), out_diff.stride(0), x.shape[0],
                                   x.shape[1], out_diff.shape[1], W_grad.data_ptr(), bias_grad.data_ptr(), _sp(stream)))


def sgd_momentum_update(param, corr, grad, momentum, lr, stream=None):
    """corr = momentum*corr + grad ; param -= lr*corr  on flat contiguous CUDA tensors."""
    lib = load_library()
    assert param.is_contiguous() and corr.is_contiguous() and grad.is_contiguous()
    _chk(lib.klstm_sgd_momentum_update(param.data_ptr(), corr.data_ptr(), grad.data_ptr(), param.numel(),
                                       float(momentum), float(lr), _sp(stream)))


def softmax(x, out, stream=None):
    lib = load_library()
    _chk(lib.klstm_softmax(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), out.stride(0), _sp(stream)))


def xent_eval_masked(net_out, target, mask, diff, stream=None, lazy=False, rows_out=None, totals=None):
    """Returns (cross_entropy_sum, correct, valid_frames); fills diff = (net_out - onehot) * mask.
    lazy: no synchronisation -- the three statistics come back as 0-d device tensors (a trainer that reports every N
    minibatches adds them up on the device and reads them once); rows_out = (row_xent, row_correct) buffers to reuse.
    totals: a float64[3] device tensor -- the statistics are ADDED to it by one small launch (klstm_xent_accumulate) and
    (None, None, None) comes back: nothing else runs per minibatch."""
    import torch
    lib = load_library()
    assert target.dtype == torch.int32 and mask.dtype == torch.float32
    rows = net_out.shape[0]
    rx, rc = rows_out if rows_out is not None else (torch.empty(rows, device=net_out.device), torch.empty(rows, device=net_out.device))
    _chk(lib.klstm_xent_eval_masked(net_out.data_ptr(), rows, net_out.shape[1], net_out.stride(0), target.data_ptr(),
                                    mask.data_ptr(), diff.data_ptr(), diff.stride(0), rx.data_ptr(), rc.data_ptr(), _sp(stream)))
    if totals is not None:
        assert totals.dtype == torch.float64 and totals.numel() == 3 and totals.is_contiguous()
        _chk(lib.klstm_xent_accumulate(rx.data_ptr(), rc.data_ptr(), mask.data_ptr(), rows, totals.data_ptr(), _sp(stream)))
        return None, None, None
    if lazy:
        return rx.sum(dtype=torch.float64), rc.sum(), mask.sum()
    if stream is not None:
        stream.synchronize()
    else:
        torch.cuda.synchronize()
    return float(rx.double().sum().item()), int(rc.sum().item()), int(mask.sum().item())


def softmax_xent_masked(net_in, target, mask, diff, post=None, stream=None, lazy=False, rows_out=None, totals=None):
    """Softmax + Xent::EvalMasked in one pass over the rows (klstm_softmax_xent_masked): same results as softmax() followed by
    xent_eval_masked(), the posterior matrix only written when `post` is given (narrow or unaligned rows need it as the buffer
    between the two kernels).  Statistics as xent_eval_masked (lazy / totals)."""
    import torch
    lib = load_library()
    assert target.dtype == torch.int32 and mask.dtype == torch.float32
    rows = net_in.shape[0]
    rx, rc = rows_out if rows_out is not None else (torch.empty(rows, device=net_in.device), torch.empty(rows, device=net_in.device))
    if totals is not None:
        assert totals.dtype == torch.float64 and totals.numel() == 3 and totals.is_contiguous()
    _chk(lib.klstm_softmax_xent_masked(net_in.data_ptr(), rows, net_in.shape[1], net_in.stride(0), post.data_ptr() if post is not None else None,
                                       post.stride(0) if post is not None else 0, target.data_ptr(), mask.data_ptr(), diff.data_ptr(),
                                       diff.stride(0), rx.data_ptr(), rc.data_ptr(), totals.data_ptr() if totals is not None else None,
                                       _sp(stream)))
    if totals is not None:
        return None, None, None                            # (added to `totals` inside the same launch)
    if lazy:
        return rx.sum(dtype=torch.float64), rc.sum(), mask.sum()
    if stream is not None:
        stream.synchronize()
    else:
        torch.cuda.synchronize()
    return float(rx.double().sum().item()), int(rc.sum().item()), int(mask.sum().item())


def xent_eval_masked_post(net_out, post, mask, diff, stream=None):
    """Xent::EvalMasked for general posteriors: post = one list of (pdf, weight) pairs per frame.  Returns
    (cross_entropy_sum, target_entropy_sum, correct, valid_frames); fills diff = (net_out - target) * mask."""
    import torch
    lib = load_library()
    rows, cols = net_out.shape
    assert len(post) == rows and mask.dtype == torch.float32
    off, pdf, w = [0], [], []
    for fr in post:
        for p_, w_ in fr:
            if not 0 <= p_ < cols:              # nnet-loss.cc:89-92 raises while it builds the dense matrix
                raise KlstmError(-1, f"Posterior pdf-id out of NN-output dimension: nn-outputs {cols}, posterior pdf-id {p_}")
            pdf.append(int(p_)); w.append(float(w_))
        off.append(len(pdf))
    dev = net_out.device
    off_d = torch.tensor(off, dtype=torch.int32, device=dev)
    pdf_d = torch.tensor(pdf or [0], dtype=torch.int32, device=dev)
    w_d = torch.tensor(w or [0.0], dtype=torch.float32, device=dev)
    rx = torch.empty(rows, device=dev); re_ = torch.empty(rows, device=dev); rc = torch.empty(rows, device=dev)
    _chk(lib.klstm_xent_eval_masked_post(net_out.data_ptr(), rows, cols, net_out.stride(0), off_d.data_ptr(), pdf_d.data_ptr(),
                                         w_d.data_ptr(), mask.data_ptr(), diff.data_ptr(), diff.stride(0), rx.data_ptr(),
                                         re_.data_ptr(), rc.data_ptr(), _sp(stream)))
    if stream is not None:
        stream.synchronize()
    else:
        torch.cuda.synchronize()
    return float(rx.double().sum().item()), float(re_.double().sum().item()), int(rc.sum().item()), int((mask == 1).sum().item())


def debug_gemm_bf16_nt2(jobs, force_nj=0, force_ks=0, stream=None, copies=None):
    """klstm_debug_gemm_bf16_nt2 / _nt2h (klstm.h): jobs = [(A [M x K], B [N x K], C [M x N], bias or None, add or None), ...] (one or
    two torch CUDA fp32 tensors each, row strides honoured): C = A B^T (+ bias) (+ add), operands rounded to bf16, fp32 accumulate, one
    launch.  copies = [(Ah, Bh), ...]: torch.bfloat16 tensors of A's and B's shapes and row strides -- the kernel reads them instead
    (LDS-DMA).  Returns the plan that ran (nj, ks, output tiles)."""
    lib = load_library()
    n = len(jobs)
    mnk = (ctypes.c_int * (3 * n))()
    ptrs = (ctypes.c_void_p * (5 * n))()
    lds = (ctypes.c_int * (4 * n))()
    for q, (A, B, C, bias, add) in enumerate(jobs):
        assert A.stride(1) == 1 and B.stride(1) == 1 and C.stride(1) == 1 and A.shape[1] == B.shape[1]
        mnk[3 * q:3 * q + 3] = [A.shape[0], B.shape[0], A.shape[1]]
        ptrs[5 * q:5 * q + 5] = [A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr() if bias is not None else None,
                                 add.data_ptr() if add is not None else None]
        lds[4 * q:4 * q + 4] = [A.stride(0), B.stride(0), C.stride(0), add.stride(0) if add is not None else 0]
    plan = (ctypes.c_int * 3)()
    if copies is not None:
        cp = (ctypes.c_void_p * (2 * n))()
        for q, (Ah, Bh) in enumerate(copies):
            A, B = jobs[q][0], jobs[q][1]
            assert Ah.element_size() == 2 and Bh.element_size() == 2 and Ah.shape == A.shape and Bh.shape == B.shape
            assert Ah.stride() == A.stride() and Bh.stride() == B.stride()
            cp[2 * q:2 * q + 2] = [Ah.data_ptr(), Bh.data_ptr()]
        st = lib.klstm_debug_gemm_bf16_nt2h(n, mnk, ptrs, lds, cp, force_nj, force_ks, ctypes.c_void_p(stream) if stream else None, plan)
    else:
        st = lib.klstm_debug_gemm_bf16_nt2(n, mnk, ptrs, lds, force_nj, force_ks, ctypes.c_void_p(stream) if stream else None, plan)
    if st != 0:
        raise KlstmError(st, lib.klstm_last_error().decode())
    return tuple(plan)
