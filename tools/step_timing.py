"""Diagnostic: per-phase and per-kernel in-chain timings of the engine (no profiler)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
from oracle.oracle import make_params
I, C, R, T = 40, 800, 512, 20
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
BF = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if len(sys.argv) > 5: I, C, R = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
e = k.Engine(I, C, R, S)
e.set_option("bf16", BF)
FOLD = int(os.environ.get("FOLD", "-1"))
e.set_option("fold", FOLD)
if "FINE" in os.environ: e.set_option("fat_fine", int(os.environ["FINE"]))
if "NT2" in os.environ: e.set_option("small_nt2", int(os.environ["NT2"]))
e.set_params(make_params(I, C, R, 0.01, 7))
x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
torch.cuda.synchronize()
def timeit(fn, n=100):
    for _ in range(10): fn()
    e.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    e.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("S=%d bf16=%d dims %d/%d/%d" % (S, BF, I, C, R))
print("fwd graph        : %8.1f us" % timeit(lambda: e.propagate(x, out)))
def fb(): e.propagate(x, out); e.backpropagate(x, od, ind, 0.9)
def fb2(): e.propagate(x, out); e.backpropagate(x, od, None, 0.9)
def fbu(): e.propagate(x, out); e.backpropagate(x, od, ind, 0.9); e.update(1e-5)
print("fwd+bwd          : %8.1f us" % timeit(fb))
print("fwd+bwd(no indiff): %8.1f us" % timeit(fb2))
print("fwd+bwd+update   : %8.1f us" % timeit(fbu))
lib = e.lib
lib.klstm_debug_chain.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
for w in ("gates_fold", "dmf", "fold", "rbatch", "bwd_tail", "gates", "proj", "gates+proj", "dr", "dm", "dr+dm", "grads", "update", "pack", "pack_fwd"):
    us = ctypes.c_float()
    rc = lib.klstm_debug_chain(e.h, w.encode(), 200, ctypes.byref(us))
    print("chain %-11s: %6.2f us/launch (rc=%d)" % (w, us.value, rc))
