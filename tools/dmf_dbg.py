"""Diagnostic: debug_chain timings of one kernel under the engine's debug variants (option "dmf_dbg")."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
I, C, R, T, S = 40, 800, 512, 20, 4
e = k.Engine(I, C, R, S); e.set_option("fold", 1)
rng = np.random.RandomState(7)
e.set_params(((rng.rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
e.propagate(x, out); e.backpropagate(x, od, ind, 0.9); e.synchronize()
lib = e.lib
lib.klstm_debug_chain.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
what = sys.argv[1] if len(sys.argv) > 1 else "dmf"
for v in [int(a) for a in sys.argv[2:]] or [0, 1, 4, 5]:
    e.set_option("dmf_dbg", v)
    us = ctypes.c_float(); lib.klstm_debug_chain(e.h, what.encode(), 50, ctypes.byref(us))
    print("%s dbg=%d: %.2f us" % (what, v, us.value))
