import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import kaldi_lstm_amd as k
from oracle.oracle import make_params
I, C, R, S = 512, 1024, 512, 32
for T in (10, 20, 40):
    for xl in (1, 0):
        e = k.Engine(I, C, R, S); e.set_option("bf16", 1); e.set_option("persist_xl", xl); e.set_option("persist_xl_bwd", xl); e.set_option("graph", 0)
        e.set_params(make_params(I, C, R, 0.02, 7))
        x = torch.randn(T * S, I, device="cuda"); out = torch.empty(T * S, R, device="cuda")
        od = 0.1 * torch.randn(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
        for _ in range(3):
            e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
        e.set_option("profile", 1)
        for _ in range(3): e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
        e.profile_query("k_fold_ms"); e.set_option("profile", 1)
        for _ in range(10): e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
        tot, n = e.profile_query("k_fwd_persist_xl" if xl else "k_fwd_persist_ms")
        tb, nb = e.profile_query("k_bwd_persist_xl")
        print("T=%d xl=%d: forward launch %.1f us (%d), BPTT launch %.1f us (%d)" % (T, xl, tot / max(n, 1), n, tb / max(nb, 1), nb), flush=True)
        e.close()
