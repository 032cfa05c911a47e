import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import kaldi_lstm_amd as k
from oracle.oracle import make_params
I, C, R, S, T = 512, 1024, 512, 32, 20
lib = k.load_library()
p0 = make_params(I, C, R, 0.02, 7)
def hog(n):
    where = torch.full((2 * n,), -1, dtype=torch.int32).pin_memory()
    torch.cuda.synchronize()
    lib.klstm_debug_occupy(0, n, 30000, None, where.data_ptr())
    while (where.numpy() == -1).any(): time.sleep(0.0005)
for held in (40, 0):
    e = k.Engine(I, C, R, S); e.set_option("bf16", 1); e.set_params(p0); e.set_option("persist_spin_us", 3000)
    t = k.Engine(I, C, R, S); t.set_option("bf16", 1); t.set_params(p0); t.set_option("persist", 0)
    x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
    out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
    for eng in (e, t):
        eng.propagate(x, out); eng.backpropagate(x, od, ind, 0.9, 0); eng.update(1e-4); eng.synchronize()
    torch.cuda.synchronize()
    if held: hog(held)
    t0 = time.perf_counter()
    for _ in range(3):
        e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 0); e.update(1e-4)
    e.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    time.sleep(0.04)
    for _ in range(3):
        t.propagate(x, out); t.backpropagate(x, od, ind, 0.9, 0); t.update(1e-4)
    t.synchronize()
    g, r, d = (e.profile_query(n)[1] for n in ("persist_giveups", "persist_replayed", "persist_dropped"))
    pe, pt = e.get_params(), t.get_params()
    print("per-XCD chains, 32 streams bf16, foreign kernel on %2d CUs: 3 minibatches in %.1f ms, give-ups %d, run again %d, dropped %d, parameters vs launch-per-step twin %.1e"
          % (held, ms, g, r, d, float(np.abs(pe - pt).max() / np.abs(pt).max())), flush=True)
    e.close(); t.close()
