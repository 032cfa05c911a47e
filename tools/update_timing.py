import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import kaldi_lstm_amd as k
I, C, R, T, S = 40, 800, 512, 20, 4
stream = torch.cuda.Stream()
e = k.Engine(I, C, R, S, stream=stream)
e.set_params(((np.random.RandomState(7).rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
e.set_option("graph", 0)
x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
with torch.cuda.stream(stream):
    for flags in (0, 1):
        def step():
            e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, flags)
            if flags == 1: e.apply_momentum(0.9)
            e.update(1e-5)
        for _ in range(5): step()
        e.set_option("profile", 1)
        for _ in range(3): step()
        for n in ("k_grads", "k_update_repack"): e.profile_query(n)
        e.set_option("profile", 1)
        for _ in range(10): step()
        print("flags", flags, {n: round(e.profile_query(n)[0] / 10, 2) for n in ("k_grads", "k_update_repack")})
        e.set_option("profile", 0)
