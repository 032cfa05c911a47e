import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
import kaldi_lstm_amd as k
import numpy as np
I, C, R, T = 40, 800, 512, 20
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for graph in (2, 0, 2, 0):
    e = k.Engine(I, C, R, S)
    e.set_option("graph", graph)
    rng = np.random.RandomState(7)
    e.set_params(((rng.rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
    x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
    out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
    def fbu(): e.propagate(x, out); e.backpropagate(x, od, ind, 0.9); e.update(1e-5)
    for _ in range(20): fbu()
    e.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fbu()
    e.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("S=%d graph=%d: %.1f us/minibatch" % (S, graph, dt * 1e6))
    e.close()
