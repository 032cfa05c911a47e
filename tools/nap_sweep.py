import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import kaldi_lstm_amd as k
I, C, R, T = 40, 800, 512, 20
S = int(sys.argv[1])
BWD = int(os.environ.get("NAP0_BWD", "0"))      # backward launch's nap0 (0 = its default)
stream = torch.cuda.Stream()
for nap0 in [int(v) for v in sys.argv[2:]]:
    e = k.Engine(I, C, R, S, stream=stream)
    rng = np.random.RandomState(7)
    e.set_params(((rng.rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
    e.set_option("graph", 0); e.set_option("persist_nap0_bwd", BWD); e.set_option("persist_nap0", nap0)
    x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
    out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
    with torch.cuda.stream(stream):
        def step():
            e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
        for _ in range(10): step()
        e.synchronize(); t0 = time.perf_counter()
        for _ in range(200): step()
        e.synchronize(); us = (time.perf_counter() - t0) / 200 * 1e6
    print("S=%d nap0=%d: %.1f us per minibatch" % (S, nap0, us), flush=True)
    e.close()
