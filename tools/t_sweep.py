"""Device time of the two persistent launches against frames per minibatch: slope = cost of a recurrence step inside the
engine (with r / P / d_r / in_diff on board), intercept = prologue + epilogue of a launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
I, C, R, S = 40, 800, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 4
stream = torch.cuda.Stream()
res = {}
for T in (10, 20, 40, 80):
    e = k.Engine(I, C, R, S, stream=stream)
    e.set_params(((np.random.RandomState(7).rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
    x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
    out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
    with torch.cuda.stream(stream):
        def step():
            e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
        for _ in range(5): step()
        e.set_option("profile", 1)
        for _ in range(3): step()
        for n in ("k_fwd_persist", "k_bwd_persist"): e.profile_query(n)
        e.set_option("profile", 1)
        for _ in range(10): step()
        res[T] = {n: e.profile_query(n)[0] / 10 for n in ("k_fwd_persist", "k_bwd_persist")}
    print("T=%d: fwd %.1f us, bwd %.1f us" % (T, res[T]["k_fwd_persist"], res[T]["k_bwd_persist"]), flush=True)
    e.close()
for n in ("k_fwd_persist", "k_bwd_persist"):
    slope = (res[80][n] - res[20][n]) / 60
    print("%s: %.2f us per step, %.1f us per launch outside the steps (T = 20)" % (n, slope, res[20][n] - 20 * slope))
