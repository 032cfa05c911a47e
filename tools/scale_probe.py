"""Diagnostic: minibatch time vs NumStream, and a whole-utterance (T=1000, S=1) run."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
from oracle.oracle import make_params
I, C, R = 40, 800, 512
FL = 6 * (4 * C * I + 4 * C * R + R * C)
def run(S, T, n=50, small_max=None, bf16=0, dims=None):
    global I, C, R, FL
    if dims: I, C, R = dims; FL = 6 * (4 * C * I + 4 * C * R + R * C)
    e = k.Engine(I, C, R, S)
    e.set_option("bf16", bf16)
    if small_max is not None:
        e.set_option("small_max", small_max)
    e.set_params(make_params(I, C, R, 0.01, 7))
    x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
    out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
    torch.cuda.synchronize()
    def fbu(): e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)   # (flags 2: the Update follows, klstm.h)
    t0 = time.perf_counter(); fbu(); e.synchronize(); first = time.perf_counter() - t0
    for _ in range(3): fbu()
    e.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fbu()
    e.synchronize(); dt = (time.perf_counter() - t0) / n
    print("S=%3d T=%4d : %8.1f us/minibatch  %9.0f frames/s  %6.2f TF/s  (first call incl. graph capture %.0f ms)" %
          (S, T, dt * 1e6, T * S / dt, T * S / dt * FL / 1e12, first * 1e3), flush=True)
    e.close()
if len(sys.argv) > 1 and sys.argv[1] == "small":
    for S in (8, 12, 16):
        for sm in (4, 16):
            print("small_max=%d " % sm, end=""); run(S, 20, small_max=sm)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    for S in (4, 8, 16, 32, 64, 256):
        for b in (0, 1):
            print("bf16=%d " % b, end=""); run(S, 20, bf16=b)
    for b in (0, 1):                                  # BASELINE.json configs[4] inner layer, 32 streams per GPU
        print("c5 layer 512->1024/512 bf16=%d " % b, end=""); run(32, 20, bf16=b, dims=(512, 1024, 512))
    sys.exit(0)
for S in (1, 2, 4, 8, 12, 16, 32, 64, 128, 256):
    run(S, 20)
run(1, 1000, n=5)
