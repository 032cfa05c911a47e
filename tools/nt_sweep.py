import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import torch, numpy as np
import kaldi_lstm_amd as k
N, K, M = 80, 512, 16624
x = torch.randn(N, K, device="cuda"); W = 0.01 * torch.randn(M, K, device="cuda"); b = torch.randn(M, device="cuda")
out = torch.empty(N, M, device="cuda")
e = k.Engine(40, 64, 32, 4)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
ref = x @ W.t() + b
for shape in (11, 12, 21, 22, 41):
    e.set_option("direct_nt_shape", shape)
    us = t(lambda: k.affine_propagate(x, W, b, out))
    print("NI=%d waves=%d: %.1f us, max err %.2e" % (shape // 10, shape % 10, us, (out - ref).abs().max().item()))
e.set_option("fold_direct", 0)
print("tiled kernel: %.1f us" % t(lambda: k.affine_propagate(x, W, b, out)))
