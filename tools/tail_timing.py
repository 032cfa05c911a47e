"""Diagnostic: the output tail of BASELINE.json configs[3] (AffineTransform 512 -> 16624 + Softmax + masked Xent) per
minibatch of N = T*S frames, per op (device ops of the C-ABI)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaldi_lstm_amd as k
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
K, M = 512, 16624
x = torch.randn(N, K, device="cuda"); W = 0.01 * torch.randn(M, K, device="cuda"); b = torch.zeros(M, device="cuda")
out = torch.empty(N, M, device="cuda"); post = torch.empty_like(out); diff = torch.empty_like(out)
ind = torch.empty(N, K, device="cuda"); gW = torch.empty_like(W); gb = torch.empty_like(b)
Wc = torch.zeros_like(W); bc = torch.zeros_like(b)
tg = torch.randint(0, M, (N,), device="cuda", dtype=torch.int32); mk = torch.ones(N, device="cuda")
rx = torch.empty(N, device="cuda"); rc = torch.empty(N, device="cuda")
lib = k.load_library()
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("N=%d frames, affine %d -> %d" % (N, K, M))
print("affine_propagate    : %7.1f us" % t(lambda: k.affine_propagate(x, W, b, out)))
print("softmax             : %7.1f us" % t(lambda: k.softmax(out, post)))
print("xent (kernel only)  : %7.1f us" % t(lambda: lib.klstm_xent_eval_masked(post.data_ptr(), N, M, M, tg.data_ptr(), mk.data_ptr(), diff.data_ptr(), M, rx.data_ptr(), rc.data_ptr(), None)))
tot = torch.zeros(3, device="cuda", dtype=torch.float64)
print("softmax + xent, one pass (rows only)       : %7.1f us" % t(lambda: k.softmax_xent_masked(out, tg, mk, diff, lazy=True, rows_out=(rx, rc))))
print("softmax + xent, one pass (+ device totals) : %7.1f us" % t(lambda: k.softmax_xent_masked(out, tg, mk, diff, rows_out=(rx, rc), totals=tot)))
print("affine_backpropagate: %7.1f us" % t(lambda: k.affine_backpropagate(diff, W, ind)))
print("affine_gradient     : %7.1f us" % t(lambda: k.affine_gradient(x, diff, gW, gb)))
print("sgd_momentum_update : %7.1f us" % t(lambda: k.sgd_momentum_update(W.view(-1), Wc.view(-1), gW.view(-1), 0.9, 1e-5)))
print("affine_update (fused momentum+update, single GPU): %7.1f us" % t(lambda: k.affine_update(x, diff, W, b, Wc, bc, 1e-5, 1e-5, 0.9)))
print("torch (rocBLAS) x@W.T for scale: %7.1f us ; diff@W: %7.1f us ; diff.T@x: %7.1f us" % (t(lambda: torch.mm(x, W.t())), t(lambda: torch.mm(diff, W)), t(lambda: torch.mm(diff.t(), x))))
