"""Random shapes through the output-layer ops (klstm_affine_propagate / _backpropagate / _gradient / _update) around the dispatch
limits of the f16 x 2 kernels (rows <= 80 / 96, widths 2048 / 4096 / 8192), strided views, against float64.  Prints the worst
relative errors; exits 1 on a miss."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kaldi_lstm_amd as k
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"propagate": 0.0, "in_diff": 0.0, "gradient": 0.0, "bias_grad": 0.0, "update_W": 0.0, "update_corr": 0.0, "update_bias": 0.0}
def rel(a, b): return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
bad = 0
for it in range(n):
    N = int(rng.choice([1, 3, 16, 17, 37, 64, 80, 81, 96, 97, 128]))
    K = 4 * int(rng.randint(16, 257))                    # in_dim 64 .. 1024
    M = 4 * int(rng.choice([rng.randint(500, 2600), rng.randint(2000, 5200)]))     # out_dim 2000 .. 20800
    x = torch.randn(N, K + 4, device="cuda")[:, :K]
    W = torch.randn(M, K, device="cuda") * 0.05; b = torch.randn(M, device="cuda") * 0.1
    diff = (torch.rand(N, M + 8, device="cuda") - 0.5)[:, :M]
    out = torch.full((N, M + 4), 3.0, device="cuda"); ind = torch.full((N, K + 4), 3.0, device="cuda")
    gW = torch.empty(M, K, device="cuda"); gb = torch.empty(M, device="cuda")
    Wc = torch.randn(M, K, device="cuda") * 0.01; bc = torch.randn(M, device="cuda") * 0.01
    W2, b2 = W.clone(), b.clone()
    xd, Wd, dd = x.double(), W.double(), diff.double()
    k.affine_propagate(x, W, b, out[:, :M])
    k.affine_backpropagate(diff, W, ind[:, :K])
    k.affine_gradient(x, diff, gW, gb)
    corr_ref = 0.9 * Wc.double() + dd.t() @ xd; bc_ref = 0.9 * bc.double() + dd.sum(0)
    W_ref = Wd - 1e-3 * corr_ref; b_ref = b.double() - 2e-3 * bc_ref
    k.affine_update(x, diff, W2, b2, Wc, bc, 1e-3, 2e-3, 0.9)
    torch.cuda.synchronize()
    e = {"propagate": rel(out[:, :M], xd @ Wd.t() + b.double()), "in_diff": rel(ind[:, :K], dd @ Wd), "gradient": rel(gW, dd.t() @ xd),
         "bias_grad": rel(gb, dd.sum(0)), "update_W": rel(W2, W_ref), "update_corr": rel(Wc, corr_ref), "update_bias": rel(b2, b_ref)}
    ok = all(v <= 2e-5 for v in e.values()) and bool((out[:, M:] == 3.0).all()) and bool((ind[:, K:] == 3.0).all())
    for q in e: worst[q] = max(worst[q], e[q])
    if not ok:
        bad += 1
        print("MISS rows %d in %d out %d:" % (N, K, M), {q: "%.1e" % v for q, v in e.items()}, flush=True)
print("%d shapes, %d misses; worst relative errors:" % (n, bad), {q: "%.1e" % v for q, v in worst.items()})
sys.exit(1 if bad else 0)
