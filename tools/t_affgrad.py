"""Device time of the output layer's gradient side at few rows: klstm_affine_gradient (W_grad = out_diff^T x + column sums),
klstm_sgd_momentum_update over the layer, and the fused klstm_affine_update."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kaldi_lstm_amd as k
s = torch.cuda.Stream()
e = k.Engine(40, 64, 32, 4)
def t(label, fn):
    with torch.cuda.stream(s):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(50): fn()
        e1.record(s); e1.synchronize()
        print("%-58s %.1f us" % (label, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
for N, K, M in ((80, 512, 16624), (37, 512, 16624), (80, 512, 9000), (160, 512, 16624)):
    x = torch.randn(N, K, device="cuda"); diff = torch.randn(N, M, device="cuda") * 0.1
    W = torch.randn(M, K, device="cuda") * 0.1; b = torch.zeros(M, device="cuda")
    gW = torch.empty(M, K, device="cuda"); gb = torch.empty(M, device="cuda")
    Wc = torch.zeros(M, K, device="cuda"); bc = torch.zeros(M, device="cuda")
    torch.cuda.synchronize()
    ref = diff.double().t() @ x.double()
    for on in (1, 0):
        e.set_option("outer_f16", on)
        t("rows %d in %d out %d: gradient (%s)" % (N, K, M, "f16 x 2 planes" if on else "fp32 tiles"), lambda: k.affine_gradient(x, diff, gW, gb, s))
        print("   vs fp64: max |difference| %.2e of %.1f; column sums %.2e" % ((gW.double() - ref).abs().max().item(), ref.abs().max().item(),
              (gb.double() - diff.double().sum(0)).abs().max().item()))
    e.set_option("outer_f16", 1)
    t("rows %d in %d out %d: momentum + update" % (N, K, M), lambda: k.sgd_momentum_update(W.view(-1), Wc.view(-1), gW.view(-1), 0.9, 1e-4, s))
    t("rows %d in %d out %d: fused gradient + momentum + update" % (N, K, M), lambda: k.affine_update(x, diff, W, b, Wc, bc, 1e-4, 1e-4, 0.9, s))
