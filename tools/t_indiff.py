"""Device time of klstm_affine_backpropagate (in_diff = out_diff W, few rows, long contraction) by shape: the skinny kernel pair of
klstm_fold.hip against the tiled split-K pair (option fold_direct = 0)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kaldi_lstm_amd as k
s = torch.cuda.Stream()
e = k.Engine(40, 64, 32, 4)
def t(label, diff, W, ind):
    with torch.cuda.stream(s):
        for _ in range(5): k.affine_backpropagate(diff, W, ind, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(50): k.affine_backpropagate(diff, W, ind, s)
        e1.record(s); e1.synchronize()
        print("%-40s %.1f us" % (label, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
for N, K, M in ((80, 512, 16624), (16, 512, 16624), (80, 128, 16624), (80, 512, 8192)):
    diff = torch.randn(N, M, device="cuda"); W = torch.randn(M, K, device="cuda") * 0.1; ind = torch.empty(N, K, device="cuda")
    torch.cuda.synchronize()
    e.set_option("fold_direct", 1)
    t("rows %d in %d out %d: skinny, f16 x 2 in registers" % (N, K, M), diff, W, ind)
    r16 = ind.clone()
    e.set_option("skinny_f16", 0)
    t("rows %d in %d out %d: skinny, fp32 MFMA" % (N, K, M), diff, W, ind)
    e.set_option("skinny_f16", 1)
    ref = ind.clone()
    r64 = (diff.double() @ W.double())
    print("   vs fp64: f16 x 2 %.2e, fp32 %.2e of %.1f" % ((r16 - r64).abs().max().item(), (ref - r64).abs().max().item(), r64.abs().max().item()))
    e.set_option("fold_direct", 0)
    t("rows %d in %d out %d: tiled split-K" % (N, K, M), diff, W, ind)
    print("   max |difference| %.2e of %.1f" % ((ref - ind).abs().max().item(), ind.abs().max().item()))
