"""Device time of klstm_affine_propagate (few rows, wide layer): k_nt_shared_a16 (f16 x 2, the default; option direct_nt_shape = 98), k_nt_resident_a16 (97: A in registers, measured and not the default), the fp32 k_nt_shared_a (99) and k_direct_nt (0)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kaldi_lstm_amd as k
s = torch.cuda.Stream()
e = k.Engine(40, 64, 32, 4)
def t(label, x, W, b, out):
    with torch.cuda.stream(s):
        for _ in range(5): k.affine_propagate(x, W, b, out, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(50): k.affine_propagate(x, W, b, out, s)
        e1.record(s); e1.synchronize()
        print("%-44s %.1f us" % (label, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
for N, K, M in ((80, 512, 16624), (37, 512, 16624), (80, 256, 16624), (80, 512, 9000)):
    x = torch.randn(N, K, device="cuda"); W = torch.randn(M, K, device="cuda") * 0.1; b = torch.randn(M, device="cuda"); out = torch.empty(N, M, device="cuda")
    torch.cuda.synchronize()
    e.set_option("direct_nt_shape", 97)                  # round 6: A resident in registers, K in four quarters (measured, not the default)
    t("rows %d in %d out %d: A resident, f16 x 2 split" % (N, K, M), x, W, b, out)
    rres = out.clone()
    e.set_option("direct_nt_shape", 98)
    t("rows %d in %d out %d: A shared, f16 x 2 split" % (N, K, M), x, W, b, out)
    r16 = out.clone()
    print("   resident vs shared: max |difference| %.2e" % (rres - r16).abs().max().item())
    e.set_option("direct_nt_shape", 99)
    t("rows %d in %d out %d: A shared in LDS" % (N, K, M), x, W, b, out)
    print("   f16-split vs fp32 kernel: max |difference| %.2e" % (r16 - out).abs().max().item())
    ref = out.clone()
    e.set_option("direct_nt_shape", 0)
    t("rows %d in %d out %d: register-direct" % (N, K, M), x, W, b, out)
    e.set_option("direct_nt_shape", 21)
    print("   max |difference| %.2e of %.1f; vs torch %.2e" % ((ref - out).abs().max().item(), out.abs().max().item(), (ref - (x @ W.T + b)).abs().max().item()))
