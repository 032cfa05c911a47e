"""Why is a minibatch re-run after a give-up of the many-stream launch not bit-identical to a twin on the launch-per-step chain?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
from oracle.oracle import make_params
I, C, R, S, T = 512, 1024, 512, 32, 20
p = make_params(I, C, R, scale=0.02, seed=15)
rng = np.random.RandomState(16)
x = torch.from_numpy(rng.randn(T * S, I).astype(np.float32)).cuda(); od = torch.from_numpy((0.3 * rng.randn(T * S, R)).astype(np.float32)).cuda()
def run(opts, stall=0):
    e = k.Engine(I, C, R, S); e.set_params(p); e.set_option("bf16", 1)
    for kk, v in opts: e.set_option(kk, v)
    if stall: e.set_option("persist_spin_us", 3000); e.set_option("persist_test_stall_fwd", stall)
    out = torch.empty(T * S, R, device="cuda"); idf = torch.empty(T * S, I, device="cuda")
    e.propagate(x, out); e.backpropagate(x, od, idf, momentum=0.9); e.synchronize()
    res = dict(out=out.cpu().numpy(), idf=idf.cpu().numpy(), Y=e.activations(0), giveups=e.profile_query("persist_giveups")[1])
    e.close()
    return res
a = run([("persist", 0)]); b = run([("persist", 0)]); c = run([], stall=4)
def cmp(n1, r1, n2, r2):
    for key in ("out", "idf", "Y"):
        d = np.abs(r1[key].astype(np.float64) - r2[key]).reshape(T if key != "Y" else T + 2, -1).max(1)
        print(n1, n2, key, "max abs diff per frame:", " ".join("%.1e" % v for v in d))
cmp("twin", a, "twin2", b); print("giveups of the stalled engine:", c["giveups"]); cmp("twin", a, "replayed", c)
W = 7 * C + R
Ya, Yc = a["Y"].reshape(T + 2, S, W), c["Y"].reshape(T + 2, S, W)
for name, lo, hi in (("G", 0, C), ("I", C, 2 * C), ("F", 2 * C, 3 * C), ("O", 3 * C, 4 * C), ("C", 4 * C, 5 * C), ("H", 5 * C, 6 * C), ("M", 6 * C, 7 * C), ("R", 7 * C, W)):
    print(name, " ".join("%.1e" % np.abs(Ya[t, :, lo:hi].astype(np.float64) - Yc[t, :, lo:hi]).max() for t in range(T + 2)))
