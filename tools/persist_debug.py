import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
from oracle.oracle import make_params
I, C, R, S, T = 40, 64, 32, 4, 6
p = make_params(I, C, R, scale=0.3, seed=1)
rng = np.random.RandomState(0)
x = torch.from_numpy(rng.randn(T * S, I).astype(np.float32)).cuda()
od = torch.from_numpy(rng.randn(T * S, R).astype(np.float32)).cuda()
res = []
for persist in (0, 1):
    e = k.Engine(I, C, R, S); e.set_params(p); e.set_option("fold", 1); e.set_option("persist", persist)
    e.set_option("persist_waves", int(sys.argv[1]) if len(sys.argv) > 1 else 16)
    out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
    e.propagate(x, out); e.backpropagate(x, od, ind, 0.0); e.synchronize()
    res.append((e.activations(0), e.activations(1), out.cpu().numpy(), ind.cpu().numpy()))
    e.close()
names = ["G", "I", "F", "O", "C", "H", "M"]
for which, lab in ((0, "fwd"), (1, "bwd")):
    A, B = res[0][which], res[1][which]
    for t in range(0, T + 2):
        row = []
        for g, n in enumerate(names):
            a, b = A[t * S:(t + 1) * S, g * C:(g + 1) * C], B[t * S:(t + 1) * S, g * C:(g + 1) * C]
            row.append("%s %.1e" % (n, np.abs(a - b).max()))
        a, b = A[t * S:(t + 1) * S, 7 * C:], B[t * S:(t + 1) * S, 7 * C:]
        row.append("R %.1e" % np.abs(a - b).max())
        print(lab, "t=%d" % t, " ".join(row))
    if which == 0:
        t = 3
        d = np.abs(A[t * S:(t + 1) * S, 0:C] - B[t * S:(t + 1) * S, 0:C])
        np.set_printoptions(linewidth=200, precision=2)
        print("fwd t=3 G diff per stream (max over cells):", d.max(1))
        print(" per cell:", d.max(0))
        # what m(2) would explain the t=3 result?  compare the persistent G(3) with launch G(3) recomputed after permuting m(2)
        M2 = A[2 * S:3 * S, 6 * C:7 * C]
        print("m(2) stream 0 first 8 cells", M2[0, :8])
