"""Round-6 diagnostic: one engine on the tail-workgroup form of the persistent BPTT launch ("persist_tail" = 1); prints the give-up
remark (status words) if the launch gave up, per-kernel times otherwise."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
from oracle.oracle import make_params
I, C, R = 40, 800, 512
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e = k.Engine(I, C, R, S)
e.set_option("persist", 2)
for a in sys.argv[3:]:
    kk, v = a.split("="); e.set_option(kk, int(v))
e.set_params(make_params(I, C, R, 0.01, 7))
x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
for i in range(6):
    e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5); e.synchronize()
    print(i, "tail_wgs", e.profile_query("persist_tail_wgs")[1], "giveups", e.profile_query("persist_giveups")[1],
          "cooldown", e.profile_query("persist_cooldown")[1], "|", e.lib.klstm_last_error().decode(), flush=True)
e.close()
