"""Soak run of the persistent chain: two engines fed the same data must stay BIT-IDENTICAL (the in-launch exchange is deterministic:
fixed summation order, no atomics on data), over many thousands of minibatches -- a rare race in the flag / counter protocol would
show as a divergence or as an expired wait.  Usage: soak.py [streams] [seconds] [persist_verify].  persist_verify = 1: every persistent
call waits on the host-mapped done word its launch's last workgroup writes (round 5) -- thousands of waits, none may hang or miss."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
from oracle.oracle import make_params
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
verify = int(sys.argv[3]) if len(sys.argv) > 3 else 0
I, C, R, T = 40, 800, 512, 20
p = make_params(I, C, R, scale=0.01, seed=3)
stream = torch.cuda.Stream()
es = []
for _ in range(2):
    e = k.Engine(I, C, R, S, stream=stream); e.set_params(p); e.set_option("persist", 2); e.set_option("persist_verify", verify); es.append(e)
nchunk = 16
x = torch.randn(nchunk, T * S, I, device="cuda"); od = 0.1 * torch.randn(nchunk, T * S, R, device="cuda")
outs = [torch.empty(T * S, R, device="cuda") for _ in es]; inds = [torch.empty(T * S, I, device="cuda") for _ in es]
torch.cuda.synchronize()
t0 = time.time(); n = 0
with torch.cuda.stream(stream):
    while time.time() - t0 < secs:
        for _ in range(500):
            c = n % nchunk
            for e, o, d in zip(es, outs, inds):
                if c == 0: e.reset([1] * S)
                e.propagate(x[c], o); e.backpropagate(x[c], od[c], d, 0.9, 2); e.update(1e-6)
            n += 1
        for e in es: e.synchronize()                       # (raises if a wait expired)
        same = torch.equal(outs[0], outs[1]) and torch.equal(inds[0], inds[1]) and np.array_equal(es[0].get_params(), es[1].get_params())
        if not same:
            print("DIVERGED after", n, "minibatches"); sys.exit(1)
gu = [e.profile_query("persist_giveups")[1] for e in es]
print("S=%d persist_verify=%d: %d minibatches x 2 engines, bit-identical throughout, no expired wait, give-ups %s (%.0f s)" % (S, verify, n, gu, time.time() - t0))
