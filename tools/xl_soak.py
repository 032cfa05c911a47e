"""Soak run of the per-XCD chains (klstm_persist_xl.hip, bf16 operands, 32 streams x 1024 cells): two engines fed the same data must
stay BIT-IDENTICAL -- which workgroup takes which slot of an XCC group changes from launch to launch, the arithmetic must not -- and no
launch may give up.  Usage: xl_soak.py [seconds] [streams] [bptt flags: 2 = fused gradient + Update epilogue (default), 0 = separate passes] [cells = 1024] [inputs = 512].
The batched d_r + in_diff products read the bf16 operand copies (klstm_gemm16.hip, LDS-DMA form) from the second minibatch on."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
from oracle.oracle import make_params
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 2
I, C, R, T = (int(sys.argv[5]) if len(sys.argv) > 5 else 512), (int(sys.argv[4]) if len(sys.argv) > 4 else 1024), 512, 20
p = make_params(I, C, R, scale=0.02, seed=3)
stream = torch.cuda.Stream()
es = []
for _ in range(2):
    e = k.Engine(I, C, R, S, stream=stream); e.set_option("bf16", 1); e.set_params(p); es.append(e)
nchunk = 8
x = torch.randn(nchunk, T * S, I, device="cuda"); od = 0.1 * torch.randn(nchunk, T * S, R, device="cuda")
outs = [torch.empty(T * S, R, device="cuda") for _ in es]; inds = [torch.empty(T * S, I, device="cuda") for _ in es]
torch.cuda.synchronize()
t0 = time.time(); n = 0
with torch.cuda.stream(stream):
    while time.time() - t0 < secs:
        for _ in range(200):
            c = n % nchunk
            for e, o, d in zip(es, outs, inds):
                if c == 0: e.reset([1] * S)
                e.propagate(x[c], o); e.backpropagate(x[c], od[c], d, 0.9, flags); e.update(1e-6)
            n += 1
        for e in es: e.synchronize()
        same = torch.equal(outs[0], outs[1]) and torch.equal(inds[0], inds[1]) and np.array_equal(es[0].get_params(), es[1].get_params())
        if not same:
            print("DIVERGED after", n, "minibatches"); sys.exit(1)
g = [e.profile_query("persist_giveups")[1] for e in es]
cp = [e.profile_query("gemm_copies_launches")[1] for e in es]
print("%d/%d/%d S=%d flags=%d: %d minibatches x 2 engines through the per-XCD chains, bit-identical throughout, give-ups %s, launches that read the bf16 copies %s (%.0f s)" % (I, C, R, S, flags, n, g, cp, time.time() - t0))
