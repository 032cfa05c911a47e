"""Do latency-bound launch-per-step chains of INDEPENDENT engines overlap when they are issued on separate streams?  (Upper bound
of what a layer-wavefront schedule of configs[4] could gain.)  Three 512 -> 1024/512 bf16 engines at 32 streams, each on its own
input: one stream after the other vs three streams at once."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
I, C, R, S, T, NE = 512, 1024, 512, 32, 20, 3
def make(streams):
    es = []
    for i in range(NE):
        e = k.Engine(I, C, R, S, stream=streams[i]); e.set_option("bf16", 1)
        e.set_params(((np.random.RandomState(7 + i).rand(e.num_params) - 0.5) * 0.04).astype(np.float32)); es.append(e)
    return es
x = [torch.randn(T * S, I, device="cuda") for _ in range(NE)]; od = [0.1 * torch.randn(T * S, R, device="cuda") for _ in range(NE)]
out = [torch.empty(T * S, R, device="cuda") for _ in range(NE)]; ind = [torch.empty(T * S, I, device="cuda") for _ in range(NE)]
for label, streams, graph in (("one stream", [torch.cuda.Stream()] * NE, 0), ("three streams", [torch.cuda.Stream() for _ in range(NE)], 0),
                              ("one stream, graph per call", [torch.cuda.Stream()] * NE, 2),
                              ("three streams, graph per call", [torch.cuda.Stream() for _ in range(NE)], 2)):
    es = make(streams)
    for e in es: e.set_option("graph", graph)
    torch.cuda.synchronize()
    def step():
        for i, e in enumerate(es):
            with torch.cuda.stream(streams[i]):
                e.propagate(x[i], out[i])
        for i, e in enumerate(es):
            with torch.cuda.stream(streams[i]):
                e.backpropagate(x[i], od[i], ind[i], 0.9, 0); e.update(1e-5)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): step()
    torch.cuda.synchronize()
    print("%-30s %.0f us per round of %d independent layer minibatches" % (label, (time.perf_counter() - t0) / n * 1e6, NE), flush=True)
    for e in es: e.close()
