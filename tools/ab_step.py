"""One process = one library build (KLSTM_LIB_PATH): the headline minibatch (40/800/512, T = 20, fwd + BPTT + fused Update, plain
launches) at S streams: wall-clock us per minibatch over N steps and the device time of each launch.  A-B of builds: run it
alternately under the two libraries (tools/build_variant.sh)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k

I, C, R, T = 40, 800, 512, 20
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opts = [a.split("=") for a in sys.argv[2:]]
stream = torch.cuda.Stream()
e = k.Engine(I, C, R, S, stream=stream)
rng = np.random.RandomState(7)
e.set_params(((rng.rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
for kk, v in opts:
    e.set_option(kk, int(v))
xs = [torch.randn(T * S, I, device="cuda") for _ in range(8)]
od = 0.1 * torch.randn(T * S, R, device="cuda")
out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
with torch.cuda.stream(stream):
    def step(i):
        e.propagate(xs[i & 7], out); e.backpropagate(xs[i & 7], od, ind, 0.9, 2); e.update(1e-5)
    for i in range(50): step(i)
    e.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter(); N = 2000
        for i in range(N): step(i)
        e.synchronize()
        res.append(round((time.perf_counter() - t0) / N * 1e6, 2))
    e.set_option("profile", 1)
    for i in range(3): step(i)
    e.profile_query("k_fold"); e.set_option("profile", 1)
    for i in range(20): step(i)
    kern = {}
    for name in ("k_fwd_persist", "k_bwd_persist", "k_tail_reduce", "k_grads_update", "k_fold", "k_gates_step", "k_gates_fold", "k_dmf_step", "k_grads", "k_update_repack"):
        tot, n = e.profile_query(name)
        if n: kern[name] = round(tot / n, 2)
print(json.dumps({"lib": os.environ.get("KLSTM_LIB_PATH", "default"), "S": S, "opts": opts, "us_per_minibatch": res, "kernels_us": kern}))
e.close()
