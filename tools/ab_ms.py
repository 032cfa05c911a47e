"""One process = one library build (KLSTM_LIB_PATH): the many-stream bf16 layer (512 -> 1024 / 512, S streams, T = 20): us per minibatch
and device time per launch."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
I, C, R, T = 512, 1024, 512, 20
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
opts = [a.split("=") for a in sys.argv[2:]]
stream = torch.cuda.Stream()
e = k.Engine(I, C, R, S, stream=stream)
rng = np.random.RandomState(7)
e.set_params(((rng.rand(e.num_params) - 0.5) * 0.04).astype(np.float32))
e.set_option("bf16", 1)
for kk, v in opts: e.set_option(kk, int(v))
x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
with torch.cuda.stream(stream):
    def step():
        e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 1); e.apply_momentum(0.9); e.update(1e-5)
    for i in range(20): step()
    e.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter(); N = 300
        for i in range(N): step()
        e.synchronize()
        res.append(round((time.perf_counter() - t0) / N * 1e6, 1))
    e.set_option("profile", 1)
    for i in range(3): step()
    e.profile_query("k_grads"); e.set_option("profile", 1)
    for i in range(10): step()
    kern = {}
    for name in ("k_gemm_xproj", "k_fwd_persist_ms", "k_fold_ms", "k_split3", "k_gates_step", "k_proj_step", "k_dr_step", "k_dm_step", "k_dr_step0", "k_grads", "k_update_repack", "k_pack"):
        tot, n = e.profile_query(name)
        if n: kern[name] = (round(tot / n, 2), n // 10)
print(json.dumps({"lib": os.environ.get("KLSTM_LIB_PATH", "default"), "S": S, "opts": opts, "us_per_minibatch": res, "kernels_us": kern}))
e.close()
