"""A/B of the recurrence chain at 40/800/512: launch-per-step folded chain vs the persistent weights-resident chain
(option "persist"), per geometry, per stream count: whole minibatch (fwd + BPTT + update) and the per-kernel device
times of the two chain launches.  Diagnostic (docs/DESIGN_rounds_1-4.md section 4 table)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k

I, C, R, T = 40, 800, 512, 20
rows = []
for S in (1, 2, 4, 8):
    for persist, waves, tpw in ((0, 0, 0), (1, 0, 0), (2, 0, 0)):
        stream = torch.cuda.Stream()
        e = k.Engine(I, C, R, S, stream=stream)
        rng = np.random.RandomState(7)
        e.set_params(((rng.rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
        e.set_option("graph", 0); e.set_option("persist", persist); e.set_option("persist_waves", waves); e.set_option("persist_tpw", tpw)
        x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
        out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
        with torch.cuda.stream(stream):
            def step():
                e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
            try:
                for _ in range(10): step()
                e.synchronize()
                t0 = time.perf_counter()
                N = 300
                for _ in range(N): step()
                e.synchronize()
                us = (time.perf_counter() - t0) / N * 1e6
                e.set_option("profile", 1)
                for _ in range(3): step()
                e.profile_query("k_grads_update"); e.profile_query("k_grads"); e.set_option("profile", 1)
                for _ in range(10): step()
                kern = {}
                for name in ("k_gates_step", "k_gates_fold", "k_dmf_step", "k_fwd_persist", "k_bwd_persist", "k_gemm_rbatch", "k_reduce_rbatch", "k_gemm_P", "k_reduce_P",
                             "k_gemm_tail", "k_reduce_tail", "k_grads", "k_grads_update", "k_update_repack", "k_pack", "k_fold", "k_pack_foldx"):
                    tot, n = e.profile_query(name)
                    if n: kern[name] = round(tot / 10, 1)
                e.set_option("profile", 0)
                rows.append(dict(S=S, persist=persist, waves=waves, tpw=tpw, us_per_minibatch=round(us, 1), chain_us=kern))
            except Exception as ex:
                rows.append(dict(S=S, persist=persist, waves=waves, tpw=tpw, error=str(ex)))
            print(rows[-1], flush=True)
        e.close()
json.dump(rows, open("gpurun_out/persist_timing.json", "w"), indent=1)
