"""Per-kernel device time of one minibatch (fwd + BPTT + update) at a given NumStream: which launches carry the time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bf16 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
I, C, R, T = 40, 800, 512, 20
stream = torch.cuda.Stream()
e = k.Engine(I, C, R, S, stream=stream)
rng = np.random.RandomState(7)
e.set_params(((rng.rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
e.set_option("graph", 0); e.set_option("bf16", bf16)
x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
names = ("k_gemm_xproj", "k_gates_step", "k_proj_step", "k_gates_fold", "k_gemm_rbatch", "k_reduce_rbatch", "k_fwd_persist", "k_dr_step",
         "k_dm_step", "k_dr_step0", "k_dmf_step", "k_gemm_P", "k_reduce_P", "k_gemm_tail", "k_reduce_tail", "k_bwd_persist", "k_grads",
         "k_grads_update", "k_update_repack", "k_pack", "k_fold", "k_pack_foldx", "k_apply_momentum")
with torch.cuda.stream(stream):
    def step():
        e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
    for _ in range(5): step()
    e.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    e.synchronize(); us = (time.perf_counter() - t0) / 50 * 1e6
    e.set_option("profile", 1)
    for _ in range(3): step()
    for n in names: e.profile_query(n)
    e.set_option("profile", 1)
    for _ in range(10): step()
    tot = 0.0
    print("S=%d bf16=%d: %.1f us per minibatch (%.0f frames/s)" % (S, bf16, us, T * S / us * 1e6))
    for n in names:
        t, c = e.profile_query(n)
        if c:
            print("  %-18s %7.1f us per minibatch (%d launches, %.2f us each)" % (n, t / 10, c // 10, t / c)); tot += t / 10
    print("  sum of kernels     %7.1f us" % tot)
e.close()
