"""DESIGN.md 4a (5), VERDICT r03 next #2: does what Xent::EvalMasked does every minibatch -- a pageable device-to-host copy of a few
scalars (google/nnet/nnet-loss.cc:110-141) -- disturb the persistent chain WITHOUT a co-tenant?  (Round 3 saw persistent launches
start with ~30 of 200 workgroups missing while a foreign kernel's flags were polled with pageable D2H copies.)
N minibatches of the headline workload per mode: no copy / pageable 12-byte D2H per minibatch / pinned non-blocking copy + event;
reports us per minibatch, the engine's give-up counters and the longest single minibatch (a give-up would show as a >= 3 ms stall:
persist_spin_us is set to 3000)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k

I, C, R, T, S = 40, 800, 512, 20, 4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rows = []
for mode in ("none", "pageable_d2h", "pinned_async"):
    stream = torch.cuda.Stream()
    e = k.Engine(I, C, R, S, stream=stream)
    rng = np.random.RandomState(7)
    e.set_params(((rng.rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
    e.set_option("persist", 2); e.set_option("persist_spin_us", 3000)
    x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
    out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
    stats = torch.zeros(3, device="cuda"); pinned = torch.zeros(3).pin_memory()
    worst = 0.0
    with torch.cuda.stream(stream):
        for i in range(20):
            e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
        e.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            t1 = time.perf_counter()
            e.propagate(x, out)
            if mode == "pageable_d2h":
                stats.copy_(out[0, :3]); h = stats.cpu()          # what the loss does: a synchronising copy of three scalars into pageable memory
            elif mode == "pinned_async":
                stats.copy_(out[0, :3]); pinned.copy_(stats, non_blocking=True)
            e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
            if mode != "none" or (i & 63) == 63:
                pass
            worst = max(worst, time.perf_counter() - t1)
        e.synchronize()
        dt = time.perf_counter() - t0
    rows.append(dict(mode=mode, minibatches=N, us_per_minibatch=round(dt / N * 1e6, 2), worst_host_us=round(worst * 1e6, 1),
                     giveups=e.profile_query("persist_giveups")[1], replayed=e.profile_query("persist_replayed")[1],
                     dropped=e.profile_query("persist_dropped")[1], persistent_launches=e.profile_query("persist_launches")[1]))
    print(json.dumps(rows[-1]), flush=True)
    e.close()
