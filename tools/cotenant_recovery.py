"""The one scenario in which the persistent chain has been seen to lose co-residency on this box (round 3, profiles/r03_cotenant.txt): a
foreign kernel holds 40 compute units AND its residency flags are polled with PAGEABLE device-to-host copies; sequences that contain a
runtime copy then started a launch with ~30 of 200 workgroups missing.  Round 3 reported the give-up as an error; this probe runs the
same sequences and prints what the recovery of round 4 makes of them: give-ups, minibatches run again, dropped -- and the parameters
against a twin engine that never used the persistent chain (launch-per-step, no foreign kernel).  Usage: cotenant_recovery.py [pinned|pageable]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
I, C, R, S, T = 40, 800, 512, 4, 20
lib = k.load_library()
mode = sys.argv[1] if len(sys.argv) > 1 else "pageable"


def hog(n):
    if mode == "pinned":
        where = torch.full((2 * n,), -1, dtype=torch.int32).pin_memory()
        torch.cuda.synchronize()
        lib.klstm_debug_occupy(0, n, 30000, None, where.data_ptr())
        while (where.numpy() == -1).any():
            time.sleep(0.0005)
    else:                                   # the flags live on the device and are polled with pageable D2H copies (tensor.cpu())
        where = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        lib.klstm_debug_occupy(0, n, 30000, None, where.data_ptr())
        while (where.cpu().numpy() == -1).any():
            time.sleep(0.0005)


small = torch.zeros(64, device="cuda"); small2 = torch.zeros(64, device="cuda")
p0 = ((np.random.RandomState(7).rand(k.Engine(I, C, R, S).num_params) - 0.5) * 0.02).astype(np.float32)
for held in (40, 0):
    for seq in ("PBUPBUPBU", "PBUcPBUcPBU", "PBUdPBUdPBU", "PBU|PBU|PBU"):
        e = k.Engine(I, C, R, S); e.set_params(p0)
        e.set_option("persist", 2); e.set_option("persist_spin_us", 3000)
        t = k.Engine(I, C, R, S); t.set_params(p0); t.set_option("persist", 0)
        x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
        out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
        e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5); e.synchronize()
        t.propagate(x, out); t.backpropagate(x, od, ind, 0.9, 2); t.update(1e-5); t.synchronize()
        torch.cuda.synchronize()
        if held:
            hog(held)
        t0 = time.perf_counter()
        for ch in seq:
            if ch == "P": e.propagate(x, out)
            elif ch == "B": e.backpropagate(x, od, ind, 0.9, 2)
            elif ch == "U": e.update(1e-5)
            elif ch == "|": e.synchronize()
            elif ch == "c": time.sleep(0.002); y = small.cpu()
            elif ch == "d": time.sleep(0.002); small2.copy_(small)
        e.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        time.sleep(0.04)                    # (the foreign kernel has ended)
        for ch in seq:                      # the twin: the same minibatches on the launch-per-step chain, on an idle chip
            if ch == "P": t.propagate(x, out)
            elif ch == "B": t.backpropagate(x, od, ind, 0.9, 2)
            elif ch == "U": t.update(1e-5)
        t.synchronize()
        g, r, d = (e.profile_query(n)[1] for n in ("persist_giveups", "persist_replayed", "persist_dropped"))
        pe, pt = e.get_params(), t.get_params()
        err = float(np.abs(pe - pt).max() / np.abs(pt).max())
        print("flags %-8s held %2d  %-12s %6.1f ms  give-ups %d  run again %d  dropped %d  parameters vs launch-per-step twin: %.1e%s"
              % (mode, held, seq, ms, g, r, d, err, "  (a dropped minibatch = one Update less)" if d else ""), flush=True)
        e.close(); t.close()
