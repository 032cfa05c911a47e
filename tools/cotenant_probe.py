"""What, next to a foreign kernel that holds 40 compute units, makes a persistent launch lose co-residency?  Sequences of
P(ropagate) B(ackpropagate) U(pdate) with, in between: | klstm_synchronize (stream sync + a 64-byte hipMemcpy D2H of the
status words), t a 2 ms host gap, s a small kernel on the null stream, c a null-stream D2H copy, d a D2D copy, m a device
allocation.  Result on MI355X / ROCm 7.2 (profiles/r03_cotenant.txt): with the foreign kernel's residency flags in PINNED host
memory every sequence passes.  An earlier version polled the flags with pageable D2H copies (tensor.cpu() in a loop): then the
sequences containing a runtime copy (|, c, d) gave up -- the launch after the copy started with ~30 of its 200 workgroups
missing until the foreign kernel ended -- while t, s, m and back-to-back launches passed.  Not understood."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
I, C, R, S, T = 40, 800, 512, 4, 20
lib = k.load_library()
def hog(n):
    where = torch.full((2 * n,), -1, dtype=torch.int32).pin_memory()
    torch.cuda.synchronize()
    lib.klstm_debug_occupy(0, n, 30000, None, where.data_ptr())
    while (where.numpy() == -1).any(): time.sleep(0.0005)
small = torch.zeros(64, device="cuda"); small2 = torch.zeros(64, device="cuda")
for held in (40, 0):
    for seq in ("PBUPBUPBU", "PtPtP", "PsPsP", "PmPmP", "PcPcP", "PdPdP", "P|P|P", "PBU|PBU|PBU"):
        e = k.Engine(I, C, R, S)
        e.set_params(((np.random.RandomState(7).rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
        e.set_option("persist", 2); e.set_option("persist_spin_us", 3000)
        x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
        out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
        e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5); e.synchronize()
        torch.cuda.synchronize()
        if held: hog(held)
        trace = ""
        try:
            for ch in seq:
                if ch == "P": e.propagate(x, out)
                elif ch == "B": e.backpropagate(x, od, ind, 0.9, 2)
                elif ch == "U": e.update(1e-5)
                elif ch == "|": e.synchronize()
                elif ch == "t": time.sleep(0.002)
                elif ch == "s": time.sleep(0.002); small.add_(1)
                elif ch == "c": time.sleep(0.002); y = small.cpu()
                elif ch == "d": time.sleep(0.002); small2.copy_(small)
                elif ch == "m": time.sleep(0.002); y = torch.empty(1 << 20, device="cuda")
                trace += ch
            e.synchronize(); print("held %2d  %-12s ok" % (held, seq), flush=True)
        except k.KlstmError as ex:
            print("held %2d  %-12s gave up after '%s'" % (held, seq, trace), flush=True)
        torch.cuda.synchronize()
        e.close()
