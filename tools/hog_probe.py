"""How many compute units may a foreign kernel hold before the 200-workgroup persistent launches (40/800/512) stop being
co-resident?  The foreign kernel (klstm_debug_occupy: one 1024-thread, 96 KB workgroup per CU) is confirmed resident before
the minibatches start, and reports where its workgroups landed.  Result on MI355X (profiles/r03_hog_probe.txt): up to 32 held
CUs (4 per XCD = one per shader engine) the chain runs undisturbed; from 40 on (5 per XCD: two in one shader engine) the
persistent launch waits for the foreign kernel to end -- workgroups are handed to shader engines round-robin, 25 per XCD =
7 + 6 + 6 + 6 over 4 engines of 8 CUs, and an engine with 2 CUs taken has 6 left for 7."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kaldi_lstm_amd as k
I, C, R, S, T = 40, 800, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 20
lib = k.load_library()
print("CUs:", torch.cuda.get_device_properties(0).multi_processor_count)
for hog in (0, 8, 16, 24, 32, 40, 48, 56):
    e = k.Engine(I, C, R, S)
    e.set_params(((np.random.RandomState(7).rand(e.num_params) - 0.5) * 0.02).astype(np.float32))
    e.set_option("persist", 2); e.set_option("persist_spin_us", 100000)
    x = torch.randn(T * S, I, device="cuda"); od = 0.1 * torch.randn(T * S, R, device="cuda")
    out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
    e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5); e.synchronize()
    torch.cuda.synchronize()
    where = torch.full((2 * max(hog, 1),), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    per_se = ""
    if hog:
        lib.klstm_debug_occupy(0, hog, 30000, None, where.data_ptr())
        while (where.cpu() == -1).any(): time.sleep(0.0005)          # every workgroup of the foreign kernel is resident
        w = where.cpu().numpy().reshape(hog, 2)
        c = collections.Counter((int(a) & 0xf, (int(b) >> 13) & 7) for a, b in w)    # (xcc, shader engine)
        per_se = "max per (XCD, SE): %d" % max(c.values())
    t0 = time.perf_counter()
    try:
        for _ in range(3):
            e.propagate(x, out); e.backpropagate(x, od, ind, 0.9, 2); e.update(1e-5)
        e.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        print("held CUs %2d: 3 minibatches in %6.2f ms  %s  %s" % (hog, ms, "(waited for the foreign kernel)" if ms > 5 else "undisturbed", per_se), flush=True)
    except k.KlstmError as ex:
        print("held CUs %2d: gave up: %s" % (hog, str(ex)[60:130]), flush=True)
    torch.cuda.synchronize()
    e.close()
