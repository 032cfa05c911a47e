import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import numpy as np, torch, torch.distributed as dist
import kaldi_lstm_amd as k
from oracle.oracle import Oracle, make_params
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
I, C, R, S, T = 40, 800, 512, 8, 20
p = make_params(I, C, R, 0.01, 3)
e = k.Engine(I, C, R, S); e.set_params(p)
dp = k.DataParallelLstm(e, force_collective=True)
print("native comm:", dp.comm is not None)
rng = np.random.RandomState(1)
o = Oracle(I, C, R, S, np.float32); o.set_params(p)
out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
for it in range(3):
    x = rng.randn(T * S, I).astype(np.float32); od = (0.1 * rng.randn(T * S, R)).astype(np.float32)
    xd, odd = torch.from_numpy(x).cuda(), torch.from_numpy(od).cuda()
    dp.train_step(xd, out, odd, ind, 0.9, 1e-5, reset_flags=[1] * S if it == 0 else None)
    if it == 0: o.reset([1] * S)
    yo = o.propagate(x); ido = o.backpropagate(x, od, momentum=0.9); o.update(1e-5)
    e.synchronize()
    err = lambda a, b: float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
    print(it, "out", err(out.cpu().numpy(), yo), "in_diff", err(ind.cpu().numpy(), ido), "corr", err(e.get_corr(), o.get_corr()), "params", err(e.get_params(), o.get_params()))
t0 = time.perf_counter()
for _ in range(200): dp.train_step(xd, out, odd, ind, 0.9, 1e-5)
e.synchronize(); print("DP step (1 rank, collective forced): %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
dist.destroy_process_group()
