import torch, time
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
a = torch.randn(3200, 512, device="cuda"); b = torch.randn(512, 800, device="cuda")
print("torch mm 3200x512 @ 512x800 : %.1f us" % t(lambda: torch.mm(a, b)))
c = torch.mm(a, b)
print("torch transpose copy 3200x800: %.1f us" % t(lambda: c.t().contiguous()))
m = torch.randn(80, 800, device="cuda"); wm = torch.randn(512, 800, device="cuda")
print("torch mm 80x800 @ 800x512 (rbatch): %.1f us" % t(lambda: torch.mm(m, wm.t())))
od = torch.randn(80, 512, device="cuda")
print("torch mm 80x512 @ 512x800 (P): %.1f us" % t(lambda: torch.mm(od, wm)))
dg = torch.randn(80, 3200, device="cuda"); wr = torch.randn(3200, 512, device="cuda"); wx = torch.randn(3200, 40, device="cuda")
print("torch mm 80x3200 @ 3200x512 (dr): %.1f us" % t(lambda: torch.mm(dg, wr)))
print("torch mm 80x3200 @ 3200x40 (dx): %.1f us" % t(lambda: torch.mm(dg, wx)))
torch.backends.cuda.matmul.allow_tf32 = False
