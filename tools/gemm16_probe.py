"""tools/gemm16_probe.py -- device time of the pipelined bf16 product (klstm_gemm16.hip) per tile width and K split at the shapes of
BASELINE.json configs[4] (640 rows per minibatch and layer), 20 launches per hipGraph replay so that the host is out of the picture.
Round 4's kernel for comparison (profiles/r04_bench_c5.json): x-projection 16.2 us, P 11.0, d_r 20.2 + 4.3, in_diff 20.2 + 4.3."""
import sys
import torch
sys.path.insert(0, ".")
import kaldi_lstm_amd as k


def timed(fn, reps=20, rounds=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(st.cuda_stream); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn(st.cuda_stream)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); g.replay(); b.record(st); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


def main_copies():
    """--copies: the K = 4C products from bf16 copies in memory (LDS-DMA form) next to the fp32-operand form, per tile width and split"""
    M, S, T = 640, 32, 20
    gen = torch.Generator(device="cpu").manual_seed(0)
    dg = torch.randn((T + 2) * S, 4096, generator=gen).cuda(); wrT = torch.randn(512, 4096, generator=gen).cuda(); wxT = torch.randn(512, 4096, generator=gen).cuda()
    od = torch.randn(M, 512, generator=gen).cuda(); dr = torch.empty(M, 512, device="cuda"); ind = torch.empty(M, 512, device="cuda")
    dgh, wrTh, wxTh = dg.to(torch.bfloat16), wrT.to(torch.bfloat16), wxT.to(torch.bfloat16)
    jobs2 = [(dg[2 * S:], wrT, dr, None, od), (dg[S:(T + 1) * S], wxT, ind, None, None)]
    cp2 = [(dgh[2 * S:], wrTh), (dgh[S:(T + 1) * S], wxTh)]
    for label, jobs, cps in (("pair d_r + in_diff", jobs2, cp2), ("d_r alone", jobs2[:1], cp2[:1])):
        for nj in (0, 1, 2, 4):
            for ks in ((0,) if nj == 0 else (1, 2, 4, 8)):
                row = []
                for copies in (None, cps):
                    try:
                        us = timed(lambda s: k.debug_gemm_bf16_nt2(jobs, nj, ks, s, copies=copies))
                    except Exception as ex:
                        us = float("nan")
                    row.append(us)
                plan = k.debug_gemm_bf16_nt2(jobs, nj, ks, copies=cps)
                print("%-19s nj %d ks %d : fp32 operands %6.1f us   bf16 copies %6.1f us" % (label, plan[0], plan[1], row[0], row[1]), flush=True)
    # the short-K products, for completeness (their A operand is the caller's: no copy exists in the engine)
    for name, N, K in (("xproj", 4096, 512), ("P", 1024, 512)):
        A = torch.randn(M, K, generator=gen).cuda(); B = torch.randn(N, K, generator=gen).cuda(); C = torch.empty(M, N, device="cuda")
        for nj in (1, 2, 4):
            r0 = timed(lambda s: k.debug_gemm_bf16_nt2([(A, B, C, None, None)], nj, 1, s))
            Ah, Bh = A.to(torch.bfloat16), B.to(torch.bfloat16)
            r1 = timed(lambda s: k.debug_gemm_bf16_nt2([(A, B, C, None, None)], nj, 1, s, copies=[(Ah, Bh)]))
            print("%-19s nj %d ks 1 : fp32 operands %6.1f us   bf16 copies %6.1f us" % (name, nj, r0, r1), flush=True)


def main():
    if "--copies" in sys.argv:
        return main_copies()
    M = 640
    gen = torch.Generator(device="cpu").manual_seed(0)
    shapes = [("xproj", 4096, 512), ("P", 1024, 512), ("d_r", 512, 4096)]
    import os
    print("KLSTM_G16_NF=%s" % os.environ.get("KLSTM_G16_NF", "2"))
    for name, N, K in shapes:
        if K % 8 or K < 64:
            continue
        A = torch.randn(M, K, generator=gen).cuda(); B = torch.randn(N, K, generator=gen).cuda(); C = torch.empty(M, N, device="cuda")
        bias = torch.randn(N, generator=gen).cuda()
        for nj in (1, 2, 4):
            for ks in (1, 2, 4, 8):
                if K // ks < 128 or (K <= 512 and ks > 2):
                    continue
                try:
                    us = timed(lambda s: k.debug_gemm_bf16_nt2([(A, B, C, bias, None)], nj, ks, s))
                except Exception as ex:
                    print(name, nj, ks, "failed:", ex); continue
                fl = 2.0 * M * N * K
                by = 4.0 * (M * K + N * K + M * N)
                print("%-7s %4d x %4d x %4d  nj %d ks %d : %6.1f us  %6.1f TF/s  %5.2f TB/s" % (name, M, N, K, nj, ks, us, fl / us / 1e6, by / us / 1e6), flush=True)
        plan = k.debug_gemm_bf16_nt2([(A, B, C, bias, None)])
        print("%-7s planned: nj %d ks %d" % (name, plan[0], plan[1]))
    # d_r + in_diff in one launch
    S, T = 32, 20
    dg = torch.randn((T + 2) * S, 4096, generator=gen).cuda(); wrT = torch.randn(512, 4096, generator=gen).cuda(); wxT = torch.randn(512, 4096, generator=gen).cuda()
    od = torch.randn(M, 512, generator=gen).cuda(); dr = torch.empty(M, 512, device="cuda"); ind = torch.empty(M, 512, device="cuda")
    for nj in (1, 2, 4):
        for ks in (2, 4, 8):
            us = timed(lambda s: k.debug_gemm_bf16_nt2([(dg[2 * S:], wrT, dr, None, od), (dg[S:(T + 1) * S], wxT, ind, None, None)], nj, ks, s))
            print("pair d_r + in_diff  nj %d ks %d : %6.1f us  %6.1f TF/s" % (nj, ks, us, 2 * 2.0 * M * 512 * 4096 / us / 1e6), flush=True)
    print("planned pair:", k.debug_gemm_bf16_nt2([(dg[2 * S:], wrT, dr, None, od), (dg[S:(T + 1) * S], wxT, ind, None, None)]))


if __name__ == "__main__":
    main()
