"""tools/gemm16_anatomy.py -- where a workgroup of the pipelined bf16 product (klstm_gemm16.hip) spends its shader clocks: the TIMING
instantiation records, per workgroup, the MFMA wave's time at the stage barriers (= the stage had not landed), in its LDS reads, its
whole K loop; the loader wave's time waiting for its own requests and its whole loop; the epilogue."""
import ctypes
import sys
import torch
sys.path.insert(0, ".")
import kaldi_lstm_amd as k


def run(name, jobs, nj, ks, copies=None):
    lib = k.load_library()
    dbg = torch.zeros(16 * 8192, dtype=torch.int64, device="cuda")
    lib.klstm_debug_gemm_bf16_nt2(0, ctypes.c_void_p(dbg.data_ptr()), None, None, 0, 0, None, None)
    for _ in range(3):
        k.debug_gemm_bf16_nt2(jobs, nj, ks, copies=copies)
    torch.cuda.synchronize()
    dbg.zero_(); torch.cuda.synchronize()
    plan = k.debug_gemm_bf16_nt2(jobs, nj, ks, copies=copies)
    torch.cuda.synchronize()
    lib.klstm_debug_gemm_bf16_nt2(0, None, None, None, 0, 0, None, None)
    d = dbg.view(-1, 16).cpu().double()
    d = d[d[:, 3] > 0]
    m = d.mean(0)
    print("%-14s nj %d ks %d: %4d workgroups x %3d stages | MFMA wave: K loop %7.0f clk = barrier %6.0f + LDS reads %6.0f + rest %6.0f | per stage %5.0f | "
          "loader: loop %7.0f = wait + convert %6.0f (wait %5.0f) + issue %6.0f + barrier %6.0f | epilogue %6.0f (last arrival %6.0f)" %
          (name, plan[0], plan[1], d.shape[0], int(m[3]), m[2], m[0], m[1], m[2] - m[0] - m[1], m[2] / max(m[3], 1), m[5], m[8], m[4], m[9], m[10], m[6],
           d[:, 7].max()), flush=True)


def main():
    M = 640
    gen = torch.Generator(device="cpu").manual_seed(0)
    for name, N, K, cfgs in (("xproj", 4096, 512, ((2, 1), (1, 1), (4, 1))), ("P", 1024, 512, ((1, 1), (2, 1))), ("d_r", 512, 4096, ((2, 4), (2, 8), (1, 4)))):
        A = torch.randn(M, K, generator=gen).cuda(); B = torch.randn(N, K, generator=gen).cuda(); C = torch.empty(M, N, device="cuda")
        for nj, ks in cfgs:
            run(name, [(A, B, C, None, None)], nj, ks)
    # d_r + in_diff of a configs[4] layer in one launch
    S, T = 32, 20
    dg = torch.randn((T + 2) * S, 4096, generator=gen).cuda(); wrT = torch.randn(512, 4096, generator=gen).cuda(); wxT = torch.randn(512, 4096, generator=gen).cuda()
    od = torch.randn(M, 512, generator=gen).cuda(); dr = torch.empty(M, 512, device="cuda"); ind = torch.empty(M, 512, device="cuda")
    for nj, ks in ((4, 4), (4, 2), (2, 4), (4, 8)):
        run("d_r + in_diff", [(dg[2 * S:], wrT, dr, None, od), (dg[S:(T + 1) * S], wxT, ind, None, None)], nj, ks)
    # ... from the bf16 copies in memory (LDS-DMA form; loader columns: wait = own requests, issue = the DMA instructions)
    dgh, wrTh, wxTh = dg.to(torch.bfloat16), wrT.to(torch.bfloat16), wxT.to(torch.bfloat16)
    for nj, ks in ((4, 4), (2, 2), (1, 1), (2, 4), (4, 2)):
        run("pair, copies", [(dg[2 * S:], wrT, dr, None, od), (dg[S:(T + 1) * S], wxT, ind, None, None)], nj, ks,
            copies=[(dgh[2 * S:], wrTh), (dgh[S:(T + 1) * S], wxTh)])


if __name__ == "__main__":
    main()
