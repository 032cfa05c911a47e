"""klstm_gemm16.hip -- the pipelined bf16 product of the many-stream chains (x-projection, P, d_r, in_diff of BASELINE.json
configs[4]) through klstm_debug_gemm_bf16_nt2, against the same product in float64 on operands rounded to bf16 the way the kernel
rounds them (RNE): what is left is fp32 accumulation order, bounded at 2e-6 of the result's maximum.  Every tile width, every K split
(the in-launch reduction by the last-arriving slice), ragged M / N / K tails, two products in one launch, pitched operands, and
bit-identical results from launch to launch (slabs are added in slice order whoever arrives last)."""
import numpy as np
import pytest
import torch

from tests.margins import bound

pytestmark = pytest.mark.gpu


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _ref(A, B, bias, add):
    r = _bf16(A) @ _bf16(B).t()
    if bias is not None:
        r = r + bias.double()
    if add is not None:
        r = add.double() + r
    return r


def _rel(c, r):
    return float((c.double() - r).abs().max() / r.abs().max())


@pytest.mark.parametrize("M,N,K,nj,ks", [
    (640, 4096, 512, 0, 0), (640, 1024, 512, 0, 0), (640, 512, 4096, 0, 0),          # configs[4]: x-projection, P, d_r as planned
    (640, 512, 4096, 1, 8), (640, 512, 4096, 2, 8), (640, 512, 4096, 4, 8), (640, 512, 4096, 2, 4), (640, 512, 4096, 2, 2), (640, 512, 4096, 2, 1),
    (300, 72, 192, 1, 1), (300, 72, 256, 2, 2), (257, 100, 320, 4, 1), (520, 40, 3200, 1, 4), (260, 96, 1024, 2, 4),   # ragged M / N tails, uneven slices
    (256, 32, 64, 1, 1), (256, 32, 256, 1, 2), (256, 64, 320, 2, 1),                                          # the smallest the kernel takes
])
def test_single_product(M, N, K, nj, ks):
    import kaldi_lstm_amd as k
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda(); B = (0.05 * torch.randn(N, K, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda(); add = torch.randn(M, N, generator=g).cuda()
    for use_bias, use_add in ((False, False), (True, False), (False, True)):
        C = torch.full((M, N), float("nan"), device="cuda")
        plan = k.debug_gemm_bf16_nt2([(A, B, C, bias if use_bias else None, add if use_add else None)], nj, ks)
        torch.cuda.synchronize()
        assert (nj == 0 or plan[0] == nj) and (ks == 0 or plan[1] == ks), plan
        bound(_rel(C, _ref(A, B, bias if use_bias else None, add if use_add else None)), 2e-6, "C")


def test_two_products_in_one_launch_pitched_and_deterministic():
    """d_r and in_diff of a configs[4] layer: the same dgifo plane one time block apart, W_gifo_r^T and W_gifo_x^T, out_diff added to the
    first; pitched output and `add` views; 20 launches give the same bits (and leave the ticket words at zero: the 20th still works)."""
    import kaldi_lstm_amd as k
    S, T, C4, R, I = 32, 20, 4096, 512, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    dg = (0.1 * torch.randn((T + 2) * S, C4, generator=g)).cuda()
    wrT = (0.05 * torch.randn(R, C4, generator=g)).cuda(); wxT = (0.05 * torch.randn(I, C4, generator=g)).cuda()
    od = torch.randn(T * S, R + 8, generator=g).cuda()[:, :R]
    dr = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I + 12, device="cuda")[:, :I]
    first = None
    for it in range(20):
        dr.fill_(float("nan")); ind.fill_(float("nan"))
        plan = k.debug_gemm_bf16_nt2([(dg[2 * S:(T + 2) * S], wrT, dr, None, od), (dg[S:(T + 1) * S], wxT, ind, None, None)])
        torch.cuda.synchronize()
        if first is None:
            assert plan[1] > 1, plan                                            # K = 4096 is split
            bound(_rel(dr, _ref(dg[2 * S:(T + 2) * S], wrT, None, od)), 2e-6, "d_r")
            bound(_rel(ind, _ref(dg[S:(T + 1) * S], wxT, None, None)), 2e-6, "in_diff")
            first = (dr.clone(), ind.clone())
        else:
            assert torch.equal(dr, first[0]) and torch.equal(ind, first[1]), it


def test_engine_takes_the_pipelined_products_and_matches_the_old_kernels():
    """A configs[4] layer (512 -> 1024 / 512, 32 streams, bf16 operand mode) with option "gemm_nt2" = 1 (default) and 0 (round 4's
    kernel + reduction launches): the same operands rounded at the same points, different fp32 accumulation order -- out, in_diff and
    the gradients agree to 2e-5 -- and the profile says which kernels ran (no reduction launches on the pipelined path)."""
    import kaldi_lstm_amd as k
    from oracle.oracle import make_params
    I, C, R, S, T = 512, 1024, 512, 32, 20
    p = make_params(I, C, R, scale=0.02, seed=3)
    rng = np.random.RandomState(4)
    x = torch.from_numpy(rng.randn(T * S, I).astype(np.float32)).cuda()
    od = torch.from_numpy((0.1 * rng.randn(T * S, R)).astype(np.float32)).cuda()
    res = []
    for nt2 in (1, 0):
        e = k.Engine(I, C, R, S); e.set_params(p); e.set_option("bf16", 1); e.set_option("gemm_nt2", nt2)
        out = torch.empty(T * S, R, device="cuda"); ind = torch.empty(T * S, I, device="cuda")
        e.set_option("profile", 1)
        e.propagate(x, out); e.backpropagate(x, od, ind, momentum=0.9); e.synchronize()
        nred = e.profile_query("k_reduce_dr")[1] + e.profile_query("k_reduce_indiff")[1]
        assert nred == (0 if nt2 else 2), (nt2, nred)
        assert e.profile_query("k_gemm_dr")[1] == 1 and e.profile_query("k_gemm_indiff")[1] == (0 if nt2 else 1)
        res.append((out.cpu().numpy(), ind.cpu().numpy(), e.get_corr()))
        e.close()
    for name, a, b in zip(("out", "in_diff", "corr"), res[0], res[1]):
        bound(float(np.abs(a - b).max() / np.abs(b).max()), 2e-5, name)


@pytest.mark.parametrize("M,N,K,nj,ks", [
    (640, 512, 4096, 0, 0), (640, 512, 4096, 4, 8), (640, 512, 4096, 2, 4), (640, 512, 4096, 1, 2), (640, 4096, 512, 0, 0), (640, 1024, 512, 0, 0),
    (300, 72, 192, 1, 1), (257, 100, 320, 4, 1), (520, 40, 3200, 1, 4), (260, 96, 1024, 2, 4), (256, 32, 64, 1, 1), (256, 32, 128, 2, 1), (256, 64, 192, 4, 1),
])
def test_bf16_copies_in_memory_give_the_same_bits(M, N, K, nj, ks):
    """klstm_debug_gemm_bf16_nt2h: the operands' bf16 copies (what the BPTT chain and the Update write next to their fp32 results) read
    by LDS-DMA -- same stages, same slices, same MFMA order as the form that rounds fp32 operands while staging them: the SAME BITS,
    also with fewer stages than stage buffers (K = 64, 128), ragged tails and uneven slices."""
    import kaldi_lstm_amd as k
    g = torch.Generator(device="cpu").manual_seed(M + N + K + 1)
    A = torch.randn(M, K, generator=g).cuda(); B = (0.05 * torch.randn(N, K, generator=g)).cuda()
    add = torch.randn(M, N, generator=g).cuda()
    Ah, Bh = A.to(torch.bfloat16), B.to(torch.bfloat16)
    C0 = torch.full((M, N), float("nan"), device="cuda"); C1 = torch.full((M, N), float("nan"), device="cuda")
    p0 = k.debug_gemm_bf16_nt2([(A, B, C0, None, add)], nj, ks)
    p1 = k.debug_gemm_bf16_nt2([(A, B, C1, None, add)], nj, ks, copies=[(Ah, Bh)])
    torch.cuda.synchronize()
    assert p0 == p1
    assert torch.equal(C0, C1)
    bound(_rel(C1, _ref(A, B, None, add)), 2e-6, "C (bf16 copies)")
    # the kernel reads the COPIES: with a copy that is not the rounding of A the result follows the copy
    C2 = torch.empty(M, N, device="cuda")
    k.debug_gemm_bf16_nt2([(A, B, C2, None, None)], nj, ks, copies=[((2 * A).to(torch.bfloat16), Bh)])
    torch.cuda.synchronize()
    bound(_rel(C2, 2 * _ref(A, B, None, None)), 2e-6, "C (copy of 2A)")


def test_two_products_from_bf16_copies_pitched():
    """d_r + in_diff of a configs[4] layer in one launch from the bf16 copies: the dgifo plane one time block apart (the SAME copy
    buffer at two row offsets), pitched weight copies; bits equal to the fp32-operand launch, 10 launches the same bits."""
    import kaldi_lstm_amd as k
    S, T, C4, R, I = 32, 20, 4096, 512, 512
    g = torch.Generator(device="cpu").manual_seed(6)
    dg = (0.1 * torch.randn((T + 2) * S, C4, generator=g)).cuda()
    wrT = (0.05 * torch.randn(R, C4 + 8, generator=g)).cuda()[:, :C4]; wxT = (0.05 * torch.randn(I, C4 + 8, generator=g)).cuda()[:, :C4]
    dgh = dg.to(torch.bfloat16)
    wrTh = torch.empty(R, C4 + 8, device="cuda", dtype=torch.bfloat16)[:, :C4]; wrTh.copy_(wrT)
    wxTh = torch.empty(I, C4 + 8, device="cuda", dtype=torch.bfloat16)[:, :C4]; wxTh.copy_(wxT)
    od = torch.randn(T * S, R, generator=g).cuda()
    out = []
    for copies in (None, [(dgh[2 * S:(T + 2) * S], wrTh), (dgh[S:(T + 1) * S], wxTh)]):
        dr = torch.full((T * S, R), float("nan"), device="cuda"); ind = torch.full((T * S, I), float("nan"), device="cuda")
        jobs = [(dg[2 * S:(T + 2) * S], wrT, dr, None, od), (dg[S:(T + 1) * S], wxT, ind, None, None)]
        k.debug_gemm_bf16_nt2(jobs, 4, 4, copies=copies)          # (the same tile width and split for both forms: the planner's differ)
        torch.cuda.synchronize()
        out.append((dr.clone(), ind.clone()))
        if copies is not None:
            for _ in range(10):
                k.debug_gemm_bf16_nt2(jobs, 4, 4, copies=copies)
            torch.cuda.synchronize()
            assert torch.equal(dr, out[-1][0]) and torch.equal(ind, out[-1][1])
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    # the planner's own choice for the copies (128 x 64 tiles, two slices): another summation order of the same products
    dr = torch.full((T * S, R), float("nan"), device="cuda"); ind = torch.full((T * S, I), float("nan"), device="cuda")
    jobs = [(dg[2 * S:(T + 2) * S], wrT, dr, None, od), (dg[S:(T + 1) * S], wxT, ind, None, None)]
    plan = k.debug_gemm_bf16_nt2(jobs, copies=[(dgh[2 * S:(T + 2) * S], wrTh), (dgh[S:(T + 1) * S], wxTh)])
    torch.cuda.synchronize()
    assert plan[:2] == (2, 2), plan
    bound(_rel(dr, out[0][0].double()), 2e-6, "d_r (copies, own plan)"); bound(_rel(ind, out[0][1].double()), 2e-6, "in_diff (copies, own plan)")
