"""tools/gemm16_fuzz.py -- random shapes through klstm_gemm16.hip: the LDS-DMA form (bf16 copies of both operands in memory) against the
form that rounds fp32 operands while staging them -- same tile width and K split forced on both, so every output must be BIT-IDENTICAL --
and against the product in float64 on the rounded operands (2e-6 of the result's maximum).  Random M, N, K (multiples of 64), tile
widths, splits, row pitches, one or two products per launch, `add` / bias on or off; every launch repeated: an LDS-DMA stage read before
it has landed, or restaged before it has been read, would show as a wrong tile that comes and goes.
Usage: gemm16_fuzz.py [launch configurations = 300] [seed = 1]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import kaldi_lstm_amd as k

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def pitched(rows, cols, pad, dtype=torch.float32, scale=1.0):
    t = torch.empty(rows, cols + pad, device="cuda", dtype=torch.float32).normal_() * scale
    return t[:, :cols]


def job(M, N, K, use_add, use_bias):
    A = pitched(M, K, 8 * rng.randint(0, 3)); B = pitched(N, K, 8 * rng.randint(0, 3), scale=0.05)
    Ah = torch.empty(M, A.stride(0), device="cuda", dtype=torch.bfloat16)[:, :K]; Ah.copy_(A)
    Bh = torch.empty(N, B.stride(0), device="cuda", dtype=torch.bfloat16)[:, :K]; Bh.copy_(B)
    add = pitched(M, N, 4 * rng.randint(0, 3)) if use_add else None
    bias = torch.randn(N, device="cuda") if use_bias else None
    return A, B, Ah, Bh, add, bias


done = skipped = 0
worst = 0.0
while done < n_cfg:
    njobs = 1 + (rng.rand() < 0.3)
    K = 64 * rng.randint(1, 70)
    nj = int(rng.choice([1, 2, 4])); ks = int(rng.choice([1, 1, 2, 4, 8]))
    specs = [(int(rng.randint(256, 900)), int(rng.randint(32, 1100))) for _ in range(njobs)]
    js = [job(M, N, K, rng.rand() < 0.5, rng.rand() < 0.3 and njobs == 1) for M, N in specs]
    outs = []
    try:
        for copies in (None, [(j[2], j[3]) for j in js]):
            Cs = [torch.full((j[0].shape[0], j[1].shape[0]), float("nan"), device="cuda") for j in js]
            jobs = [(j[0], j[1], C, j[5], j[4]) for j, C in zip(js, Cs)]
            plan = k.debug_gemm_bf16_nt2(jobs, nj, ks, copies=copies)
            torch.cuda.synchronize()
            first = [C.clone() for C in Cs]
            for _ in range(3):
                k.debug_gemm_bf16_nt2(jobs, nj, ks, copies=copies)
            torch.cuda.synchronize()
            for a, b in zip(first, Cs):
                assert torch.equal(a, b), ("not reproducible", specs, K, plan, copies is not None)
            outs.append(first)
    except k.KlstmError as ex:                        # (a plan the launcher refuses: an empty K slice, too many tiles)
        skipped += 1
        continue
    for j, c0, c1 in zip(js, outs[0], outs[1]):
        assert torch.equal(c0, c1), ("copies form differs", specs, K, plan, float((c0 - c1).abs().max()))
        ref = j[0].to(torch.bfloat16).double() @ j[1].to(torch.bfloat16).double().t()
        if j[5] is not None: ref = ref + j[5].double()
        if j[4] is not None: ref = j[4].double() + ref
        err = float((c1.double() - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err <= 2e-6, ("accuracy", specs, K, plan, err)
    done += 1
print("%d launch configurations (1 or 2 products, K = 64 .. 4416, tiles 128 x 32 / 64 / 128, 1 .. 8 K slices, pitched operands; %d plans refused "
      "by the launcher): the LDS-DMA form bit-identical to the fp32-operand form, every launch reproducible four times, largest error against "
      "float64 on the rounded operands %.2e" % (done, skipped, worst))
