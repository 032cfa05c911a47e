"""Numpy emulation of the engine's bf16 operand mode (option "bf16"), test infrastructure only.

Semantics being pinned: in the four per-step contractions (gates :275 [+ :246 when the x term is fused, or batched over >= 256 frames of >= 128 inputs], projection
:312, d_r/in_diff :391/:457, d_m :408) BOTH operands are rounded to bf16 (round-to-nearest-even) and the products
are accumulated in fp32; so are the three gradient products (:468, :471, :486) from 256 frames per minibatch on.  Everything else -- elementwise math,
activation planes, the bias / peephole sums, momentum, Update, the fp32 master weights -- is unchanged fp32.  Products of bf16 values are exact in
fp64, so this emulation accumulates in fp64 and differs from the GPU only by fp32 summation order and by the rare
1-ulp bf16 flips that order causes (tolerances in tests/test_engine_gpu.py).
"""
import numpy as np
import torch


def rb(a):
    """fp32 -> bf16 (RNE) -> fp64"""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(torch.bfloat16).to(torch.float32).numpy().astype(np.float64)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


GRADS_BF16_MIN_ROWS = 256      # klstm_kernels.hip: the gradient products run on the bf16 pipe from this many frames on


def minibatch(parts, x, od, c0, r0, S, fuse_x, want_in_diff=True, fold=False, fold_bwd=False):
    """parts = [wx, wr, b, pi, pf, po, wm] fp32 arrays; x [T*S, I], od [T*S, R] time-major; c0 [S, C], r0 [S, R].
    Returns out, in_diff, grads (7 arrays, pure gradient), cT, rT.
    fold: the many-stream weights-resident forward launch (klstm_persist_ms.hip): steps 2..T close over m(t-1) through
    W_rm = W_gifo_r W_r_m -- itself a bf16 product with fp32 accumulation, its fp32 result rounded to bf16 when the launch loads it;
    step 1 closes over the carried r as before; r(t) is still bf16(m(t)) x bf16(W_r_m) (output rows, BPTT operands).
    fold_bwd: the BPTT chain of klstm_persist_xl.hip closes over dgifo the same way: d_m(t) = P(t) + bf16(dgifo(t+1)) x bf16(W_rm) with
    P = bf16(out_diff) x bf16(W_r_m) (fp32 accumulate, kept in fp32); d_r (the W_r_m gradient's operand) and in_diff are unchanged."""
    wx, wr, b, pi, pf, po, wm = [np.asarray(p, np.float64) for p in parts]
    C, R, I = pi.shape[0], wm.shape[0], wx.shape[1]
    T = x.shape[0] // S
    wrb, wmb, wxb = rb(wr), rb(wm), rb(wx)
    wrmb = rb((wrb @ wmb).astype(np.float32)) if (fold or fold_bwd) else None       # [4C x C]
    x = np.asarray(x, np.float64)
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    g = np.zeros((T + 2, S, C)); i = np.zeros_like(g); f = np.zeros_like(g); o = np.zeros_like(g)
    c = np.zeros_like(g); h = np.zeros_like(g); m = np.zeros_like(g); r = np.zeros((T + 2, S, R))
    c[0], r[0] = c0, r0
    x_bf16 = fuse_x or (T * S >= GRADS_BF16_MIN_ROWS and I >= 128)   # the batched x-projection runs on the bf16 pipe from 256 frames and 128 inputs on
    for t in range(1, T + 1):
        xt = x[(t - 1) * S:t * S]
        rec = rb(m[t - 1]) @ wrmb.T if (fold and t >= 2) else rb(r[t - 1]) @ wrb.T
        a = (rb(xt) @ wxb.T if x_bf16 else f32(xt @ wx.T)) + b + rec
        ag, ai, af, ao = a[:, :C], a[:, C:2 * C], a[:, 2 * C:3 * C], a[:, 3 * C:]
        i[t] = sigmoid(ai + c[t - 1] * pi)
        f[t] = sigmoid(af + c[t - 1] * pf)
        g[t] = np.tanh(ag)
        c[t] = np.clip(g[t] * i[t] + c[t - 1] * f[t], -50, 50)
        h[t] = np.tanh(c[t])
        o[t] = sigmoid(ao + c[t] * po)
        m[t] = f32(h[t] * o[t])
        r[t] = f32(rb(m[t]) @ wmb.T)
    out = r[1:T + 1].reshape(T * S, R)
    dg = np.zeros((T + 2, S, 4 * C)); dc = np.zeros((T + 2, S, C)); dr = np.zeros((T + 2, S, R))
    od = np.asarray(od, np.float64).reshape(T, S, R)
    Pf = f32(rb(od.reshape(T * S, R)) @ wmb).reshape(T, S, C) if fold_bwd else None
    for t in range(T, 0, -1):
        dr[t] = f32(od[t - 1] + rb(dg[t + 1]) @ wrb)
        dm = Pf[t - 1] + rb(dg[t + 1]) @ wrmb if fold_bwd else rb(dr[t]) @ wmb
        dh = dm * o[t] * (1 - h[t] ** 2)
        do = dm * h[t] * o[t] * (1 - o[t])
        dgn = dg[t + 1]
        dct = dh + dc[t + 1] * f[t + 1] + dgn[:, C:2 * C] * pi + dgn[:, 2 * C:3 * C] * pf + do * po
        df = dct * c[t - 1] * f[t] * (1 - f[t])
        di = dct * g[t] * i[t] * (1 - i[t])
        dgg = dct * i[t] * (1 - g[t] ** 2)
        dg[t] = f32(np.concatenate([dgg, di, df, do], 1))
        dc[t] = dct
    D = dg[1:T + 1].reshape(T * S, 4 * C)
    in_diff = rb(D) @ wxb if want_in_diff else None
    Rm1 = r[0:T].reshape(T * S, R)
    Cm1 = c[0:T].reshape(T * S, C)
    C1 = c[1:T + 1].reshape(T * S, C)
    # the three gradient PRODUCTS (:468, :471, :486) round both operands to bf16; the bias / peephole sums stay fp32
    rg = rb if T * S >= GRADS_BF16_MIN_ROWS else (lambda a: np.asarray(a, np.float64))
    Db = rg(D)
    grads = [Db.T @ rg(x), Db.T @ rg(Rm1), D.sum(0), (D[:, C:2 * C] * Cm1).sum(0), (D[:, 2 * C:3 * C] * Cm1).sum(0),
             (D[:, 3 * C:] * C1).sum(0), rg(dr[1:T + 1].reshape(T * S, R)).T @ rg(m[1:T + 1].reshape(T * S, C))]
    return out, in_diff, grads, c[T], r[T]
