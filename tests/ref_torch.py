"""Independent fp64 formulation of the LSTMP forward equations (SURVEY.md Appendix A /
reference ...streams.h:246-312) in torch, differentiated by autograd.  It shares no code
with oracle/lstmp_oracle.c or the HIP engine: it is the third leg that checks the
oracle's hand-written BPTT (reference ...streams.h:367-487).

The reference's cell clip has NO gradient mask (:296-297 vs :424-428): reproduced with a
straight-through clamp.
"""
import torch


def split(flat, I, C, R):
    o = 0
    out = []
    for shp in [(4 * C, I), (4 * C, R), (4 * C,), (C,), (C,), (C,), (R, C)]:
        n = 1
        for d in shp:
            n *= d
        out.append(flat[o:o + n].reshape(shp))
        o += n
    return out


def forward(flat, x, c0, r0, I, C, R, S):
    """x [T*S, I] time-major; c0 [S,C], r0 [S,R].  Returns out [T*S,R], cT, rT."""
    wx, wr, b, pi, pf, po, wm = split(flat, I, C, R)
    T = x.shape[0] // S
    c, r = c0, r0
    outs = []
    for t in range(T):
        xt = x[t * S:(t + 1) * S]
        a = xt @ wx.t() + b + r @ wr.t()
        ag, ai, af, ao = a[:, :C], a[:, C:2 * C], a[:, 2 * C:3 * C], a[:, 3 * C:]
        i = torch.sigmoid(ai + c * pi)
        f = torch.sigmoid(af + c * pf)
        g = torch.tanh(ag)
        cu = g * i + c * f
        c = cu + (cu.clamp(-50, 50) - cu).detach()      # straight-through clip
        h = torch.tanh(c)
        o = torch.sigmoid(ao + c * po)
        m = h * o
        r = m @ wm.t()
        outs.append(r)
    return torch.cat(outs, 0), c, r


def grads(flat_np, x_np, od_np, c0_np, r0_np, I, C, R, S):
    """Truncated-BPTT gradients of sum(out*out_diff) wrt params and x (state inputs are
    constants: no gradient flows into the previous batch)."""
    flat = torch.tensor(flat_np, dtype=torch.float64, requires_grad=True)
    x = torch.tensor(x_np, dtype=torch.float64, requires_grad=True)
    od = torch.tensor(od_np, dtype=torch.float64)
    c0 = torch.tensor(c0_np, dtype=torch.float64)
    r0 = torch.tensor(r0_np, dtype=torch.float64)
    out, cT, rT = forward(flat, x, c0, r0, I, C, R, S)
    (out * od).sum().backward()
    return out.detach().numpy(), flat.grad.numpy(), x.grad.numpy(), cT.detach().numpy(), rT.detach().numpy()
