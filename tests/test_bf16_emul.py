"""The bf16 emulation (tests/bf16_emul.py) with rounding switched off must equal the oracle: pins the emulation's
formulas before the GPU test uses it to check the engine's bf16 mode."""
import numpy as np

from oracle.oracle import Oracle, make_params, split_blob, param_sizes
from tests import bf16_emul


def test_emulation_without_rounding_equals_oracle(monkeypatch):
    I, C, R, S, T = 8, 16, 8, 3, 5
    monkeypatch.setattr(bf16_emul, "rb", lambda a: np.asarray(a, np.float64))
    rng = np.random.RandomState(3)
    p = make_params(I, C, R, scale=0.3, seed=4)
    o = Oracle(I, C, R, S, np.float64)
    o.set_params(p)
    parts = [split_blob(p, I, C, R)[n] for n, _ in param_sizes(I, C, R)]
    c0, r0 = np.zeros((S, C)), np.zeros((S, R))
    for ck in range(2):                                   # second chunk: carried state
        x = rng.randn(T * S, I)
        od = rng.randn(T * S, R)
        out_o = o.propagate(x)
        id_o = o.backpropagate(x, od, momentum=0.0)
        for fuse_x in (0, 1):
            out, idf, grads, cT, rT = bf16_emul.minibatch(parts, x, od, c0, r0, S, fuse_x)
            assert np.abs(out - out_o).max() < 1e-6       # f32() hooks round planes to fp32
            assert np.abs(idf - id_o).max() < 1e-5
            g = np.concatenate([a.ravel() for a in grads])
            assert np.abs(g - o.get_corr()).max() < 1e-5 * max(1.0, np.abs(g).max())
        st = o.get_state()
        c0, r0 = st[:, 4 * C:5 * C].copy(), st[:, 7 * C:].copy()
