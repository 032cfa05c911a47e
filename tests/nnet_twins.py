"""Oracle-backed CPU twins of the DataParallelNnet layer protocol (kaldi-lstm_amd/dp.py): test infrastructure.
LSTM layers run oracle/lstmp_oracle.c, the tail runs oracle/components.py (numpy).  Tensors are torch CPU tensors of
`dtype`; gradients are written into the bound slice of the fused blob exactly like the device layers do."""
import numpy as np
import torch

from oracle.oracle import Oracle
from oracle import components as comp


class OracleLstmLayer:
    def __init__(self, I, C, R, S, params, dtype=np.float64):
        self.o = Oracle(I, C, R, S, dtype)
        self.o.set_params(params)
        self.num_params = self.o.num_params
        self.dtype = dtype

    def bind_grad(self, view):
        self.g = view

    def reset(self, flags):
        self.o.reset(flags)

    def propagate(self, x):
        return torch.from_numpy(self.o.propagate(x.numpy()))

    def backpropagate(self, x, out_diff, want_in_diff):
        saved = self.o.get_corr()
        self.o.set_corr(np.zeros_like(saved))
        d = self.o.backpropagate(x.numpy(), out_diff.numpy(), momentum=0.0)
        self.g.copy_(torch.from_numpy(self.o.get_corr()))
        self.o.set_corr(saved)
        return torch.from_numpy(d) if want_in_diff else None

    def apply(self, momentum, lr):
        self.o.set_corr(momentum * self.o.get_corr() + self.g.numpy())
        self.o.update(lr)

    def params(self):
        return self.o.get_params()


class NumpyAffineLayer:
    def __init__(self, W, b):
        self.W, self.b = W.copy(), b.copy()
        self.Wc, self.bc = np.zeros_like(W), np.zeros_like(b)
        self.num_params = W.size + b.size

    def bind_grad(self, view):
        self.g = view

    def reset(self, flags):
        pass

    def propagate(self, x):
        return torch.from_numpy(comp.affine_propagate(x.numpy(), self.W, self.b))

    def backpropagate(self, x, out_diff, want_in_diff):
        od = out_diff.numpy()
        g = np.concatenate([(od.T @ x.numpy()).ravel(), od.sum(0)])
        self.g.copy_(torch.from_numpy(g))
        return torch.from_numpy(comp.affine_backpropagate(od, self.W)) if want_in_diff else None

    def apply(self, momentum, lr):
        g = self.g.numpy()
        self.Wc[...] = momentum * self.Wc + g[:self.W.size].reshape(self.W.shape)
        self.bc[...] = momentum * self.bc + g[self.W.size:]
        self.W -= lr * self.Wc
        self.b -= lr * self.bc

    def params(self):
        return np.concatenate([self.W.ravel(), self.b])


class NumpyLoss:
    def eval(self, net_out, targets, mask):
        post = comp.softmax(net_out.numpy())
        diff, xe, _ent, correct, valid = comp.xent_eval_masked(post, targets.numpy(), mask.numpy().astype(post.dtype))
        return torch.from_numpy(diff), xe, correct, valid


def make_stack(dims, S, seed, dtype=np.float64, scale=0.3):
    """dims = (I, C, R, n_lstm, n_out): n_lstm stacked LSTMs (I->R, R->R, ...) + Affine R->n_out.  Returns the flat
    parameter arrays so that the same numbers can be loaded into device layers."""
    from oracle.oracle import make_params
    I, C, R, n_lstm, n_out = dims
    rng = np.random.RandomState(seed)
    lstm = [make_params(I if l == 0 else R, C, R, scale=scale, seed=seed + 1 + l, dtype=dtype) for l in range(n_lstm)]
    W = ((rng.rand(n_out, R) - 0.5) * 2 * scale).astype(dtype)
    b = ((rng.rand(n_out) - 0.5) * 2 * scale).astype(dtype)
    return lstm, W, b


def cpu_layers(dims, S, lstm, W, b, dtype=np.float64):
    I, C, R, n_lstm, n_out = dims
    layers = [OracleLstmLayer(I if l == 0 else R, C, R, S, lstm[l], dtype) for l in range(n_lstm)]
    layers.append(NumpyAffineLayer(W.astype(dtype), b.astype(dtype)))
    return layers
