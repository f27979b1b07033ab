"""oracle/whvi_oracle.py -- TEST INFRASTRUCTURE: numpy restatement of the WHVI weight-sample path
(src/weights.py, src/layers.py, src/utils.py) on top of the C butterfly oracle.

Pinned by tests/test_oracle.py against bundles recorded from the live reference
(tests/golden/whvi_golden.npz).  Every random draw is an explicit argument (``eps``), in the order
the reference draws them.  float32 throughout, one numpy op per reference torch op.
"""
import math

import numpy as np

from . import binding as _b

F32 = np.float32


def softplus(x):
    """F.softplus with torch's defaults (beta=1, threshold=20), src/weights.py:50."""
    x = np.asarray(x, dtype=F32)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))).astype(F32)


def kl_diag_normal(mu1, sd1, mu2, sd2):
    """src/utils.py:49-71, same term order."""
    d = F32(len(mu1))
    delta = (mu2 - mu1).astype(F32)
    return F32(0.5) * (np.sum(np.log(sd2), dtype=F32) - np.sum(np.log(sd1), dtype=F32) - d
                       + np.sum(sd1 / sd2, dtype=F32) + np.dot(delta, (delta / sd2).astype(F32)))


def w_bar(s1, s2, u):
    """src/weights.py:73: matmul_diag_left(s1, fwht(matmul_diag_left(u, fwht(diag(s2))))) with the
    butterfly FWHT (what the GPU path and the host path for D >= 4096 run)."""
    D = len(s1)
    x = np.diag(np.asarray(s2, dtype=F32)).astype(F32)
    return _b.pipeline(x, a=s1, b=u, c=None, n_samples=1, sample_stride=D, group_rows=D, axis="row")


class Square:
    """WHVISquarePow2Matrix (src/weights.py:13-108)."""

    def __init__(self, s1, s2, g_mu, g_rho, lambda_, bias=None):
        self.s1, self.s2, self.g_mu, self.g_rho = (np.asarray(v, dtype=F32) for v in (s1, s2, g_mu, g_rho))
        self.lambda_ = lambda_
        self.bias = None if bias is None else np.asarray(bias, dtype=F32)
        self.D = len(self.s1)

    @property
    def g_sigma(self):
        return softplus(self.g_rho)

    @property
    def kl(self):
        return kl_diag_normal(self.g_mu, self.g_sigma, np.zeros(self.D, F32),
                              (np.ones(self.D, F32) * F32(self.lambda_)).astype(F32))

    def sample(self, eps):
        g_tilde = (self.g_mu + self.g_sigma * np.asarray(eps, F32)).astype(F32)
        return w_bar(self.s1, self.s2, g_tilde)

    def sample_lrt(self, h, eps):
        W = (w_bar(self.s1, self.s2, self.g_mu)
             + w_bar(self.s1, self.s2, (self.g_sigma * np.asarray(eps, F32)).astype(F32))).astype(F32)
        return (np.asarray(h, F32) @ W.T).astype(F32)

    def forward(self, x, eps):
        out = self.sample_lrt(x, eps)
        return out + self.bias if self.bias is not None else out


def setup_dimensions(D_in, D_out):
    """src/weights.py:135-160."""
    next_power = 2 ** math.ceil(math.log(D_in, 2))
    if next_power == 2 * D_in:
        padding = 0
    else:
        padding = next_power - D_in
        D_in = next_power
    stack, remainder = divmod(D_out, D_in)
    if remainder != 0:
        stack += 1
        D_out = D_in * stack
    return D_in, D_out, padding, stack


class Stacked:
    """WHVIStackedMatrix (src/weights.py:111-208); ``eps`` is one vector per sub-matrix, in order."""

    def __init__(self, n_in, n_out, squares, bias=None):
        self.n_in, self.n_out = n_in, n_out
        self.D_in, self.D_out, self.padding, self.stack = setup_dimensions(n_in, n_out)
        assert len(squares) == self.stack
        self.squares = squares
        self.bias = None if bias is None else np.asarray(bias, dtype=F32)

    @property
    def kl(self):
        total = F32(0)
        for s in self.squares:
            total = F32(total + s.kl)
        return total

    def forward(self, x, eps_list):
        x = np.asarray(x, F32)
        xp = np.zeros(x.shape[:-1] + (self.D_in,), F32)
        xp[..., :self.n_in] = x
        out = np.concatenate([s.sample_lrt(xp, e) for s, e in zip(self.squares, eps_list)], axis=1)
        if self.bias is not None:
            out = out + self.bias
        return out[..., :self.n_out]


class Column:
    """WHVIColumnMatrix (src/weights.py:211-251)."""

    def __init__(self, n, square, transposed, bias=None):
        self.D, self.square, self.transposed = n, square, transposed
        self.bias = None if bias is None else np.asarray(bias, dtype=F32)

    @property
    def kl(self):
        return self.square.kl

    def forward(self, x, eps):
        m = self.square.sample(eps).reshape(-1, 1)[:self.D]
        if self.transposed:
            m = m.T
        out = (np.asarray(x, F32) @ m.T).astype(F32)      # F.linear(x, W, bias)
        return out + self.bias if self.bias is not None else out


def layer_from_params(n_in, n_out, lambda_, params):
    """Build the oracle object for WHVILinear(n_in, n_out) (dispatch of src/layers.py:31-38) from a
    dict of arrays keyed like the reference's ``named_parameters()``."""
    def square(prefix):
        return Square(params[prefix + "s1"], params[prefix + "s2"], params[prefix + "g_mu"],
                      params[prefix + "g_rho"], lambda_, params.get(prefix + "bias"))

    is_pow2 = n_in > 0 and (n_in & (n_in - 1)) == 0
    if n_in == 1 or n_out == 1:
        n = n_out if n_in == 1 else n_in
        sq = square("weight_submodule.weight_submodule.")
        sq.bias = None
        return Column(n, sq, transposed=(n_in != 1), bias=params.get("weight_submodule.bias"))
    if n_in == n_out and is_pow2:
        return square("weight_submodule.")
    stack = setup_dimensions(n_in, n_out)[3]
    squares = [square(f"weight_submodule.weight_matrices.{j}.") for j in range(stack)]
    return Stacked(n_in, n_out, squares, bias=params.get("weight_submodule.bias"))
