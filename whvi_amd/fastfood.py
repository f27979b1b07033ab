"""Opt-in "fastfood" mode: the TEXTBOOK operator  x -> S1 . H . diag(g) . H . S2 . x  applied to activations.

NOT reference-equivalent, on purpose.  As written, the reference's weight construction scales ROWS between row
transforms and collapses to the diagonal matrix D diag(s1 g s2) (SURVEY.md finding 1); `WHVILinear` reproduces
that by default.  The paper's parameterisation (Rossi et al., the report's eq. for W = S1 H diag(g) H S2) is the
column-scaling pipeline the synthetic (batch, n_samples, D) kernel benchmarks exercise (BASELINE config 3).  This
module gives that pipeline -- `whvi_fused_shs_*` with `axis = COL`, include/whvi_hip.h -- a Module-level consumer:

    y[k] = s1 * fwht(g_k * fwht(s2 * x[k])),      g_k = g_mu + softplus(g_rho) * eps_k,   eps_k ~ N(0, I)

for every Monte-Carlo sample k in ONE launch, O(D log D) per row and without ever materialising a D x D weight
(the reference-equivalent path builds S matrices of D^2 floats and a batched GEMM).  Same parameter names, shapes,
initialisation and KL as `WHVISquarePow2Matrix` (src/weights.py:28-32, :52-64), so checkpoints interchange.

Parity: bit-exact against `oracle.pipeline(axis="col")` (compositions of the reference's own primitives --
`matmul_diag_right`, src/utils.py:15-23, and the C++ FWHT) and against the dense product with `build_H` in float64
(tests/test_fastfood.py).  Host tensors run the same ops through the host FWHT.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from whvi_amd.utils import is_pow_of_2

__all__ = ["FastfoodFunction", "WHVIFastfoodMatrix"]


def _fwht(x):
    """Row FWHT of a 2-D tensor on its own device: the HIP kernels or the native host library."""
    if x.device.type == "cuda":
        from whvi_amd import _hip
        return _hip.fwht_rows(x)
    import fwht_cpp
    return fwht_cpp.forward(x)


def _pipeline(x, a, b, c, n_samples, sample_stride, shared=False):
    """a * fwht(b[s(r)] * fwht(c * x[r])) for every row r, s(r) = (r // sample_stride) % n_samples.  ``shared``: ``x`` holds
    ONE sample's rows (``sample_stride`` of them) and every sample reads them -- on the GPU straight from the caches
    (WHVI_FUSED_SRC_SHARED), elsewhere after expanding."""
    if shared:
        if x.device.type == "cuda":
            from whvi_amd import _hip
            if _hip.fused_src_shared_supported(x.dtype, x.size(1)):
                # fwht(c * x) is the same for every sample: once, then ONE transform per sample on the shared result
                # (the same multiplies and butterflies in the same order as the two-transform launch: the same bits)
                t = _hip.fused_shs(x, None, c.reshape(1, -1), None, axis="col", n_samples=1, one_transform=True)
                return _hip.fused_shs(t, a, b, None, axis="col", n_samples=n_samples, sample_stride=sample_stride,
                                      src_shared=True, one_transform=True)
        x = x.repeat(n_samples, 1)
    if x.device.type == "cuda":
        from whvi_amd import _hip
        if _hip.fused_supported(x.dtype, x.size(1)):
            return _hip.fused_shs(x, a, b, c, axis="col", n_samples=n_samples, sample_stride=sample_stride)
        # rows longer than one wavefront tile (D > 8192; f64 > 4096): the fused launch does not exist, the plain
        # transform does (a block per row and beyond) -- the same multiplies and butterflies as separate launches
    rows = torch.arange(x.size(0), device=x.device) // sample_stride % n_samples
    return a * _fwht(b[rows] * _fwht(c * x))


def _scale_fwht(x, vec, n_samples=1, sample_stride=1):
    """``fwht(vec[s(r)] * x[r])`` for every row -- ``vec`` of shape ``(D,)`` (shared) or ``(n_samples, D)`` with
    s(r) = (r // sample_stride) % n_samples -- as one launch of the fused kernel's one-transform form on the GPU."""
    per_sample = vec.dim() == 2
    if x.device.type == "cuda":
        from whvi_amd import _hip
        if _hip.fused_src_shared_supported(x.dtype, x.size(1)):
            return _hip.fused_shs(x, None, vec if per_sample else vec.reshape(1, -1), None, axis="col",
                                  n_samples=n_samples if per_sample else 1, sample_stride=sample_stride if per_sample else 1,
                                  one_transform=True)
    if per_sample:
        rows = torch.arange(x.size(0), device=x.device) // sample_stride % n_samples
        return _fwht(vec[rows] * x)
    return _fwht(vec * x)


class FastfoodFunction(torch.autograd.Function):
    """``y = a * fwht(b_s * fwht(c * x))`` on rows ``(n_samples, rows_per_sample, D)`` flattened, ``b``: (S, D).

    Forward: one fused launch.  The operator is linear in x and H is symmetric, so the gradient with respect to x is
    the same launch with a and c exchanged; the gradients of the three diagonals are products with the two
    intermediate transforms (recomputed, not stored) summed over rows.  First order only."""

    @staticmethod
    def forward(ctx, x, a, b, c, n_samples, sample_stride, shared=False):
        """``shared``: ``x`` is ``(sample_stride, D)``, the same rows for every sample (a layer's first Monte-Carlo pass on
        a ``(batch, D)`` input); the result still has ``n_samples * sample_stride`` rows."""
        ctx.save_for_backward(x, a, b, c)
        ctx.n_samples, ctx.sample_stride, ctx.shared = int(n_samples), int(sample_stride), bool(shared)
        return _pipeline(x, a, b, c, ctx.n_samples, ctx.sample_stride, ctx.shared)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        x, a, b, c = ctx.saved_tensors
        S, stride = ctx.n_samples, ctx.sample_stride
        grad_y = grad_y.contiguous()
        need_x, need_a, need_b, need_c = ctx.needs_input_grad[:4]
        grad_x = grad_a = grad_b = grad_c = None
        if ctx.shared:
            # every sample read the same rows: their gradients add up (what autograd does for an expanded input)
            fold = lambda g: None if g is None else g.view(S, stride, -1).sum(dim=0)   # noqa: E731
            x = x.repeat(S, 1)
        else:
            fold = lambda g: g                                                        # noqa: E731
        if need_x and not (need_a or need_b or need_c):
            return fold(_pipeline(grad_y, c, b, a, S, stride)), None, None, None, None, None, None
        # Every transform of the backward pass is "scale, then FWHT" (optionally scaled again): ONE launch each through the
        # one-transform form of the fused kernel where it exists, the multiply + plain transform elsewhere -- the same
        # roundings either way (tests/test_streaming_parity_gpu.py pins the launch to multiply + fwht_rows bit for bit)
        t1 = _scale_fwht(x, c)                              # forward intermediate fwht(c * x), recomputed
        if need_a:
            grad_a = (grad_y * _scale_fwht(t1, b, S, stride)).sum(dim=0)
        v = _scale_fwht(grad_y, a)                          # gradient at (b * t1)
        if need_b:
            prod = v * t1
            if stride * S == x.size(0):                     # (S, rows_per_sample, D) layout: one segmented sum
                grad_b = prod.view(S, stride, -1).sum(dim=1)
            else:
                rows = torch.arange(x.size(0), device=x.device) // stride % S
                grad_b = torch.zeros_like(b).index_add_(0, rows, prod)
        if need_c or need_x:
            w = _scale_fwht(v, b, S, stride)                # gradient at (c * x)
            if need_c:
                grad_c = (w * x).sum(dim=0)
            if need_x:
                grad_x = fold(c * w)
        return grad_x, grad_a, grad_b, grad_c, None, None, None


class WHVIFastfoodMatrix(nn.Module):
    """Square (D, D) WHVI layer in fastfood mode (see the module docstring): parameters ``s1, s2, g_mu, g_rho``
    (+ optional ``bias``) as in ``WHVISquarePow2Matrix``; ``forward(x)`` draws one eps, ``forward_mc(x, S)`` draws S
    and runs all samples in one launch."""

    def __init__(self, D, lambda_=1e-5, bias=False):
        super().__init__()
        if not is_pow_of_2(D):
            raise ValueError("fastfood mode needs a power-of-two width")
        self.D, self.lambda_, self.padding = D, lambda_, 0
        # creation order = WHVISquarePow2Matrix's (bias, s1, s2, g_mu, g_rho): the same seed gives the same parameters
        self.bias = nn.Parameter(torch.zeros(1, D)) if bias else None
        self.s1 = nn.Parameter(torch.randn(D) * 0.01)
        self.s2 = nn.Parameter(torch.randn(D) * 0.01)
        self.g_mu = nn.Parameter(torch.zeros(D))
        self.g_rho = nn.Parameter(torch.rand(D) - 3)

    @property
    def g_sigma(self):
        return F.softplus(self.g_rho)

    @property
    def kl(self):
        from whvi_amd.weights import _posterior_kl
        return _posterior_kl(self.g_mu.unsqueeze(0), self.g_rho.unsqueeze(0), self.lambda_)

    def dense_weight(self, g):
        """The (D, D) matrix this layer applies for one g -- ``diag(s1) H diag(g) H diag(s2)`` -- built densely (tests,
        inspection; never used by forward)."""
        from whvi_amd.utils import build_H
        H = build_H(self.D, self.g_mu.device).to(self.g_mu.dtype)
        return self.s1.unsqueeze(1) * (H @ (g.unsqueeze(1) * (H * self.s2.unsqueeze(0))))

    def forward_mc(self, x, n_samples):
        """(batch, D) or (n_samples, batch, D) -> (n_samples, batch, D); sample k uses row k of one
        ``randn(n_samples, D)`` draw."""
        eps = torch.randn(n_samples, self.D, device=self.g_mu.device)
        g = self.g_mu + self.g_sigma * eps                                        # (S, D)
        if x.dim() == 2:
            # a (batch, D) input shared by all samples: read by every sample straight from the caches, never expanded
            batch = x.size(0)
            out = FastfoodFunction.apply(x.contiguous(), self.s1, g, self.s2, n_samples, batch, True)
            out = out.view(n_samples, batch, self.D)
            return out + self.bias if self.bias is not None else out
        batch = x.size(1)
        rows = x.reshape(n_samples * batch, self.D).contiguous()
        out = FastfoodFunction.apply(rows, self.s1, g, self.s2, n_samples, batch).view(n_samples, batch, self.D)
        return out + self.bias if self.bias is not None else out

    def forward(self, x):
        out = self.forward_mc(x.reshape(-1, self.D), 1)[0].reshape(x.shape)
        return out
