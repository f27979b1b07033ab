"""Observation models for WHVI networks; interface of the reference's src/likelihoods.py."""
import math

import torch
import torch.nn as nn

__all__ = ["Likelihood", "GaussianLikelihood"]

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


class GaussMNLLFunction(torch.autograd.Function):
    """``scale * sum log N(y | y_hat, sigma^2)`` over an MC prediction tensor as one HIP reduction
    (``whvi_gauss_mnll_f32``) with a one-launch closed-form backward; float32 GPU tensors.
    ``y`` is an expanded view of ``y_hat``'s shape; ``y_hat`` is read (and its gradient written) in its own
    memory layout, so the ``(batch, out, n_mc)`` permuted view of the batched pass costs no copy."""

    @staticmethod
    def forward(ctx, y, y_hat, sigma, scale):
        from whvi_amd import _hip
        if any(st == 0 and sz > 1 for st, sz in zip(y_hat.stride(), y_hat.size())):
            y_hat = y_hat.contiguous()                 # expanded predictions: the gradient needs real storage
        part = _hip.gauss_mnll(y, y_hat, sigma, scale)
        ctx.save_for_backward(y, y_hat, sigma, part)
        ctx.scale = float(scale)
        # one block of partial sums (up to 1 024 predictions: a training batch of the UCI protocol): its slot IS the result
        return part[0, 0] if part.size(0) == 1 else part[:, 0].sum()

    @staticmethod
    def backward(ctx, grad_out):
        from whvi_amd import _hip
        y, y_hat, sigma, part = ctx.saved_tensors
        if torch.is_grad_enabled():                    # create_graph=True: the closed form as differentiable ops
            diff = y - y_hat
            grad_yhat = grad_out * ctx.scale * diff / (sigma * sigma)
            grad_sigma = grad_out * ctx.scale * ((diff / sigma).square().sum() - diff.numel()) / sigma
        else:
            grad_yhat, grad_sigma = _hip.gauss_mnll_bwd(grad_out.contiguous(), part, y, y_hat, sigma, ctx.scale)
        grad_y = -grad_yhat if ctx.needs_input_grad[0] else None        # autograd sums the expanded axis
        return grad_y, grad_yhat, grad_sigma, None


class Likelihood:
    """Base class: a likelihood that contributes nothing (mean negative log likelihood 0)."""

    def mnll_batch_estimate(self, *args, **kwargs):
        return 0.0


class GaussianLikelihood(nn.Module, Likelihood):
    """Homoscedastic Gaussian noise model ``y ~ N(y_hat, sigma^2)`` with a learnable scalar ``sigma``
    (the parameter is called ``sigma`` in checkpoints: ``likelihood.sigma``)."""

    def __init__(self, sigma: float = 1.0):
        super().__init__()
        self.sigma = nn.Parameter(torch.tensor(sigma))

    def log_density(self, y: torch.Tensor, y_hat: torch.Tensor) -> torch.Tensor:
        """Element-wise ``log N(y | y_hat, sigma^2)``; ``y`` broadcasts against ``y_hat``."""
        z = (y - y_hat) / self.sigma
        return -0.5 * z * z - torch.log(self.sigma) - _HALF_LOG_2PI

    def mnll_batch_estimate(self, y: torch.Tensor, y_hat: torch.Tensor, n: int) -> torch.Tensor:
        """Mini-batch estimate of the data set's mean negative log likelihood.

        ``y_hat`` holds Monte-Carlo predictions ``(m, n_out, n_mc)`` and ``y`` the targets ``(m, n_out)``; the
        log density is summed over batch, outputs and samples, averaged over samples and rescaled from the
        batch (m points) to the data set (n points): ``-n / (m * n_mc) * sum`` -- the estimator of
        src/likelihoods.py:18-29, evaluated as one fused reduction instead of a Python loop over outputs."""
        m, n_out, n_mc = y_hat.size()
        if (y_hat.device.type == "cuda" and y_hat.dtype == y.dtype == self.sigma.dtype == torch.float32
                and y.device == y_hat.device == self.sigma.device and y_hat.numel() > 0):
            return GaussMNLLFunction.apply(y.reshape(m, n_out, 1).expand(m, n_out, n_mc), y_hat, self.sigma,
                                           -n / (m * n_mc))
        total = self.log_density(y.reshape(m, n_out, 1), y_hat).sum()
        return -n / (m * n_mc) * total
