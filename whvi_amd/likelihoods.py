"""Likelihoods (mirror of src/likelihoods.py)."""
import math

import torch
import torch.nn as nn

__all__ = ["Likelihood", "GaussianLikelihood"]


class Likelihood:
    def mnll_batch_estimate(self, *args, **kwargs):
        return 0.0


class GaussianLikelihood(nn.Module, Likelihood):
    """Homoscedastic Gaussian likelihood with a learnable ``sigma`` (src/likelihoods.py:13-29)."""

    def __init__(self, sigma: float = 1.0):
        super().__init__()
        self.sigma = nn.Parameter(torch.tensor(sigma))

    def mnll_batch_estimate(self, y: torch.Tensor, y_hat: torch.Tensor, n: int) -> torch.Tensor:
        """Mini-batch estimate of the mean negative log likelihood,
        ``-n / (m * n_mc) * sum log N(y | y_hat, sigma)`` over batch, outputs and MC samples
        (src/likelihoods.py:18-29); ``y_hat`` is ``(m, n_out, n_mc)``, ``y`` is ``(m, n_out)``.
        One fused reduction instead of the reference's per-output Python loop."""
        m, n_out, n_mc = y_hat.size()
        resid = (y.reshape(m, n_out, 1) - y_hat) / self.sigma
        log_prob = -0.5 * resid ** 2 - torch.log(self.sigma) - 0.5 * math.log(2 * math.pi)
        return -n / (m * n_mc) * log_prob.sum()
