"""Observation models for WHVI networks; interface of the reference's src/likelihoods.py."""
import math

import torch
import torch.nn as nn

__all__ = ["Likelihood", "GaussianLikelihood"]

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


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
        total = self.log_density(y.reshape(m, n_out, 1), y_hat).sum()
        return -n / (m * n_mc) * total
