"""WHVI layers (mirror of the reference's src/layers.py)."""
import torch.nn as nn

from whvi_amd.utils import is_pow_of_2
from whvi_amd.weights import WHVISquarePow2Matrix, WHVIStackedMatrix, WHVIColumnMatrix

__all__ = ["WHVI", "WHVILinear"]


class WHVI:
    """Marker base class: anything with a ``kl`` property (src/layers.py:7-16)."""

    @property
    def kl(self):
        return 0.0


class WHVILinear(nn.Module, WHVI):
    def __init__(self, n_in, n_out, lambda_=1e-5, bias=False):
        """WHVI feed-forward layer (src/layers.py:19-38).

        Picks the weight parameterisation from the shape: a column matrix when either side is
        1, a single square matrix when ``n_in == n_out`` is a power of two, a stack otherwise.
        The chosen module is ``self.weight_submodule`` (state_dict keys depend on that name).
        """
        super().__init__()
        if n_in == 1:
            self.weight_submodule = WHVIColumnMatrix(n_out, lambda_=lambda_, bias=bias)
        elif n_out == 1:
            self.weight_submodule = WHVIColumnMatrix(n_in, lambda_=lambda_, transposed=True, bias=bias)
        elif n_in == n_out and is_pow_of_2(n_in):
            self.weight_submodule = WHVISquarePow2Matrix(n_in, lambda_=lambda_, bias=bias)
        else:
            self.weight_submodule = WHVIStackedMatrix(n_in, n_out, lambda_=lambda_, bias=bias)

    @property
    def kl(self):
        return self.weight_submodule.kl

    def forward(self, x):
        return self.weight_submodule.forward(x)

    def forward_mc(self, x, n_samples):
        """``n_samples`` stochastic passes at once: (batch, n_in) or (n_samples, batch, n_in) ->
        (n_samples, batch, n_out).  Used by ``WHVINetwork`` instead of its per-sample loop."""
        out = self.weight_submodule.forward_mc(x, n_samples)
        # KL of exactly this pass when the fused reparameterisation kernel produced it (GPU), else None
        self._mc_kl = getattr(self.weight_submodule, "_mc_kl", None)
        return out
