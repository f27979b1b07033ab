"""The WHVI layer: interface of the reference's src/layers.py (``WHVILinear(n_in, n_out, lambda_, bias)``,
``.weight_submodule``, ``.kl``, ``.forward``) plus the batched Monte-Carlo entry point ``forward_mc``."""
import torch.nn as nn

from whvi_amd.utils import is_pow_of_2
from whvi_amd.weights import WHVIColumnMatrix, WHVISquarePow2Matrix, WHVIStackedMatrix

__all__ = ["WHVI", "WHVILinear"]


class WHVI:
    """Mixin marking a module as variational; non-variational members contribute no KL."""

    @property
    def kl(self):
        return 0.0


def _choose_parameterisation(n_in, n_out, lambda_, bias, mode="reference"):
    """Shape -> weight module, the rule of src/layers.py:31-38:

    =====================================  ==========================================
    one input feature                      column matrix (n_out, 1)
    one output feature                     the same, transposed: row matrix (1, n_in)
    n_in == n_out, a power of two          one square WHVI matrix
    anything else                          stack of square blocks of size 2^ceil(log2 n_in)
    =====================================  ==========================================
    """
    if mode == "fastfood":
        from whvi_amd.fastfood import WHVIFastfoodMatrix
        if n_in != n_out or not is_pow_of_2(n_in):
            raise ValueError("mode='fastfood' is implemented for square power-of-two layers")
        return WHVIFastfoodMatrix(n_in, lambda_=lambda_, bias=bias)
    if mode != "reference":
        raise ValueError("mode must be 'reference' or 'fastfood'")
    if n_in == 1:
        return WHVIColumnMatrix(n_out, lambda_=lambda_, bias=bias)
    if n_out == 1:
        return WHVIColumnMatrix(n_in, lambda_=lambda_, transposed=True, bias=bias)
    if n_in == n_out and is_pow_of_2(n_in):
        return WHVISquarePow2Matrix(n_in, lambda_=lambda_, bias=bias)
    return WHVIStackedMatrix(n_in, n_out, lambda_=lambda_, bias=bias)


class WHVILinear(nn.Module, WHVI):
    """Feed-forward layer whose weight matrix is a Walsh-Hadamard variational factorisation.

    ``lambda_`` is the prior variance; ``bias`` adds a plain (non-variational) bias.  The parameterisation
    lives in ``self.weight_submodule`` -- that attribute name is part of the checkpoint format.

    ``mode`` (keyword, not in the reference): ``"reference"`` (default) reproduces the reference as written;
    ``"fastfood"`` opts in to the textbook operator S1 H diag(g) H S2 applied to activations without materialising W
    (``whvi_amd.fastfood``) -- same parameters and KL, different (non-diagonal) weight matrix, square power-of-two
    layers only."""

    def __init__(self, n_in, n_out, lambda_=1e-5, bias=False, mode="reference"):
        super().__init__()
        self.weight_submodule = _choose_parameterisation(n_in, n_out, lambda_, bias, mode)

    @property
    def kl(self):
        """KL from the prior to the variational posterior of this layer's weights."""
        return self.weight_submodule.kl

    def forward(self, x):
        """One stochastic pass (one draw of the weights)."""
        return self.weight_submodule.forward(x)

    def fuses_relu(self, x):
        """True when this layer's batched pass can fold an ``nn.ReLU`` in front of / behind it into its own launch."""
        fn = getattr(self.weight_submodule, "fuses_relu", None)
        return bool(fn is not None and fn(x))

    def forward_mc(self, x, n_samples, relu_in=False, relu_out=False):
        """``n_samples`` stochastic passes at once: (batch, n_in) or (n_samples, batch, n_in) ->
        (n_samples, batch, n_out).  Used by ``WHVINetwork`` instead of its per-sample loop.  ``relu_in`` / ``relu_out``
        (``WHVINetwork.forward_batched`` sets them only where ``fuses_relu``): neighbouring activations folded in."""
        if relu_in or relu_out:
            out = self.weight_submodule.forward_mc(x, n_samples, relu_in=relu_in, relu_out=relu_out)
        else:
            out = self.weight_submodule.forward_mc(x, n_samples)
        # KL of exactly this pass when the fused reparameterisation kernel produced it (GPU), else None.  MOVED, not copied:
        # the tensor carries this pass's autograd graph, and a second reference on the inner module would keep that graph --
        # and the parameters' gradient accumulators, created on whatever stream ran this pass -- alive until the NEXT pass
        # (round 3: the accumulators of GraphedTrainStep's side-stream warm-up survived into the capture this way)
        self._mc_kl = getattr(self.weight_submodule, "_mc_kl", None)
        self.weight_submodule._mc_kl = None
        return out
